class DictConfig(dict):
    __getattr__ = dict.__getitem__


class OmegaConf:
    """``OmegaConf.load`` (reference data/node.py:22): the yaml as plain containers (golden generation never reads it)."""

    @staticmethod
    def load(path):
        import yaml
        with open(path) as f:
            return yaml.safe_load(f)
