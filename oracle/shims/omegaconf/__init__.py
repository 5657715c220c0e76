class DictConfig(dict):
    __getattr__ = dict.__getitem__
