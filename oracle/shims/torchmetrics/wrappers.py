from . import Metric


class BootStrapper(Metric):
    def __init__(self, base_metric, **kwargs):
        super().__init__()
        self.base = base_metric

    def update(self, *args):
        self.base.update(*args)

    def compute(self):
        return self.base.compute()

    def reset(self):
        self.base.reset()
