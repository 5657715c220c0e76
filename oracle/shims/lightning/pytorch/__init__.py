from .. import LightningModule  # noqa: F401
