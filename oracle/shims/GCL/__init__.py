"""Shim: see ../README.md (golden generation only)."""
from . import losses, models  # noqa: F401
