"""Shim: see ../README.md (golden generation only)."""
