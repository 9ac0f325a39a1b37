"""Drop-in import path `src.utils` (reference src/utils.py)."""
from carca_replication_amd.modules import get_mask, to  # noqa: F401

__all__ = ["get_mask", "to"]
