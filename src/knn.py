"""Drop-in import path `src.knn` (the reference's src/knn.py:8-21): the attribute-similarity baseline model."""
from carca_replication_amd.modules import KNN  # noqa: F401

__all__ = ["KNN"]
