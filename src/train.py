"""Drop-in import path `src.train` (scripts/training.py:30 of the reference)."""
from carca_replication_amd.train import compute_HR, compute_NDCG, evaluate, train  # noqa: F401
