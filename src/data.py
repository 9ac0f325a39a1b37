"""Drop-in import path `src.data` (scripts/training.py:29 of the reference)."""
from carca_replication_amd.data import (  # noqa: F401
    CARCADataset,
    get_sequences,
    get_test_sequences,
    get_train_sequences,
    load_attrs,
    load_ctx,
    load_profiles,
    pad_profile,
    sample_negatives,
    set_datapath,
)
