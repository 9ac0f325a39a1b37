"""Drop-in import path `src.carca` (scripts/training.py:14-28 of the reference).

The hot-path classes (AllEmbedding, SelfAttentionBlock, CrossAttentionBlock, CARCA, BinaryCrossEntropy and
the encodings) are the HIP-backed mirrors in carca_replication_amd.modules.
"""
from carca_replication_amd.modules import (  # noqa: F401
    CARCA,
    AllEmbedding,
    AttrCtxEmbedding,
    AttrEmbedding,
    BinaryCrossEntropy,
    CrossAttentionBlock,
    DotProduct,
    IdEmbedding,
    IdentityEncoding,
    LearnableEncoding,
    MLPIdEmbedding,
    MultiHeadAttention,
    PositionalEncoding,
    SelfAttentionBlock,
    WeightedDotProduct,
)
