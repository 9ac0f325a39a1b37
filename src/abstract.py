"""Drop-in import path `src.abstract` (scripts/training.py:13 of the reference): the plug-in ABCs."""
from carca_replication_amd.modules import Decoder, Embedding, Encoder, Encoding, Model  # noqa: F401

__all__ = ["Model", "Embedding", "Encoding", "Encoder", "Decoder"]
