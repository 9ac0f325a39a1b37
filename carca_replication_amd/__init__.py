"""carca_replication_amd: the CARCA forward/backward hot path as hand-written gfx950 (MI355X) kernels.

Layout
  csrc/        HIP kernels + the C ABI of include/carca_hip.h  (built in-tree into libcarca_hip.so)
  _lib.py      build + ctypes binding of the C ABI
  ops.py       torch-tensor front end of the C ABI (pointers in, tensors out)
  modules.py   the reference's nn.Module surface (same names / state_dict / forward contracts)
  autograd.py  torch.autograd.Function wrappers around the backward kernels
  dist.py      user sharding + RCCL gradient all-reduce for N GPUs
"""
from ._lib import CarcaHipError, build, is_built, load  # noqa: F401

__all__ = ["CarcaHipError", "build", "is_built", "load"]
