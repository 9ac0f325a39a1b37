"""The optimizer of the train driver as one launch per step.

The reference builds `torch.optim.Adam(model.parameters(), lr=..., betas=(0.9, 0.98), weight_decay=...)`
(scripts/training.py:174) and calls `.step()` once per batch (src/train.py:96).  torch's own Adam works unchanged on
the HIP-backed modules; this class is the same update (same state keys, interchangeable state_dict) issued through
carca_adam_step: every parameter of every group in ONE kernel launch, ~0.03 ms of host time instead of ~0.25 ms.
"""
from __future__ import annotations

from typing import Iterable, Tuple

import torch

from . import _lib
from .ops import CarcaHipError, _stream


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) without amsgrad / maximize / capturable: fp32 CUDA
    parameters with dense gradients.  State per parameter: 'step' (CPU float tensor), 'exp_avg', 'exp_avg_sq'."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._steps = {}  # id(parameter) -> step count as a python int (state['step'] stays the tensor torch expects)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            steps = set()
            arr = (_lib.AdamTensor * len(live))()
            keep, counters = [], []
            for i, p in enumerate(live):
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise CarcaHipError("carca Adam: parameters must be contiguous fp32 CUDA tensors with dense gradients")
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = self._steps.get(id(p))
                if t is None:
                    t = int(st["step"])
                self._steps[id(p)] = t + 1
                steps.add(t + 1)
                counters.append(st["step"])
                a = arr[i]
                a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), \
                    p.numel()
            torch._foreach_add_(counters, 1.0)
            # the kernel writes the parameters behind torch's back: bump their version counters, which is what the
            # modules' packed-weight caches (and autograd's saved-tensor checks) go by
            torch.autograd.graph.increment_version(live)
            b1, b2 = group["betas"]
            if len(steps) == 1:
                _lib.check(lib.carca_adam_step(arr, len(live), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                               float(group["weight_decay"]), steps.pop(), _stream()), "adam_step")
            else:  # parameters that joined later carry their own step count: one launch per count
                for t in sorted(steps):
                    idx = [i for i, p in enumerate(live) if self._steps[id(p)] == t]
                    sub = (_lib.AdamTensor * len(idx))(*[arr[i] for i in idx])
                    _lib.check(lib.carca_adam_step(sub, len(idx), float(group["lr"]), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]), t, _stream()),
                               "adam_step")
        return loss
