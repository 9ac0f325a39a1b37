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
        self._marked = set()  # id(parameter) of the tables whose touched rows were announced for the coming step

    def mark_rows(self, param: torch.nn.Parameter, ids: torch.Tensor) -> bool:
        """Touched-row update of an embedding table: `ids` are the rows this step's gradient
        can be non-zero in.  The step then skips every row that has never been marked -- exactly the rows whose
        g = m = v = 0, for which dense Adam's update is exactly 0 -- so parameters and state keep the bits of the dense
        sweep without moving 7 floats per element of the whole table (1 M x 128: 2.5 GB per step).
        Must be called before EVERY step from the first one on; a step without it marks the whole table touched (dense
        from then on: still correct).  Refused (returns False) when weight decay would move untouched rows, or for a parameter of no group."""
        group = next((g for g in self.param_groups if any(q is param for q in g["params"])), None)
        if group is None:  # not this optimizer's parameter (frozen, or stepped by another optimizer): nothing to announce
            return False
        if group["weight_decay"] != 0 or param.dim() != 2:
            return False
        st = self.state[param]
        if "row_touched" not in st:
            started = bool(st) and float(st.get("step", 0.0)) > 0
            fill = torch.ones if started else torch.zeros  # (steps already taken densely: every row may carry state)
            st["row_touched"] = fill(param.shape[0], dtype=torch.uint8, device=param.device)
            self._fast = {}
        from . import ops

        ops.mark_rows(st["row_touched"], ids)
        self._marked.add(id(param))
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._steps = {}
        self._fast = {}
        for st in self.state.values():  # (torch casts every state tensor to the parameter's dtype on load)
            if "row_touched" in st:
                st["row_touched"] = (st["row_touched"] != 0).to(torch.uint8)

    def state_dict(self):
        sd = super().state_dict()
        # The group's entries share ONE step tensor here; torch's format has one each.  The per-parameter dicts torch
        # hands back ARE this optimizer's own state dicts: build new ones -- writing the clone into them cut the live
        # state off the shared counter, and every later state_dict() (a checkpoint) reported the step count of the
        # first call (found by tools/two_rank_check.py: a resumed optimizer took its next step with a stale bias correction).
        sd["state"] = {k: (dict(st, step=st["step"].clone()) if "step" in st else st) for k, st in sd["state"].items()}
        return sd

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        fast = self.__dict__.setdefault("_fast", {})
        for gi, group in enumerate(self.param_groups):
            params = group["params"]
            # steady state: every parameter has a gradient and state, all at the same step -> the cached table (p, m, v
            # pointers and sizes are fixed) only needs this step's gradient pointers
            c = fast.get(gi)
            if c is not None and c["n"] == len(params) and all(p.grad is not None for p in params) and \
                    all(pp == p.data_ptr() for pp, p in zip(c["pp"], params)):
                arr = c["arr"]
                keep = []
                for i, p in enumerate(params):
                    g = p.grad
                    if not g.is_contiguous() or g.is_sparse:
                        c = None
                        break
                    arr[i].g = g.data_ptr()
                if c is not None and c["masked"] and not all(i in self._marked for i in c["masked"]):
                    c = None  # a masked table was not announced for this step: the slow path below turns it dense
                if c is not None:
                    c["t"] += 1
                    c["step"] += 1  # ONE CPU tensor shared by the group's state entries (37 separate ones cost 75 us)
                    for p in params:
                        self._steps[id(p)] = c["t"]
                    torch.autograd.graph.increment_version(params)
                    b1, b2 = group["betas"]
                    _lib.check(lib.carca_adam_step(arr, len(params), float(group["lr"]), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]), c["t"], _stream()),
                               "adam_step")
                    continue
            fast.pop(gi, None)
            live = [p for p in params if p.grad is not None]
            if not live:
                continue
            steps = set()
            arr = (_lib.AdamTensor * len(live))()
            keep = []
            masked = []
            for i, p in enumerate(live):
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise CarcaHipError("carca Adam: parameters must be contiguous fp32 CUDA tensors with dense gradients")
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                st = self.state[p]
                if "step" not in st:  # (mark_rows may have put the row mask there already)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t = self._steps.get(id(p))
                if t is None:
                    t = int(st["step"])
                self._steps[id(p)] = t + 1
                steps.add(t + 1)
                a = arr[i]
                a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), \
                    p.numel()
                if "row_touched" in st:
                    if id(p) not in self._marked:  # an unannounced step: rows unknown -> every row counts as touched
                        st["row_touched"].fill_(1)
                    a.row_mask, a.row_len = st["row_touched"].data_ptr(), p.shape[1]
                    masked.append(id(p))
            # the kernel writes the parameters behind torch's back: bump their version counters, which is what the
            # modules' packed-weight caches (and autograd's saved-tensor checks) go by
            torch.autograd.graph.increment_version(live)
            b1, b2 = group["betas"]
            if len(steps) == 1:
                t = steps.pop()
                _lib.check(lib.carca_adam_step(arr, len(live), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                               float(group["weight_decay"]), t, _stream()), "adam_step")
                shared = torch.tensor(float(t), dtype=torch.float32)
                for p in live:
                    self.state[p]["step"] = shared
                if len(live) == len(params) and not keep:
                    fast[gi] = dict(arr=arr, n=len(params), t=t, pp=[p.data_ptr() for p in params], step=shared,
                                    masked=masked)
            else:  # parameters that joined later carry their own step count: one launch per count
                for p in live:
                    self.state[p]["step"] = torch.tensor(float(self._steps[id(p)]), dtype=torch.float32)
                for t in sorted(steps):
                    idx = [i for i, p in enumerate(live) if self._steps[id(p)] == t]
                    sub = (_lib.AdamTensor * len(idx))(*[arr[i] for i in idx])
                    _lib.check(lib.carca_adam_step(sub, len(idx), float(group["lr"]), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]), t, _stream()),
                               "adam_step")
        self._marked.clear()
        return loss
