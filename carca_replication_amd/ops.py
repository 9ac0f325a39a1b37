"""Torch-tensor front end of the C ABI (include/carca_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every function below checks its
tensors, hands raw device pointers to libcarca_hip.so and returns torch tensors that alias the
buffers the kernels wrote.  Nothing in this file computes model math on the host or in ATen.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import CarcaHipError


# bench.py brackets single kernels with events on the launch stream: {"feat": (e0, e1), "cross": (e0, e1)}
_stage_events = None


def set_stage_events(ev) -> None:
    global _stage_events
    _stage_events = ev


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise CarcaHipError(
                "the CARCA HIP path needs CUDA/ROCm tensors (got a CPU tensor); there is no CPU implementation "
                "in this package -- the CPU oracle lives under oracle/ and is test infrastructure only")


def _f32(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        raise CarcaHipError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ids32(t: Tensor) -> Tensor:
    t = t if t.dtype == torch.int32 else t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()


def padded_dims(d: int, H: int) -> Tuple[int, int, int]:
    lib = _lib.load()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.carca_padded_dims(d, H, C.byref(a), C.byref(b), C.byref(c)), "padded_dims")
    return a.value, b.value, c.value


# --------------------------------------------------------------------------------------------------
# weight packing
# --------------------------------------------------------------------------------------------------
@dataclass
class PackItem:
    src: Tensor  # 1-D or 2-D fp32 parameter (a [d,d,1] conv weight is passed as its [d,d] view)
    dst_rows: int
    dst_cols: int
    row_heads: Tuple[int, int] = (0, 0)  # (dh, dhp) when rows are output features split into heads
    col_heads: Tuple[int, int] = (0, 0)


class PackedWeights:
    """One device buffer holding every packed matrix, plus the offsets of each item."""

    def __init__(self, items: Sequence[PackItem], device):
        self.items = list(items)
        sizes = [((it.dst_rows * it.dst_cols + 3) // 4) * 4 for it in self.items]
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        self.buf = torch.empty(max(self.offsets[-1], 4), dtype=torch.float32, device=device)
        self._descs = (_lib.PackDesc * len(self.items))()

    def ptr(self, i: int) -> int:
        return self.buf.data_ptr() + 4 * self.offsets[i]

    def view(self, i: int) -> Tensor:
        it = self.items[i]
        return self.buf[self.offsets[i]: self.offsets[i] + it.dst_rows * it.dst_cols].view(it.dst_rows, it.dst_cols)

    def pack(self) -> None:
        lib = _lib.load()
        keep = []
        for i, it in enumerate(self.items):
            src = it.src.detach()
            src2 = src.reshape(1, -1) if src.dim() == 1 else src.reshape(src.shape[0], -1)
            src2 = _f32(src2)
            _need_cuda(src2)
            keep.append(src2)
            d = self._descs[i]
            d.src, d.dst = src2.data_ptr(), self.ptr(i)
            d.rows, d.cols, d.src_ld = src2.shape[0], src2.shape[1], src2.stride(0)
            d.dst_rows, d.dst_cols = it.dst_rows, it.dst_cols
            d.row_dh, d.row_dhp = it.row_heads
            d.col_dh, d.col_dhp = it.col_heads
        _lib.check(lib.carca_pack_weights(self._descs, len(self.items), _stream()), "pack_weights")


# --------------------------------------------------------------------------------------------------
# embed
# --------------------------------------------------------------------------------------------------
def embed_fwd(segs: Sequence[Tuple[Tensor, Tensor, Tensor, bool]], items_w: Tensor, feats_w: Tensor, feats_b: Tensor,
              joint_w: Tensor, joint_b: Tensor, pos: Optional[Tensor], ld_e: int) -> Tuple[List[Tensor], Tensor]:
    """AllEmbedding over several (ids [B,T], attrs [B,T,A], ctx [B,T,Cx], add_pos) segments in one call.

    Returns ([e_s [B,T,ld_e]], zq [sum B*T, d+g]); e_s[..., d:] is zero.
    """
    lib = _lib.load()
    if not 1 <= len(segs) <= _lib.MAX_SEGS:
        raise CarcaHipError(f"embed_fwd takes 1..{_lib.MAX_SEGS} segments, got {len(segs)}")
    d, g = items_w.shape[1], feats_w.shape[0]
    n_attrs = segs[0][1].shape[-1]
    n_ctx = segs[0][2].shape[-1] if segs[0][2] is not None else 0
    if feats_w.shape[1] != n_attrs + n_ctx or joint_w.shape != (d, d + g):
        raise CarcaHipError("embed_fwd: weight shapes do not match the inputs")
    items_w, feats_w, feats_b, joint_w, joint_b = map(_f32, (items_w.detach(), feats_w.detach(), feats_b.detach(),
                                                            joint_w.detach(), joint_b.detach()))
    _need_cuda(items_w, feats_w, feats_b, joint_w, joint_b)
    dev = items_w.device
    arr = (_lib.RowSeg * len(segs))()
    keep, outs, total = [], [], 0
    for i, (ids, attrs, ctx, add_pos) in enumerate(segs):
        _need_cuda(ids, attrs, ctx)
        B, T = ids.shape
        ids32, attrs, ctx = _ids32(ids), _f32(attrs), _f32(ctx)
        if attrs.shape != (B, T, n_attrs) or ctx.shape != (B, T, n_ctx):
            raise CarcaHipError("embed_fwd: attrs/ctx shapes do not match ids")
        e = torch.empty(B, T, ld_e, dtype=torch.float32, device=dev)
        keep += [ids32, attrs, ctx]
        outs.append(e)
        a = arr[i]
        a.ids, a.attrs, a.ctx, a.e_out = ids32.data_ptr(), attrs.data_ptr(), ctx.data_ptr(), e.data_ptr()
        a.rows, a.T, a.add_pos = B * T, T, int(bool(add_pos))
        total += B * T
    zq = torch.empty(total, d + g, dtype=torch.float32, device=dev)
    pos_ptr = None
    if pos is not None:
        pos = _f32(pos.detach())
        _need_cuda(pos)
        if pos.dim() != 2 or pos.shape[1] != d:
            raise CarcaHipError("embed_fwd: pos must be [T, d]")
        pos_ptr = pos.data_ptr()
    def call(stages):
        _lib.check(lib.carca_embed_fwd(arr, len(segs), n_attrs, n_ctx, d, g, items_w.data_ptr(), feats_w.data_ptr(),
                                       feats_b.data_ptr(), joint_w.data_ptr(), joint_b.data_ptr(), pos_ptr,
                                       zq.data_ptr(), ld_e, stages, _stream()), "embed_fwd")

    ev = _stage_events.get("feat") if _stage_events else None
    if ev is None:
        call(7)
    else:  # same three launches, with events around the feature GEMM
        call(1)
        ev[0].record()
        call(2)
        ev[1].record()
        call(4)
    return outs, zq


# --------------------------------------------------------------------------------------------------
# self-attention block
# --------------------------------------------------------------------------------------------------
def sa_block_fwd(x: Tensor, ids: Tensor, w: "_lib.SaWeights", d: int, H: int, residual: bool) -> Tensor:
    """x [B, L, ldx] (ldx >= d) -> y [B, L, DPI]; `ids` [B, L] (any integer/bool type, 0 = pad)."""
    lib = _lib.load()
    _need_cuda(x, ids)
    x = _f32(x)
    B, L, ldx = x.shape
    dpi, _, _ = padded_dims(d, H)
    ids32 = _ids32(ids)
    y = torch.empty(B, L, dpi, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_sa_block_fwd(x.data_ptr(), ldx, ids32.data_ptr(), y.data_ptr(), dpi, B, L, d, H, C.byref(w),
                                      int(bool(residual)), _stream()), "sa_block_fwd")
    return y


# --------------------------------------------------------------------------------------------------
# final norm + grouped cross-attention scoring
# --------------------------------------------------------------------------------------------------
def cross_score_fwd(p_raw: Tensor, p_ids: Tensor, groups: Sequence[Tuple[Tensor, Tensor]], w: "_lib.CaWeights", d: int,
                    H: int, residual: bool, training: bool, want_normed: bool = False):
    """p_raw [B, L, ldp]; groups: [(o [B,N,ldo], ids [B,N])] -> ([y [B,N]], p_normed or None)."""
    lib = _lib.load()
    _need_cuda(p_raw, p_ids)
    p_raw = _f32(p_raw)
    B, L, ldp = p_raw.shape
    if not 1 <= len(groups) <= _lib.MAX_GROUPS:
        raise CarcaHipError(f"cross_score_fwd takes 1..{_lib.MAX_GROUPS} target groups, got {len(groups)}")
    p_ids32 = _ids32(p_ids)
    arr = (_lib.TargetGroup * len(groups))()
    keep, ys, ldo = [], [], None
    for i, (o, ids) in enumerate(groups):
        _need_cuda(o, ids)
        o, ids32 = _f32(o), _ids32(ids)
        if o.shape[0] != B or o.shape[:2] != ids32.shape:
            raise CarcaHipError("cross_score_fwd: group shapes do not match")
        if ldo is None:
            ldo = o.shape[2]
        elif ldo != o.shape[2]:
            raise CarcaHipError("cross_score_fwd: all groups must share one row stride")
        y = torch.empty(B, o.shape[1], dtype=torch.float32, device=o.device)
        keep += [o, ids32]
        ys.append(y)
        arr[i].o, arr[i].ids, arr[i].y, arr[i].N = o.data_ptr(), ids32.data_ptr(), y.data_ptr(), o.shape[1]
    p_normed = torch.empty_like(p_raw) if want_normed else None
    ev = _stage_events.get("cross") if _stage_events else None
    if ev is not None:
        ev[0].record()
    _lib.check(lib.carca_cross_score_fwd(p_raw.data_ptr(), ldp, p_ids32.data_ptr(),
                                         p_normed.data_ptr() if want_normed else None, arr, len(groups), ldo, B, L, d,
                                         H, C.byref(w), int(bool(residual)), int(bool(training)), _stream()),
               "cross_score_fwd")
    if ev is not None:
        ev[1].record()
    return ys, p_normed


# --------------------------------------------------------------------------------------------------
# loss and metrics
# --------------------------------------------------------------------------------------------------
def bce_fwd(y: Tensor, y_true: Tensor, ids: Tensor, eps: float = 1e-8, want_grad: bool = False):
    lib = _lib.load()
    _need_cuda(y, y_true, ids)
    y = _f32(y)
    yt, ids32 = _ids32(y_true), _ids32(ids)
    if y.numel() != yt.numel() or y.numel() != ids32.numel():
        raise CarcaHipError("bce_fwd: y_pred, y_true and mask ids must have the same number of elements")
    scratch = torch.empty(2, dtype=torch.float32, device=y.device)
    loss = torch.empty(1, dtype=torch.float32, device=y.device)
    dy = torch.empty_like(y) if want_grad else None
    _lib.check(lib.carca_bce_fwd(y.data_ptr(), yt.data_ptr(), ids32.data_ptr(), y.numel(), eps, scratch.data_ptr(),
                                 loss.data_ptr(), dy.data_ptr() if want_grad else None, _stream()), "bce_fwd")
    return loss[0], dy


def rank_metrics(y: Tensor, k: int, sums: Optional[Tensor] = None, want_rank: bool = False):
    """Accumulates [HR@k sum, NDCG@k sum, ties] into `sums` (device float[3]) for y [B, N], positive in column 0."""
    lib = _lib.load()
    _need_cuda(y)
    y = _f32(y)
    B, N = y.shape
    if sums is None:
        sums = torch.zeros(3, dtype=torch.float32, device=y.device)
    rank = torch.empty(B, dtype=torch.int32, device=y.device) if want_rank else None
    _lib.check(lib.carca_rank_metrics(y.data_ptr(), B, N, k, rank.data_ptr() if want_rank else None, sums.data_ptr(),
                                      _stream()), "rank_metrics")
    return sums, rank
