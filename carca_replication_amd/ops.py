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


# raw hipEvent_t handles [feat0, feat1, cross0, cross1(, sa0, sa1, joint0, joint1)] for the one-call forward (bench.py),
# or None; entries may be None (CarcaForwardDesc.n_events)
_fused_events = None


def set_fused_events(handles) -> None:
    global _fused_events
    _fused_events = handles


def fused_events():
    return _fused_events


class HipEvent:
    """A hipEvent_t owned through the C ABI (no torch.cuda.Event: those cannot be handed to carca_forward)."""

    def __init__(self):
        h = C.c_void_p()
        _lib.check(_lib.load().carca_event_create(C.byref(h)), "event_create")
        self.handle = h.value

    def elapsed_ms(self, stop: "HipEvent") -> float:
        ms = C.c_float()
        _lib.check(_lib.load().carca_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)), "event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            _lib.load().carca_event_destroy(self.handle)
        except Exception:
            pass


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() builds a Stream object through several layers
    of Python (~10 us; 40+ launches per training step); the raw accessors are a pair of C calls."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
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


def _btk_view(t: Tensor) -> Tuple[Tensor, int]:
    """[B, T, K] fp32 tensor -> (tensor the kernels can walk, user stride in elements or 0 when dense).

    A view whose rows are dense (stride(2) == 1, stride(1) == K) but whose users are further apart -- e.g.
    o_a[:, :L] of train.py:86-88 -- is passed through without a copy; anything else is made contiguous."""
    if t.dtype != torch.float32:
        raise CarcaHipError(f"expected float32, got {t.dtype}")
    B, T, K = t.shape
    if K > 0 and t.stride(2) == 1 and t.stride(1) == K and (B == 1 or t.stride(0) >= T * K):
        bs = t.stride(0) if B > 1 else T * K
        return t, (0 if bs == T * K else bs)
    return t.contiguous(), 0


def padded_dims(d: int, H: int) -> Tuple[int, int, int]:
    lib = _lib.load()
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.carca_padded_dims(d, H, C.byref(a), C.byref(b), C.byref(c)), "padded_dims")
    return a.value, b.value, c.value


FUSED_MAX_D = 128  # the per-user kernels keep rows of at most this many floats (carca_padded_dims); wider models run composed


def row_ld(d: int) -> int:
    """Row stride of the [rows, ld] activations between kernels: the fused kernels' padded width up to FUSED_MAX_D, the
    next multiple of four floats beyond (long_profile.py's composed path takes any d the reference takes)."""
    return padded_dims(d, 1)[0] if d <= FUSED_MAX_D else (d + 3) // 4 * 4


# --------------------------------------------------------------------------------------------------
# dropout plumbing
# --------------------------------------------------------------------------------------------------
def new_dropout_seed() -> int:
    """A fresh 62-bit seed from torch's CPU generator (so torch.manual_seed makes training runs repeatable)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


_SEED_OFFSET: Optional[Tensor] = None  # device int64[1] added to every dropout seed inside the kernels, or None


def set_dropout_seed_offset(counter: Optional[Tensor]) -> None:
    """While set, every dropout site's kernel adds counter[0] (device, int64) to its seed when it starts.  A train step
    captured into a hipGraph repeats its launch arguments -- the seeds among them -- so the graph increments this
    counter per replay instead (engine.GraphedTrainStep); None (default) = seeds are used as passed."""
    global _SEED_OFFSET
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
        raise CarcaHipError("set_dropout_seed_offset: need a CUDA int64 tensor of one element")
    _SEED_OFFSET = counter


def dropout_seed_offset_ptr() -> Optional[int]:
    return _SEED_OFFSET.data_ptr() if _SEED_OFFSET is not None else None


def _drop_struct(p: float, seed: int, site: int):
    if not p:
        return None
    d = _lib.Dropout()
    d.p, d.seed, d.site = float(p), int(seed), int(site)
    d.seed_offset = dropout_seed_offset_ptr()
    return d


def dropout_fwd(x: Tensor, cols: int, p: float, seed: int, site: int) -> Tensor:
    """In-place dropout of x [..., ld] over its first `cols` columns; returns the uint8 keep-mask [rows, cols]."""
    lib = _lib.load()
    _need_cuda(x)
    x2 = x.view(-1, x.shape[-1])
    mask = torch.empty(x2.shape[0], cols, dtype=torch.uint8, device=x.device)
    d = _drop_struct(p, seed, site)
    _lib.check(lib.carca_dropout_fwd(x2.data_ptr(), x2.shape[0], cols, x2.stride(0), C.byref(d), mask.data_ptr(),
                                     _stream()), "dropout_fwd")
    return mask


def mask_mul(x: Tensor, mask: Tensor, scale: float, cols: int, out_ld: int) -> Tensor:
    """out [rows, out_ld] = x[:, :cols] * mask[:, :cols] * scale (pad columns zero)."""
    lib = _lib.load()
    x = _row2d(x, "x")
    out = torch.empty(x.shape[0], out_ld, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_mask_mul(x.data_ptr(), x.stride(0), mask.data_ptr(), mask.stride(0), scale, out.data_ptr(),
                                  out_ld, x.shape[0], cols, out_ld, _stream()), "mask_mul")
    return out


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
    transposed: bool = False  # pack src^T (dst_rows/dst_cols and the head maps refer to the transposed matrix)
    frag16: bool = False      # MFMA-fragment order (CarcaPackDesc.frag16): the attention kernels' operand layout
    fold_vec: Optional[Tensor] = None  # CarcaPackDesc.fold_vec: contract src's rows per head with this [rows] vector ...
    fold_H: int = 0                    # ... into a logical [fold_H, cols] source


class PackedWeights:
    """One device buffer holding every packed matrix, plus the offsets of each item."""

    def __init__(self, items: Sequence[PackItem], device, buf: Optional[Tensor] = None):
        """buf: optional caller-provided storage of at least size_of(items) floats (e.g. a zeroed slice of a bigger
        buffer, so that several staging areas share one fill launch)."""
        self.items = list(items)
        self.offsets = [0]
        for it in self.items:
            self.offsets.append(self.offsets[-1] + ((it.dst_rows * it.dst_cols + 3) // 4) * 4)
        need = max(self.offsets[-1], 4)
        if buf is None:
            buf = torch.empty(need, dtype=torch.float32, device=device)
        elif buf.numel() < need or buf.dtype != torch.float32 or not buf.is_contiguous():
            raise CarcaHipError("PackedWeights: provided buffer too small / not contiguous fp32")
        self.buf = buf
        self._descs = (_lib.PackDesc * len(self.items))()

    @staticmethod
    def size_of(items: Sequence[PackItem]) -> int:
        return max(sum(((it.dst_rows * it.dst_cols + 3) // 4) * 4 for it in items), 4)

    def ptr(self, i: int) -> int:
        return self.buf.data_ptr() + 4 * self.offsets[i]

    def view(self, i: int) -> Tensor:
        it = self.items[i]
        if it.frag16:
            raise CarcaHipError("PackedWeights.view: item is in fragment order, not a row-major matrix")
        return self.buf[self.offsets[i]: self.offsets[i] + it.dst_rows * it.dst_cols].view(it.dst_rows, it.dst_cols)

    def _fill(self, tensors) -> list:
        """Point the descriptors at `tensors` (the items' sources, or gradient tensors of the same shapes).  Everything
        that does not depend on the tensors is written once (this runs for ~70 descriptors per train step)."""
        keep, extra = [], []
        base = self.buf.data_ptr()
        first = not self.__dict__.get("_static_done")
        # steady state (the same parameters repacked into the same buffer every training step): nothing to rewrite
        sig = (base,) + tuple((t.data_ptr(), t.shape, t.stride()) for t in tensors) + \
            tuple(it.fold_vec.data_ptr() for it in self.items if it.fold_vec is not None)
        if not first and self.__dict__.get("_filled_sig") == sig:
            return list(tensors)
        self._filled_sig = None
        for i, (it, src) in enumerate(zip(self.items, tensors)):
            if src.requires_grad:
                src = src.detach()
            if src.dim() != 2:
                src = src.reshape(1, -1) if src.dim() == 1 else src.reshape(src.shape[0], -1)
            if src.dtype != torch.float32 or src.stride(1) != 1 or not src.is_cuda:
                raise CarcaHipError("pack: parameters must be fp32 CUDA tensors with unit inner stride")
            keep.append(src)
            d = self._descs[i]
            d.src, d.dst, d.src_ld = src.data_ptr(), base + 4 * self.offsets[i], src.stride(0)
            if it.fold_vec is not None:
                fv = it.fold_vec.detach()
                if fv.dtype != torch.float32 or not fv.is_cuda or not fv.is_contiguous() or fv.numel() != src.shape[0]:
                    raise CarcaHipError("pack: fold_vec must be a contiguous fp32 CUDA vector with one entry per source row")
                extra.append(fv)
                d.fold_vec, d.fold_H = fv.data_ptr(), it.fold_H
            r, c = (src.shape[1], src.shape[0]) if it.transposed else (src.shape[0], src.shape[1])
            if first:
                d.rows, d.cols = r, c
                d.dst_rows, d.dst_cols = it.dst_rows, it.dst_cols
                d.row_dh, d.row_dhp = it.row_heads
                d.col_dh, d.col_dhp = it.col_heads
                d.transposed = int(it.transposed)
                d.frag16 = int(it.frag16)
            elif d.rows != r or d.cols != c:
                raise CarcaHipError("pack: tensor shape differs from the item this descriptor was built for")
        self._static_done = True
        if all(k.data_ptr() == t.data_ptr() for k, t in zip(keep, tensors)):  # (detach / reshape stand-ins share the memory)
            self._filled_sig = sig
        return keep + extra

    def pack(self) -> None:
        pack_many([self])

    def unpack_into(self, grads, accumulate: bool = False) -> None:
        """Inverse map for gradients: grads[i] (real shape of items[i].src) (+)= packed item i of this buffer."""
        lib = _lib.load()
        keep = self._fill(grads)
        _lib.check(lib.carca_unpack_grads(self._descs, len(self.items), int(accumulate), _stream()), "unpack_grads")
        del keep


def pack_many(pws: Sequence[PackedWeights]) -> None:
    """(Re)fill several PackedWeights buffers with ONE carca_pack_weights call (the modules of a model repack together
    at every training step)."""
    if not pws:
        return
    keep = [pw._fill([it.src for it in pw.items]) for pw in pws]
    n = sum(len(pw.items) for pw in pws)
    if len(pws) == 1:
        arr = pws[0]._descs
    else:
        arr = (_lib.PackDesc * n)()
        i = 0
        for pw in pws:
            for d in pw._descs:
                C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(_lib.PackDesc))
                i += 1
    _lib.check(_lib.load().carca_pack_weights(arr, n, _stream()), "pack_weights")
    del keep


# --------------------------------------------------------------------------------------------------
# embed
# --------------------------------------------------------------------------------------------------
def embed_fwd(segs: Sequence[Tuple[Tensor, Tensor, Tensor, bool]], items_w: Tensor, feats_w: Tensor, feats_b: Tensor,
              joint_w: Tensor, joint_b: Tensor, pos: Optional[Tensor], ld_e: int,
              attrs_table: Optional[Tensor] = None) -> Tuple[List[Tensor], Tensor]:
    """AllEmbedding over several (ids [B,T], attrs [B,T,A], ctx [B,T,Cx], add_pos) segments in one call.

    A segment's attrs may be None when `attrs_table` [n_items, A] is given: its rows are then gathered by item id
    inside the GEMM's operand load.  Returns ([e_s [B,T,ld_e]], zq [sum B*T, d+g]); e_s[..., d:] is zero.
    """
    lib = _lib.load()
    if not 1 <= len(segs) <= _lib.MAX_SEGS:
        raise CarcaHipError(f"embed_fwd takes 1..{_lib.MAX_SEGS} segments, got {len(segs)}")
    d, g = items_w.shape[1], feats_w.shape[0]
    if attrs_table is not None:
        attrs_table = _f32(attrs_table)
        _need_cuda(attrs_table)
    n_attrs = attrs_table.shape[1] if attrs_table is not None else segs[0][1].shape[-1]
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
        if attrs is None and attrs_table is None:
            raise CarcaHipError("embed_fwd: attrs is None and no attribute table is registered")
        if (attrs is not None and attrs.shape != (B, T, n_attrs)) or ctx.shape != (B, T, n_ctx):
            raise CarcaHipError("embed_fwd: attrs/ctx shapes do not match ids")
        ids32 = _ids32(ids)
        attrs, a_bs = _btk_view(attrs) if attrs is not None else (None, 0)
        ctx, c_bs = _btk_view(ctx) if n_ctx > 0 else (ctx, 0)
        e = torch.empty(B, T, ld_e, dtype=torch.float32, device=dev)
        keep += [ids32, attrs, ctx]
        outs.append(e)
        a = arr[i]
        a.ids, a.ctx, a.e_out = ids32.data_ptr(), ctx.data_ptr(), e.data_ptr()
        if attrs is not None:
            a.attrs = attrs.data_ptr()
        else:
            a.attrs_table = attrs_table.data_ptr()
            a.attrs_table_rows = attrs_table.shape[0]
        a.rows, a.T, a.add_pos = B * T, T, int(bool(add_pos))
        a.attrs_bstride, a.ctx_bstride = a_bs, c_bs
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
def sa_block_fwd(x: Tensor, ids: Tensor, w: "_lib.SaWeights", d: int, H: int, residual: bool, save: bool = False,
                 drop: Optional[Tuple[float, int, int]] = None, pads_uniform: bool = False):
    """x [B, L, ldx] (ldx >= d) -> y [B, L, DPI]; `ids` [B, L] (any integer/bool type, 0 = pad).

    save=True also returns the dict of tensors the backward pass needs (CarcaSaSave).
    pads_uniform=True (eval only): the caller's promise that each user's leading pad rows of x are equal
    (carca_sa_block_eval): one of them is computed and written to all of them."""
    lib = _lib.load()
    _need_cuda(x, ids)
    x = _f32(x)
    B, L, ldx = x.shape
    dpi, _, dpo = padded_dims(d, H)
    ids32 = _ids32(ids)
    y = torch.empty(B, L, dpi, dtype=torch.float32, device=x.device)
    sv, saved = None, None
    if save:
        sv = _lib.SaSave()
        mk = lambda w_: torch.empty(B * L, w_, dtype=torch.float32, device=x.device)  # noqa: E731
        saved = dict(qn=mk(dpi), qh=mk(dpo), kh=mk(dpo), vh=mk(dpo), r=mk(dpi), s2=mk(dpi), h1=mk(dpi))
        if drop and drop[0] > 0:
            u8 = lambda *sh: torch.empty(*sh, dtype=torch.uint8, device=x.device)  # noqa: E731
            saved.update(m_attn=u8(B, H, L, L), m_ffn1=u8(B * L, dpi), m_ffn2=u8(B * L, dpi))
        for k, t in saved.items():
            setattr(sv, k, t.data_ptr())
    dstruct = _drop_struct(*drop) if drop else None
    if pads_uniform:
        if save or dstruct is not None:
            raise CarcaHipError("sa_block_fwd: pads_uniform is an eval-mode promise (no saved tensors, no dropout)")
        _lib.check(lib.carca_sa_block_eval(x.data_ptr(), ldx, ids32.data_ptr(), y.data_ptr(), dpi, B, L, d, H, C.byref(w),
                                           int(bool(residual)), 1, _stream()), "sa_block_eval")
        return y
    _lib.check(lib.carca_sa_block_fwd(x.data_ptr(), ldx, ids32.data_ptr(), y.data_ptr(), dpi, B, L, d, H, C.byref(w),
                                      int(bool(residual)), C.byref(sv) if save else None,
                                      C.byref(dstruct) if dstruct is not None else None, _stream()), "sa_block_fwd")
    return (y, saved) if save else y


# --------------------------------------------------------------------------------------------------
# final norm + grouped cross-attention scoring
# --------------------------------------------------------------------------------------------------
def cross_score_fwd(p_raw: Tensor, p_ids: Tensor, groups: Sequence[Tuple[Tensor, Tensor]], w: "_lib.CaWeights", d: int,
                    H: int, residual: bool, training: bool, want_normed: bool = False, save: bool = False,
                    drop: Optional[Tuple[float, int, int]] = None):
    """p_raw [B, L, ldp]; groups: [(o [B,N,ldo], ids [B,N])] -> ([y [B,N]], p_normed or None[, saved])."""
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
    p_normed = torch.empty_like(p_raw) if (want_normed or save) else None
    sv, saved, dstruct = None, None, None
    if save:
        _, _, dpo = padded_dims(d, H)
        sv = _lib.CaSave()
        mk = lambda n: torch.empty(n, dpo, dtype=torch.float32, device=p_raw.device)  # noqa: E731
        saved = dict(kh=mk(B * L), vh=mk(B * L), qh=[mk(B * o.shape[1]) for (o, _) in groups])
        sv.kh, sv.vh = saved["kh"].data_ptr(), saved["vh"].data_ptr()
        for i, t in enumerate(saved["qh"]):
            sv.qh[i] = t.data_ptr()
        if drop and drop[0] > 0:
            saved["m_attn"] = [torch.empty(B, H, o.shape[1], L, dtype=torch.uint8, device=p_raw.device)
                               for (o, _) in groups]
            for i, t in enumerate(saved["m_attn"]):
                sv.m_attn[i] = t.data_ptr()
    dstruct = _drop_struct(*drop) if drop else None
    ev = _stage_events.get("cross") if _stage_events else None
    if ev is not None:
        ev[0].record()
    _lib.check(lib.carca_cross_score_fwd(p_raw.data_ptr(), ldp, p_ids32.data_ptr(),
                                         p_normed.data_ptr() if p_normed is not None else None, arr, len(groups),
                                         ldo, B, L, d, H, C.byref(w), int(bool(residual)), int(bool(training)),
                                         C.byref(sv) if save else None,
                                         C.byref(dstruct) if (save and dstruct is not None) else None, _stream()),
               "cross_score_fwd")
    if ev is not None:
        ev[1].record()
    if save:
        return ys, p_normed, saved
    return ys, p_normed


# --------------------------------------------------------------------------------------------------
# loss and metrics
# --------------------------------------------------------------------------------------------------
def bce_fwd(y: Tensor, y_true: Tensor, ids: Tensor, eps: float = 1e-8, want_grad: bool = False,
            denom: Optional[Tensor] = None):
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
                                 loss.data_ptr(), dy.data_ptr() if want_grad else None, _ptr(denom), _stream()),
               "bce_fwd")
    return loss[0], dy


EVAL_METRICS_MAX = 16384  # elements of one batch that carca_eval_metrics takes


def eval_metrics(y: Tensor, y_true: Tensor, ids: Tensor, k: int, sums: Tensor, eps: float = 1e-8) -> None:
    """sums [5] += [HR@k, NDCG@k, ties, masked-mean BCE loss, users] of one evaluation batch y [B, N] (positive in column
    0), one launch (train.py:45-51)."""
    _need_cuda(y, y_true, ids, sums)
    y = _f32(y)
    B, N = y.shape
    yt, ids32 = _ids32(y_true), _ids32(ids)
    if sums.dtype != torch.float32 or sums.numel() < 5 or not sums.is_contiguous() or yt.numel() != y.numel() or \
            ids32.numel() != y.numel():
        raise CarcaHipError("eval_metrics: sums must be float32[5], y_true / ids of y's shape")
    _lib.check(_lib.load().carca_eval_metrics(y.data_ptr(), yt.data_ptr(), ids32.data_ptr(), B, N, k, eps, sums.data_ptr(),
                                              _stream()), "eval_metrics")


def rank_metrics(y: Tensor, k: int, sums: Optional[Tensor] = None, want_rank: bool = False,
                 pos: Optional[Tensor] = None):
    """Accumulates [HR@k sum, NDCG@k sum, ties] into `sums` (device float[3]) for y [B, N]; the positive sits in
    column pos[u] (default: column 0)."""
    lib = _lib.load()
    _need_cuda(y)
    y = _f32(y)
    B, N = y.shape
    if sums is None:
        sums = torch.zeros(3, dtype=torch.float32, device=y.device)
    rank = torch.empty(B, dtype=torch.int32, device=y.device) if want_rank else None
    pos32 = _ids32(pos.reshape(-1)) if pos is not None else None
    _lib.check(lib.carca_rank_metrics(y.data_ptr(), B, N, k, _ptr(pos32), rank.data_ptr() if want_rank else None,
                                      sums.data_ptr(), _stream()), "rank_metrics")
    return sums, rank


# --------------------------------------------------------------------------------------------------
# dense building blocks (used by the backward pass; the forward reaches gemm_rows through embed_fwd)
# --------------------------------------------------------------------------------------------------
def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def gemm_rows(segs, bt0: Tensor, N: int, K0: int, out_ld: int, **kw) -> List[Tensor]:
    """One row GEMM (arguments: _gemm_desc); returns the output tensors of its segments."""
    D, outs, _keep = _gemm_desc(segs, bt0, N, K0, out_ld, **kw)
    _lib.check(_lib.load().carca_gemm_rows(C.byref(D), _stream()), "gemm_rows")
    return outs


def gemm_rows_group(calls) -> List[List[Tensor]]:
    """Several INDEPENDENT row GEMMs in one call (carca_gemm_rows_group: narrow products share a launch).
    calls: [dict(segs=..., bt0=..., N=..., K0=..., out_ld=..., **gemm_rows keywords)]; returns each product's outputs."""
    built = [_gemm_desc(**c) for c in calls]
    arr = (_lib.GemmDesc * len(built))(*[b[0] for b in built])
    _lib.check(_lib.load().carca_gemm_rows_group(arr, len(built), _stream()), "gemm_rows_group")
    return [b[1] for b in built]


def _gemm_desc(segs, bt0: Tensor, N: int, K0: int, out_ld: int, *, bt1: Optional[Tensor] = None, K1: int = 0,
               bias: Optional[Tensor] = None, pos: Optional[Tensor] = None, colvec: Optional[Tensor] = None,
               gate_slope: float = 0.01, mask_rows: bool = False, ncols_out: Optional[int] = None,
               gate_scale: float = 1.0, gate_zero_drops: bool = False, alpha: float = 1.0,
               add_table: Optional[Tensor] = None):
    """C_s[m][n] = sum_k A_s[m][k] Bt[n][k] (+ epilogue) for every row segment s: builds the descriptor.

    segs: list of dicts with keys a0 [rows, lda0] and optionally a1, ids, add, gate, rowscale, T, add_pos, out.
    All 2-D operands are row-major views whose LAST stride is 1; their row stride is taken from .stride(0).
    a0 / a1 may also be [B, T, K] views whose users are strided (o_a[:, :L] of train.py:86-88): walked in place.
    a0_gather=True: a0 is a TABLE [n, lda0] and row r reads a0[ids[r]] (then `rows` must be given by ids).
    Returns (descriptor, output tensors [rows, out_ld] -- allocated here unless the segment brings 'out' --, keep-alive).
    """
    if not 1 <= len(segs) <= _lib.MAX_SEGS:
        raise CarcaHipError("gemm_rows: 1..4 segments")
    D = _lib.GemmDesc()
    D.nseg = len(segs)
    outs, keep = [], []

    def ld_of(t, name):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
            raise CarcaHipError(f"gemm_rows: {name} must be a 2-D fp32 view with unit inner stride")
        _need_cuda(t)
        return t.stride(0)

    def operand(t, name):  # -> (tensor, row stride, rows, T or None, user stride)
        if t.dim() == 3:
            _need_cuda(t)
            t, bs = _btk_view(t)
            keep.append(t)
            return t, t.shape[2], t.shape[0] * t.shape[1], t.shape[1], bs
        return t, ld_of(t, name), t.shape[0], None, 0

    lda0 = lda1 = ld_add = ld_gate = None
    for i, sg in enumerate(segs):
        S = D.seg[i]
        a0, l0, rows, T3, bs0 = operand(sg["a0"], "a0")
        if sg.get("a0_gather"):
            rows = sg["ids"].numel()
            S.a0_gather = max(1, int(a0.shape[0]))
        lda0 = l0 if lda0 is None else lda0
        if l0 != lda0:
            raise CarcaHipError("gemm_rows: all segments must share lda0")
        S.a0, S.a0_bstride = a0.data_ptr(), bs0
        a1 = sg.get("a1")
        if K1:
            a1, l1, rows1, T31, bs1 = operand(a1, "a1")
            if rows1 != rows:
                raise CarcaHipError("gemm_rows: a0 and a1 row counts differ")
            T3 = T3 if T3 is not None else T31
            lda1 = l1 if lda1 is None else lda1
            S.a1, S.a1_bstride = a1.data_ptr(), bs1
        out = sg.get("out")
        if out is None:
            out = torch.empty(rows, out_ld, dtype=torch.float32, device=a0.device)
        if ld_of(out, "out") != out_ld:
            raise CarcaHipError("gemm_rows: out row stride != out_ld")
        S.c = out.data_ptr()
        outs.append(out)
        ids = sg.get("ids")
        if ids is not None:
            ids = _ids32(ids.reshape(-1))
            keep.append(ids)
            S.ids = ids.data_ptr()
        add, gate, rs = sg.get("add"), sg.get("gate"), sg.get("rowscale")
        if add is not None:
            la = ld_of(add, "add")
            ld_add = la if ld_add is None else ld_add
            S.add = add.data_ptr()
        if gate is not None:
            lg = ld_of(gate, "gate")
            ld_gate = lg if ld_gate is None else ld_gate
            S.gate = gate.data_ptr()
        if rs is not None:
            rs = _f32(rs.reshape(-1))
            keep.append(rs)
            S.rowscale = rs.data_ptr()
        S.rows, S.T, S.add_pos = rows, int(sg.get("T", T3 or 1)), int(bool(sg.get("add_pos", False)))
    D.lda0, D.lda1, D.K0, D.K1 = lda0, lda1 or 0, K0, K1
    D.bt0, D.ldb0 = bt0.data_ptr(), ld_of(bt0, "bt0")
    if K1:
        D.bt1, D.ldb1 = bt1.data_ptr(), ld_of(bt1, "bt1")
    D.N, D.ldc, D.ncols_out = N, out_ld, out_ld if ncols_out is None else ncols_out
    D.bias, D.pos, D.colvec = _ptr(bias), _ptr(pos), _ptr(colvec)
    D.ld_add, D.ld_gate, D.gate_slope, D.mask_rows = ld_add or 0, ld_gate or 0, gate_slope, int(mask_rows)
    D.gate_scale, D.gate_zero_drops, D.alpha = gate_scale, int(gate_zero_drops), alpha
    if add_table is not None:  # v += add_table[ids[row]] (CarcaGemmDesc.add_table: the joint embedding's projected item rows)
        D.add_table, D.ld_add_table = add_table.data_ptr(), ld_of(add_table, "add_table")
    keep += [bt0, bt1, bias, pos, colvec, add_table]
    return D, outs, keep


def gemm_wgrad(segs, N: int, K: int, dw: Tensor, db: Optional[Tensor] = None, mask_rows: bool = False,
               K1: int = 0) -> None:
    """dw[n][k] += sum_r dy[r][n] x[r][k] (and db[n] += sum_r dy[r][n]) over all segments' rows.

    segs: list of dicts with dy [rows, >=N]; x either [rows, >=K] (2-D) or a [B, T, K] view (3-D, users may be
    strided); optional x1 likewise for dw[:, K:K+K1]; optional ids.  dw is a 2-D fp32 view (unit inner stride).
    """
    lib = _lib.load()
    D, _keep = _wgrad_desc(segs, N, K, dw, db, mask_rows, K1)
    _lib.check(lib.carca_gemm_wgrad(C.byref(D), _stream()), "gemm_wgrad")


class WgradGroup:
    """Collects independent weight-gradient products (same arguments as gemm_wgrad) and issues them with ONE launch
    (carca_gemm_wgrad_group): the d x d products of a backward pass are latency-bound launches of ~100 blocks each.
    The C-side module backwards (sa_block_bwd) append their products to the same host array."""

    CAPACITY = 64

    def __init__(self):
        self.arr = (_lib.WgradDesc * self.CAPACITY)()
        self.n = C.c_int(0)
        self.keep = []  # every operand stays referenced until launch() has been issued

    def add(self, segs, N: int, K: int, dw: Tensor, db: Optional[Tensor] = None, mask_rows: bool = False,
            K1: int = 0) -> None:
        if self.n.value >= self.CAPACITY:
            raise CarcaHipError("WgradGroup: too many products")
        D, keep = _wgrad_desc(segs, N, K, dw, db, mask_rows, K1)
        self.arr[self.n.value] = D
        self.n.value += 1
        self.keep.append((keep, segs, dw, db))

    def launch(self) -> None:
        if self.n.value:
            _lib.check(_lib.load().carca_gemm_wgrad_group(self.arr, self.n.value, _stream()), "gemm_wgrad_group")
        self.n.value = 0
        self.keep = []


def sa_block_bwd(dy: Tensor, ids: Tensor, saved: dict, x_in: Tensor, wT, ln_w, grads: dict, B: int, L: int, d: int,
                 H: int, residual: bool, drop_p: float, wg: WgradGroup) -> Tensor:
    """Backward of one SelfAttentionBlock as ONE host call (carca_sa_block_bwd).  wT = (wq_t, wk_t, wv_t, w1_t, w2_t)
    transposed packs, ln_w = (norm1.weight, norm2.weight), grads = the 14 gradient buffers by CarcaSaBwdDesc name.
    Returns d(input) [B*L, DPI]; the block's five weight-gradient products are appended to `wg`."""
    lib = _lib.load()
    dpi, _, _ = padded_dims(d, H)
    rows = B * L
    ws = torch.empty(lib.carca_sa_block_bwd_workspace(B, L, d, H), dtype=torch.float32, device=dy.device)
    dx = torch.empty(rows, dpi, dtype=torch.float32, device=dy.device)
    ids32 = _ids32(ids.reshape(-1))
    D = _lib.SaBwdDesc()
    D.B, D.L, D.d, D.H, D.residual, D.drop_p = B, L, d, H, int(bool(residual)), float(drop_p)
    D.ids, D.dy = ids32.data_ptr(), _row2d(dy, "dy").data_ptr()
    if dy.stride(0) != dpi:
        raise CarcaHipError("sa_block_bwd: dy must have row stride DPI")
    D.x_in = x_in.data_ptr()
    for k in ("qn", "qh", "kh", "vh", "r", "s2", "h1"):
        setattr(D, k, saved[k].data_ptr())
    if drop_p > 0:
        D.m_attn, D.m_ffn2 = saved["m_attn"].data_ptr(), saved["m_ffn2"].data_ptr()
    D.wq_t, D.wk_t, D.wv_t, D.w1_t, D.w2_t = (t.data_ptr() for t in wT)
    D.ln1_w, D.ln2_w = ln_w[0].data_ptr(), ln_w[1].data_ptr()
    for k, t in grads.items():
        setattr(D, k, t.data_ptr())
    D.workspace, D.dx = ws.data_ptr(), dx.data_ptr()
    if wg.n.value + 5 > wg.CAPACITY:
        raise CarcaHipError("WgradGroup: too many products")
    _lib.check(lib.carca_sa_block_bwd(C.byref(D), wg.arr, C.byref(wg.n), _stream()), "sa_block_bwd")
    wg.keep.append((ws, ids32, saved, x_in, wT, ln_w, grads, dy))
    return dx


def cross_score_bwd(groups, p_ids: Tensor, kh: Tensor, vh: Tensor, p_normed: Tensor, enc_out: Optional[Tensor], wT,
                    ffn_w_pad_ptr: int, ffn_w: Tensor, norm_w: Optional[Tensor], grads: dict, B: int, L: int, d: int, H: int, residual: bool,
                    training: bool, drop_p: float, wg: WgradGroup):
    """Backward of the final LayerNorm + CrossAttentionBlock over all target groups as ONE host call
    (carca_cross_score_bwd).  groups: [(qh, y [B,N], dy [B,N], ids [B,N], o [B*N, DPI], m_attn or None)];
    wT = (wq_t, wk_t, wv_t); grads by CarcaCrossBwdDesc name.  Returns (d encoder output [B*L, DPI], [d o_g])."""
    lib = _lib.load()
    dpi, _, _ = padded_dims(d, H)
    ng = len(groups)
    D = _lib.CrossBwdDesc()
    D.B, D.L, D.d, D.H, D.ngroups = B, L, d, H, ng
    D.residual, D.training, D.drop_p = int(bool(residual)), int(bool(training)), float(drop_p)
    Ns = (C.c_int32 * ng)(*[int(g[1].shape[1]) for g in groups])
    ws = torch.empty(lib.carca_cross_score_bwd_workspace(B, L, d, H, Ns, ng), dtype=torch.float32, device=kh.device)
    keep, des = [ws], []
    for i, (qh, y, dy, ids, o, m_attn) in enumerate(groups):
        # y / dy: [B, N] dense, or column blocks of a [B, sum N] tensor (same row stride for both)
        if y.dtype != torch.float32 or dy.dtype != torch.float32 or y.stride(1) != 1 or dy.stride(1) != 1 or \
                y.stride(0) != dy.stride(0):
            y, dy = _f32(y.contiguous()), _f32(dy.contiguous())
        ids32 = _ids32(ids)
        de = torch.empty(o.shape[0], dpi, dtype=torch.float32, device=o.device)
        if o.stride(0) != dpi or o.stride(1) != 1:
            raise CarcaHipError("cross_score_bwd: embedded targets must be [rows, DPI] dense")
        G = D.group[i]
        G.qh, G.y, G.dy, G.ids, G.o, G.de, G.N = (qh.data_ptr(), y.data_ptr(), dy.data_ptr(), ids32.data_ptr(), o.data_ptr(),
                                                  de.data_ptr(), y.shape[1])
        G.ld_y = y.stride(0)
        G.m_attn = m_attn.data_ptr() if (m_attn is not None and drop_p > 0) else None
        keep += [qh, y, dy, ids32, o, m_attn]
        des.append(de)
    p_ids32 = _ids32(p_ids)
    dx = torch.empty(B * L, dpi, dtype=torch.float32, device=kh.device)
    # (norm_w = None: the stand-alone decoder, no final LayerNorm in front of it -- dx = d p_normed)
    D.p_ids, D.kh, D.vh, D.p_normed, D.enc_out = (p_ids32.data_ptr(), kh.data_ptr(), vh.data_ptr(), p_normed.data_ptr(),
                                                   _ptr(enc_out))
    if p_normed.stride(-2) != dpi or (enc_out is not None and enc_out.stride(-2) != dpi):
        raise CarcaHipError("cross_score_bwd: p_normed / enc_out must have row stride DPI")
    D.wq_t, D.wk_t, D.wv_t = (t.data_ptr() for t in wT)
    D.ffn_w_pad, D.ffn_w, D.norm_w = ffn_w_pad_ptr, ffn_w.data_ptr(), _ptr(norm_w)
    for k, t in grads.items():
        setattr(D, k, t.data_ptr())
    D.workspace, D.dx = ws.data_ptr(), dx.data_ptr()
    if wg.n.value + 4 > wg.CAPACITY:
        raise CarcaHipError("WgradGroup: too many products")
    _lib.check(lib.carca_cross_score_bwd(C.byref(D), wg.arr, C.byref(wg.n), _stream()), "cross_score_bwd")
    wg.keep.append((keep, p_ids32, kh, vh, p_normed, enc_out, wT, ffn_w, norm_w, grads))
    return dx, des


# A HipEvent the NEXT embed_bwd call records right before its last launch (CarcaEmbedBwdDesc.ev_early): set by the sharded
# train step, which starts its first gradient all-reduce behind that event (engine.train_step)
early_event = None


def embed_bwd(des, segs, zq: Tensor, joint_wt: Tensor, grads: dict, table: Optional[Tensor], d: int, g: int, n_attrs: int,
              n_ctx: int, L: int, g_pos: Optional[Tensor], joint_only=None, skip_joint: bool = False,
              only_joint: bool = False, table_stream: Optional[int] = None) -> None:
    """Backward of AllEmbedding.forward over all segments as ONE host call (carca_embed_bwd).  des[i]: d e [rows, ld]
    (unmasked); segs[i] = (ids, attrs or None, ctx, is_target); joint_wt: [d + g, ld] transposed joint weight.
    joint_only[i] / skip_joint / only_joint: a pass that runs the target rows' share on a second stream; table_stream: raw
    handle of a stream for the big weight-gradient kernel's row table (include/carca_hip.h)."""
    lib = _lib.load()
    nseg = len(des)
    D = _lib.EmbedBwdDesc()
    D.nseg, D.d, D.g, D.n_attrs, D.n_ctx, D.L = nseg, d, g, n_attrs, n_ctx, L
    keep = []
    rows = (C.c_int32 * nseg)()
    for i, (de, (x, a, c, _tgt)) in enumerate(zip(des, segs)):
        de = _row2d(de, "de")
        if i == 0:
            D.ld_de = de.stride(0)
        elif de.stride(0) != D.ld_de:
            raise CarcaHipError("embed_bwd: all segments must share the row stride of d e")
        ids32 = _ids32(x.reshape(-1))
        S = D.seg[i]
        S.de, S.ids, S.rows, S.T = de.data_ptr(), ids32.data_ptr(), de.shape[0], x.shape[1]
        rows[i] = de.shape[0]
        if only_joint or (joint_only is not None and joint_only[i]):
            S.joint_only = 1
            rows[i] = 0  # (no d [z ; q] workspace)
            keep += [de, ids32]
            continue
        if a is not None:
            a, a_bs = _btk_view(a)
            S.attrs, S.attrs_bstride = a.data_ptr(), a_bs
        elif table is not None:
            S.attrs_table, S.attrs_table_rows = table.data_ptr(), table.shape[0]
        else:
            raise CarcaHipError("embed_bwd: attrs is None and no attribute table is registered")
        if n_ctx > 0:
            c, c_bs = _btk_view(c)
            S.ctx, S.ctx_bstride = c.data_ptr(), c_bs
        keep += [de, ids32, a, c]
    ws = torch.empty(lib.carca_embed_bwd_workspace(rows, nseg, d, g), dtype=torch.float32, device=zq.device)
    D.zq, D.joint_wt, D.ld_joint_wt = zq.data_ptr(), joint_wt.data_ptr(), joint_wt.stride(0)
    for k, t in grads.items():
        setattr(D, k, t.data_ptr())
    D.g_pos = _ptr(g_pos)
    D.workspace = ws.data_ptr()
    D.ev_early = early_event.handle if early_event is not None else None
    D.skip_joint = 1 if skip_joint else 0
    D.only_joint = 1 if only_joint else 0
    D.table_stream = table_stream
    _lib.check(lib.carca_embed_bwd(C.byref(D), _stream()), "embed_bwd")
    del keep


def _wgrad_desc(segs, N: int, K: int, dw: Tensor, db: Optional[Tensor], mask_rows: bool, K1: int):
    D = _lib.WgradDesc()
    D.nseg = len(segs)
    keep = []
    ld_dy = ld_x = ld_x1 = None

    def xinfo(x, name):
        _need_cuda(x)
        if x.dim() == 3:
            x, bs = _btk_view(x)
            keep.append(x)
            return x.data_ptr(), x.shape[2], x.shape[0] * x.shape[1], x.shape[1], bs
        if x.dim() != 2 or x.stride(1) != 1 or x.dtype != torch.float32:
            raise CarcaHipError(f"gemm_wgrad: {name} must be a 2-D fp32 view with unit inner stride or a [B,T,K] tensor")
        return x.data_ptr(), x.stride(0), x.shape[0], 1, 0

    for i, sg in enumerate(segs):
        dy = sg["dy"]
        _need_cuda(dy)
        if dy.dim() != 2 or dy.stride(1) != 1 or dy.dtype != torch.float32:
            raise CarcaHipError("gemm_wgrad: dy must be a 2-D fp32 view with unit inner stride")
        xp, lx, xrows, xT, xbs = xinfo(sg["x"], "x")
        if sg.get("x_gather"):
            xrows = dy.shape[0]  # x is a table indexed by ids
        if dy.shape[0] != xrows:
            raise CarcaHipError("gemm_wgrad: dy and x row counts differ")
        ld_dy = dy.stride(0) if ld_dy is None else ld_dy
        ld_x = lx if ld_x is None else ld_x
        if dy.stride(0) != ld_dy or lx != ld_x:
            raise CarcaHipError("gemm_wgrad: all segments must share row strides")
        S = D.seg[i]
        S.dy, S.x, S.rows, S.T, S.x_bstride = dy.data_ptr(), xp, dy.shape[0], xT, xbs
        S.x_gather = max(1, int(sg["x"].shape[0])) if sg.get("x_gather", False) else 0
        if K1:
            x1p, lx1, x1rows, x1T, x1bs = xinfo(sg["x1"], "x1")
            if x1rows != dy.shape[0] or (xbs and x1bs and x1T != xT):
                raise CarcaHipError("gemm_wgrad: x1 does not match x")
            ld_x1 = lx1 if ld_x1 is None else ld_x1
            S.x1, S.x1_bstride = x1p, x1bs
            S.T = max(xT, x1T)
        ids = sg.get("ids")
        if ids is not None:
            ids = _ids32(ids.reshape(-1))
            keep.append(ids)
            S.ids = ids.data_ptr()
    if dw.dim() != 2 or dw.stride(1) != 1 or dw.dtype != torch.float32:
        raise CarcaHipError("gemm_wgrad: dw must be a 2-D fp32 view with unit inner stride")
    D.ld_dy, D.ld_x, D.ld_x1, D.N, D.K, D.K1 = ld_dy, ld_x, ld_x1 or 0, N, K, K1
    D.dw, D.ldw, D.db, D.mask_rows = dw.data_ptr(), dw.stride(0), _ptr(db), int(mask_rows)
    return D, keep


# --------------------------------------------------------------------------------------------------
# backward kernels
# --------------------------------------------------------------------------------------------------
def _row2d(t: Tensor, name: str) -> Tensor:
    _need_cuda(t)
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
        raise CarcaHipError(f"{name} must be a 2-D fp32 view with unit inner stride")
    return t


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, d: int, out_ld: int, addend: Optional[Tensor] = None,
                  dgamma: Optional[Tensor] = None, dbeta: Optional[Tensor] = None) -> Tensor:
    """dx [rows, out_ld] = dLayerNorm(dy; x, gamma) (+ addend); dgamma/dbeta accumulate (atomics)."""
    lib = _lib.load()
    dy, x = _row2d(dy, "dy"), _row2d(x, "x")
    rows = dy.shape[0]
    dx = torch.empty(rows, out_ld, dtype=torch.float32, device=dy.device)
    if addend is not None:
        addend = _row2d(addend, "addend")
    _lib.check(lib.carca_layernorm_bwd(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(), rows, d,
                                       _ptr(addend), addend.stride(0) if addend is not None else 0, dx.data_ptr(),
                                       out_ld, out_ld, _ptr(dgamma), _ptr(dbeta), _stream()), "layernorm_bwd")
    return dx


def layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, d: int, out_ld: int) -> Tensor:
    """y [rows, out_ld] = LayerNorm(x[:, :d]) with zeroed pad columns (carca.py:421 outside the cross-attention kernel)."""
    lib = _lib.load()
    x = _row2d(x, "x")
    y = torch.empty(x.shape[0], out_ld, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_layernorm_fwd(x.data_ptr(), x.stride(0), y.data_ptr(), out_ld, x.shape[0], d, w.data_ptr(),
                                       b.data_ptr(), _stream()), "layernorm_fwd")
    return y


def dot_score_fwd(p: Tensor, o: Tensor, B: int, L: int, T: int, d: int, slotwise: bool, link: int = 0) -> Tensor:
    """y [B, T] = link(p_row . o) (DotProduct / WeightedDotProduct scoring, carca.py:361-367,390-397)."""
    lib = _lib.load()
    p, o = _row2d(p, "p"), _row2d(o, "o")
    y = torch.empty(B, T, dtype=torch.float32, device=p.device)
    _lib.check(lib.carca_dot_score_fwd(p.data_ptr(), p.stride(0), o.data_ptr(), o.stride(0), y.data_ptr(), B, L, T, d,
                                       int(bool(slotwise)), int(link), _stream()), "dot_score_fwd")
    return y


def dot_score_bwd(p: Tensor, o: Tensor, y: Tensor, dy: Tensor, dp: Tensor, B: int, L: int, T: int, d: int, slotwise: bool,
                  link: int, out_ld: int) -> Tensor:
    """Returns d_o [B*T, out_ld]; ACCUMULATES into dp [B*L, ld] (caller zeroes it before the first group)."""
    lib = _lib.load()
    p, o, dp = _row2d(p, "p"), _row2d(o, "o"), _row2d(dp, "dp")
    y, dy = _f32(y.reshape(-1)), _f32(dy.reshape(-1))
    d_o = torch.empty(B * T, out_ld, dtype=torch.float32, device=p.device)
    _lib.check(lib.carca_dot_score_bwd(p.data_ptr(), p.stride(0), o.data_ptr(), o.stride(0), y.data_ptr(), dy.data_ptr(),
                                       dp.data_ptr(), dp.stride(0), d_o.data_ptr(), out_ld, B, L, T, d,
                                       int(bool(slotwise)), int(link), _stream()), "dot_score_bwd")
    return d_o


def add_positions(x: Tensor, pos: Tensor) -> Tensor:
    """Encoding.forward (carca.py:25-31, 54-60): x [B, T, d] + pos [T, d]."""
    _need_cuda(x, pos)
    x, pos = _f32(x), _f32(pos.detach()).contiguous()
    if x.dim() != 3 or x.stride(2) != 1 or x.stride(0) != x.shape[1] * x.stride(1) or pos.shape != (x.shape[1], x.shape[2]):
        raise CarcaHipError("add_positions: x must be [B, T, d] with dense rows and pos [T, d]")
    B, T, d = x.shape
    out = torch.empty(B, T, d, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().carca_add_positions(x.data_ptr(), x.stride(1), pos.data_ptr(), out.data_ptr(), d, B, T, d,
                                               _stream()), "add_positions")
    return out


def mha_core(q: Tensor, k: Tensor, v: Tensor, q_ids: Tensor, k_ids: Tensor, H: int, causal: Optional[int],
             want_w: bool, drop: Optional[Tuple[float, int, int]] = None):
    """Attention core of MultiHeadAttention.forward (carca.py:242-260) on projected q [B, Tq, d], k / v [B, Tk, d].
    drop = (p, seed, site): nn.Dropout on the weights (carca.py:258); then a third value is returned, the keep-mask
    [B, H, Tq, Tk] (uint8) the backward and the tests replay."""
    _need_cuda(q, k, v, q_ids, k_ids)
    B, Tq, d = q.shape
    Tk = k.shape[1]
    q2, k2, v2 = (_f32(t).reshape(-1, d) for t in (q, k, v))
    if k2.stride(0) != v2.stride(0):
        raise CarcaHipError("mha_core: k and v must share their row stride")
    qi, ki = _ids32(q_ids.reshape(-1)), _ids32(k_ids.reshape(-1))
    out = torch.empty(B, Tq, d, dtype=torch.float32, device=q.device)
    w = torch.empty(H * B, Tq, Tk, dtype=torch.float32, device=q.device) if want_w else None
    if drop is not None and drop[0] > 0:
        keep = torch.empty(B, H, Tq, Tk, dtype=torch.uint8, device=q.device)
        ds = _drop_struct(*drop)
        _lib.check(_lib.load().carca_mha_core_drop(q2.data_ptr(), q2.stride(0), k2.data_ptr(), v2.data_ptr(), k2.stride(0),
                                                   qi.data_ptr(), ki.data_ptr(), B, Tq, Tk, d, H, int(causal is not None),
                                                   int(causal or 0), out.data_ptr(), d, _ptr(w), C.byref(ds),
                                                   keep.data_ptr(), _stream()), "mha_core")
        return out, w, keep
    _lib.check(_lib.load().carca_mha_core(q2.data_ptr(), q2.stride(0), k2.data_ptr(), v2.data_ptr(), k2.stride(0),
                                          qi.data_ptr(), ki.data_ptr(), B, Tq, Tk, d, H, int(causal is not None),
                                          int(causal or 0), out.data_ptr(), d, _ptr(w), _stream()), "mha_core")
    return out, w


def mha_core_bwd(q: Tensor, k: Tensor, v: Tensor, q_ids: Tensor, k_ids: Tensor, H: int, causal: Optional[int],
                 d_out: Optional[Tensor], d_w: Optional[Tensor], keep: Optional[Tensor] = None, p: float = 0.0):
    """Backward of mha_core: (dq, dk, dv) for projected q [B, Tq, d], k / v [B, Tk, d]; keep / p: the forward's keep-mask
    and dropout probability."""
    B, Tq, d = q.shape
    Tk = k.shape[1]
    q2, k2, v2 = (_f32(t).reshape(-1, d).contiguous() for t in (q, k, v))
    qi, ki = _ids32(q_ids.reshape(-1)), _ids32(k_ids.reshape(-1))
    do = _f32(d_out).reshape(-1, d).contiguous() if d_out is not None else None
    dw = _f32(d_w).contiguous() if d_w is not None else None
    dq = torch.empty(B * Tq, d, dtype=torch.float32, device=q.device)
    dk = torch.zeros(B * Tk, d, dtype=torch.float32, device=q.device)
    dv = torch.zeros(B * Tk, d, dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().carca_mha_core_bwd_drop(q2.data_ptr(), d, k2.data_ptr(), v2.data_ptr(), d, qi.data_ptr(),
                                                   ki.data_ptr(), B, Tq, Tk, d, H, int(causal is not None), int(causal or 0),
                                                   _ptr(do), d, _ptr(dw), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                   _ptr(keep), 1.0 / (1.0 - p) if keep is not None else 1.0, _stream()),
               "mha_core_bwd")
    return dq.view(B, Tq, d), dk.view(B, Tk, d), dv.view(B, Tk, d)


def _user_rows(t: Tensor, name: str) -> Tensor:
    """[B, T, F] fp32 with contiguous rows per user (any user stride): as is; anything else: one contiguous copy."""
    _need_cuda(t)
    if t.dtype != torch.float32:
        raise CarcaHipError(f"{name}: expected float32, got {t.dtype}")
    if t.dim() != 3:
        raise ValueError(f"{name}: expected [B, T, F], got {tuple(t.shape)}")
    ok = t.stride(2) == 1 and t.stride(1) == t.shape[2] and t.stride(0) >= t.shape[1] * t.shape[2]
    return t if ok or t.numel() == 0 else t.contiguous()


def knn_score(p_a: Optional[Tensor], o_a: Optional[Tensor], p_x: Optional[Tensor] = None, o_x: Optional[Tensor] = None,
              table: Optional[Tensor] = None) -> Tensor:
    """KNN.forward for one target group (knn.py:13-19): y [B, T] = attrs(last profile slot) . attrs(target).
    Dense: p_a [B, L, F], o_a [B, T, F] (views with a user stride, e.g. torch.split halves, are read in place).
    Table: `table` [rows, F] with p_x [B, L], o_x [B, T] int32 ids."""
    lib = _lib.load()
    if table is not None:
        _need_cuda(table, p_x, o_x)
        table, p_x, o_x = _f32(table), _ids32(p_x), _ids32(o_x)
        B, L = p_x.shape
        T, F = o_x.shape[1], table.shape[1]
        y = torch.empty(B, T, dtype=torch.float32, device=table.device)
        args = (table.data_ptr(), 0, None, 0, p_x.data_ptr(), o_x.data_ptr(), table.shape[0])
    else:
        p_a, o_a = _user_rows(p_a, "knn_score: p_a"), _user_rows(o_a, "knn_score: o_a")
        B, L, F = p_a.shape
        T = o_a.shape[1]
        if o_a.shape[0] != B or o_a.shape[2] != F:
            raise ValueError(f"knn_score: targets {tuple(o_a.shape)} do not match profile {tuple(p_a.shape)}")
        y = torch.empty(B, T, dtype=torch.float32, device=p_a.device)
        args = (p_a.data_ptr(), p_a.stride(0), o_a.data_ptr(), o_a.stride(0), None, None, 0)
    if B * T and L and F:
        _lib.check(lib.carca_knn_score(*args, y.data_ptr(), B, L, T, F, _stream()), "knn_score")
    elif B * T:
        y.zero_()
    return y


def slot_decay_scale(x: Tensor, B: int, L: int, d: int, gamma: float, out_ld: int) -> Tensor:
    """out[b][t] = x[b][t] * sum_{j<=t} gamma^j (WeightedDotProduct's history weights; its own backward); [B*L, out_ld]."""
    lib = _lib.load()
    x = _row2d(x, "x")
    out = torch.empty(B * L, out_ld, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_slot_decay_scale(x.data_ptr(), x.stride(0), out.data_ptr(), out_ld, B, L, d, float(gamma),
                                          _stream()), "slot_decay_scale")
    return out


def l2norm_fwd(x: Tensor, d: int, out_ld: int) -> Tensor:
    lib = _lib.load()
    x = _row2d(x, "x")
    y = torch.empty(x.shape[0], out_ld, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_l2norm_fwd(x.data_ptr(), x.stride(0), y.data_ptr(), out_ld, x.shape[0], d, _stream()),
               "l2norm_fwd")
    return y


def l2norm_bwd(x: Tensor, dy: Tensor, d: int, out_ld: int) -> Tensor:
    lib = _lib.load()
    x, dy = _row2d(x, "x"), _row2d(dy, "dy")
    dx = torch.empty(x.shape[0], out_ld, dtype=torch.float32, device=x.device)
    _lib.check(lib.carca_l2norm_bwd(x.data_ptr(), x.stride(0), dy.data_ptr(), dy.stride(0), dx.data_ptr(), out_ld,
                                    x.shape[0], d, _stream()), "l2norm_bwd")
    return dx


def _id_lists(id_lists):
    keep = [_ids32(t.reshape(-1)) for t in id_lists]
    for t in keep:
        _need_cuda(t)
    n = len(keep)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in keep])
    counts = (C.c_int64 * n)(*[t.numel() for t in keep])
    return keep, ptrs, counts, n


# --------------------------------------------------------------------------------------------------
# deterministic mode (include/carca_hip.h, tuning key 8): same inputs => same bits, run to run
# --------------------------------------------------------------------------------------------------
TUNE_DETERMINISTIC = 8
_deterministic = False


def set_deterministic(on: bool) -> None:
    """No fp32 atomics in the backward pass from here on: gradient accumulations go through a 64-bit fixed-point shadow
    of the pass's gradient buffer (order-independent integer sums), HR / NDCG sums are added in a fixed order.  A few
    per cent slower (one extra sweep over the gradient buffer per step); gradients differ from the default path's by
    fp32 round-off only.  Off by default -- torch's own CUDA backward is not deterministic either."""
    global _deterministic
    _lib.check(_lib.load().carca_set_tuning(TUNE_DETERMINISTIC, 1 if on else 0), "set_tuning")
    _deterministic = bool(on)


def deterministic() -> bool:
    return _deterministic


def det_begin(flat: Optional[Tensor], shadow: Optional[Tensor]) -> None:
    if flat is None:
        _lib.check(_lib.load().carca_det_begin(None, 0, None, _stream()), "det_begin")
        return
    if shadow.dtype != torch.int64 or shadow.numel() < flat.numel() or not shadow.is_contiguous():
        raise CarcaHipError("det_begin: the shadow must be a contiguous int64 tensor of the gradient buffer's length")
    _lib.check(_lib.load().carca_det_begin(flat.data_ptr(), flat.numel(), shadow.data_ptr(), _stream()), "det_begin")


def det_flush(flat: Tensor, shadow: Tensor, lo: int, hi: int) -> None:
    _lib.check(_lib.load().carca_det_flush(flat.data_ptr(), shadow.data_ptr(), int(lo), int(hi), _stream()), "det_flush")


def mark_rows(mask: Tensor, ids: Tensor) -> None:
    """mask[ids] = 1 (uint8 row mask of an embedding table: CarcaAdamTensor.row_mask)."""
    ids32 = _ids32(ids.reshape(-1))
    _need_cuda(mask, ids32)
    if mask.dtype != torch.uint8 or not mask.is_contiguous():
        raise CarcaHipError("mark_rows: mask must be a contiguous uint8 tensor")
    _lib.check(_lib.load().carca_mark_rows(ids32.data_ptr(), ids32.numel(), mask.data_ptr(), mask.numel(), _stream()),
               "mark_rows")


def zero_rows(table: Tensor, id_lists) -> None:
    """table[ids] = 0 for up to MAX_SEGS id tensors (the rows a previous scatter-add touched)."""
    _need_cuda(table)
    if table.dim() != 2 or not table.is_contiguous() or table.dtype != torch.float32:
        raise CarcaHipError("zero_rows: table must be a contiguous fp32 [rows, d] tensor")
    keep, ptrs, counts, n = _id_lists(id_lists)
    _lib.check(_lib.load().carca_zero_rows(table.data_ptr(), table.shape[0], table.shape[1], ptrs, counts, n, _stream()),
               "zero_rows")
    del keep


def concat_ids(id_lists, out: Tensor) -> None:
    """out[: sum of sizes] = the id tensors back to back (int32)."""
    keep, ptrs, counts, n = _id_lists(id_lists)
    if out.dtype != torch.int32 or not out.is_contiguous() or out.numel() < sum(t.numel() for t in keep):
        raise CarcaHipError("concat_ids: out must be a contiguous int32 tensor with room for every id")
    _lib.check(_lib.load().carca_concat_ids(ptrs, counts, n, out.data_ptr(), _stream()), "concat_ids")
    del keep


def embed_scatter(dz: Tensor, ids: Tensor, d: int, scale: float, d_items: Tensor) -> None:
    lib = _lib.load()
    dz = _row2d(dz, "dz")
    ids32 = _ids32(ids.reshape(-1))
    _lib.check(lib.carca_embed_scatter(dz.data_ptr(), dz.stride(0), ids32.data_ptr(), dz.shape[0], d, scale,
                                       d_items.data_ptr(), _stream()), "embed_scatter")


def colsum(x: Tensor, cols: int, out: Tensor, rowscale: Optional[Tensor] = None, ids: Optional[Tensor] = None,
           T: int = 1) -> None:
    """out[(row % T)][c] += sum_rows rowscale[row] * (ids[row] != 0) * x[row][c]."""
    lib = _lib.load()
    x = _row2d(x, "x")
    ids32 = _ids32(ids.reshape(-1)) if ids is not None else None
    rs = _f32(rowscale.reshape(-1)) if rowscale is not None else None
    _lib.check(lib.carca_colsum(x.data_ptr(), x.stride(0), x.shape[0], cols, _ptr(rs), _ptr(ids32), T, out.data_ptr(),
                                _stream()), "colsum")


def sa_attn_bwd(qh: Tensor, kh: Tensor, vh: Tensor, d_attn: Tensor, ids: Tensor, B: int, L: int, d: int, H: int,
                m_attn: Optional[Tensor] = None, drop_scale: float = 1.0):
    lib = _lib.load()
    d_attn = _row2d(d_attn, "d_attn")
    ids32 = _ids32(ids.reshape(-1))
    dqh, dkh, dvh = torch.empty_like(qh), torch.empty_like(kh), torch.empty_like(vh)
    _lib.check(lib.carca_sa_attn_bwd(qh.data_ptr(), kh.data_ptr(), vh.data_ptr(), d_attn.data_ptr(), d_attn.stride(0),
                                     ids32.data_ptr(), dqh.data_ptr(), dkh.data_ptr(), dvh.data_ptr(), B, L, d, H,
                                     _ptr(m_attn), drop_scale, _stream()), "sa_attn_bwd")
    return dqh, dkh, dvh


def cross_attn_bwd(kh: Tensor, vh: Tensor, p_ids: Tensor, groups, ffn_w_pad_ptr: int, d_ffn_w_pad: Tensor, B: int,
                   L: int, d: int, H: int, training: bool, masks=None, drop_scale: float = 1.0):
    """groups: [(qh [B*N,DPO], y [B,N], dy [B,N], ids [B,N])] -> ([dqh], [dlogit], dkh, dvh)."""
    lib = _lib.load()
    arr = (_lib.CrossBwdGroup * len(groups))()
    keep, dqhs, dls = [], [], []
    for i, (qh, y, dy, ids) in enumerate(groups):
        y, dy, ids32 = _f32(y), _f32(dy), _ids32(ids)
        dqh = torch.empty_like(qh)
        dl = torch.empty(y.numel(), dtype=torch.float32, device=y.device)
        keep += [y, dy, ids32]
        dqhs.append(dqh)
        dls.append(dl)
        g = arr[i]
        g.qh, g.y, g.dy, g.ids, g.dqh, g.dlogit, g.N = (qh.data_ptr(), y.data_ptr(), dy.data_ptr(), ids32.data_ptr(),
                                                        dqh.data_ptr(), dl.data_ptr(), y.shape[1])
        g.m_attn = masks[i].data_ptr() if masks is not None else None
    p_ids32 = _ids32(p_ids)
    dkh, dvh = torch.empty_like(kh), torch.empty_like(vh)
    _lib.check(lib.carca_cross_attn_bwd(kh.data_ptr(), vh.data_ptr(), p_ids32.data_ptr(), arr, len(groups),
                                        ffn_w_pad_ptr, dkh.data_ptr(), dvh.data_ptr(), d_ffn_w_pad.data_ptr(), B, L, d,
                                        H, int(bool(training)), drop_scale, _stream()), "cross_attn_bwd")
    return dqhs, dls, dkh, dvh


def set_tuning(key: int, value: int) -> None:
    """carca_set_tuning (include/carca_hip.h): kernel-variant knobs; 0 = the shipped choice."""
    _lib.check(_lib.load().carca_set_tuning(int(key), int(value)), "set_tuning")


def num_cus() -> int:
    return torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count


def poll_errors() -> None:
    """carca_poll_errors: raises CarcaHipError if a kernel of an earlier launch reported a failure that no launch status
    could carry (a stream-K taker of the feature GEMM that gave up waiting for its partial tile).  One read of host
    memory; meaningful wherever the host has synchronised with the device anyway."""
    _lib.check(_lib.load().carca_poll_errors(), "poll_errors")


def capture_scope() -> int:
    """Id of the hipGraph capture in progress on the current stream (0 = not capturing); memory the library allocates for
    captured kernels is accounted to it (capture_bytes) and freed by capture_release(id) once the graph is destroyed."""
    out = C.c_ulonglong(0)
    _lib.check(_lib.load().carca_capture_scope(_stream(), C.byref(out)), "capture_scope")
    return int(out.value)


def capture_bytes(scope: int) -> int:
    return int(_lib.load().carca_capture_bytes(C.c_ulonglong(scope)))


def capture_release(scope: int) -> None:
    if scope:
        _lib.check(_lib.load().carca_capture_release(C.c_ulonglong(scope)), "capture_release")


# feature-GEMM precision (include/carca_hip.h, tuning key 16): opt-in split-precision products on the 16-bit MFMA pipe
TUNE_SPLIT_GEMM = 16
SPLIT_GEMM_MODES = {"fp32": 0, "bf16x3": 1, "fp16x2": 2}
_split_mode = 0


def gemm_rows_log(on: Optional[bool] = None) -> str:
    """on=True / False: clear the calling thread's log of row-GEMM launches and switch it on / off; None: the log so far
    ("kernel rows=.. N=.. K=.. grid=..;" per launch)."""
    lib = _lib.load()
    if on is not None:
        lib.carca_gemm_rows_log(None, int(bool(on)))
        return ""
    buf = C.create_string_buffer(65536)
    lib.carca_gemm_rows_log(buf, len(buf))
    return buf.value.decode()


def set_feature_gemm_precision(mode: str, force: bool = False) -> None:
    """'fp32' (default): exact-fp32 MFMA.  'bf16x3': both operands split into three bf16 parts, six products, fp32
    accumulation.  'fp16x2': two fp16 parts, three products (|operands| < 65504).  Applies to AllEmbedding.feats_embed
    (carca.py:86) where the one-workgroup-per-CU kernel runs and n_attrs is a multiple of 32; every other product keeps
    the fp32 kernels.  force: take the split kernel wherever its own conditions hold, whatever the grid (parity tests at
    fixture sizes)."""
    global _split_mode
    set_tuning(TUNE_SPLIT_GEMM, SPLIT_GEMM_MODES[mode] | (16 if force and SPLIT_GEMM_MODES[mode] else 0))
    _split_mode = SPLIT_GEMM_MODES[mode]
    if _split_mode == 0:
        _lib.check(_lib.load().carca_split_bind(None, None, 0, 0, 0), "split_bind")


def feature_gemm_mode() -> int:
    return _split_mode


def split_launch_count() -> int:
    return int(_lib.load().carca_split_launch_count())


def split_pack(w: Tensor, K0: int, mode: int) -> Tensor:
    """Packed 16-bit planes of w[:, :K0] (carca_split_pack) for the split-precision row GEMM: prepare once per weight version."""
    w = _f32(w.detach())
    _need_cuda(w)
    lib = _lib.load()
    N = w.shape[0]
    out = torch.empty(int(lib.carca_split_bytes(N, int(K0), int(mode))), dtype=torch.uint8, device=w.device)
    _lib.check(lib.carca_split_pack(w.data_ptr(), w.stride(0), int(K0), N, int(mode), out.data_ptr(), _stream()), "split_pack")
    return out


def split_bind(w: Optional[Tensor], planes: Optional[Tensor], mode: int = 0, K0: int = 0) -> None:
    """This thread's following launches use `planes` wherever a row GEMM's weight pointer is w's (carca_split_bind)."""
    lib = _lib.load()
    if w is None:
        _lib.check(lib.carca_split_bind(None, None, 0, 0, 0), "split_bind")
    else:
        _lib.check(lib.carca_split_bind(w.data_ptr(), planes.data_ptr(), int(mode), w.shape[0], int(K0)), "split_bind")
