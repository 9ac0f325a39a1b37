"""Profiles LONGER than the fused per-user kernels hold (L > 64 slots), models WIDER than they hold (d > 128), more than three
target groups: CARCA.forward composed from the library's row-level kernels, differentiable through torch.autograd.

The fused attention kernels (sa_eval / sa_block / cross_stream / cross_score and their backward kernels) keep one user's
whole profile -- keys, values, the block's activations -- in one workgroup's LDS: 64 slots at d <= 128.  The reference
takes any `--seq_len` (scripts/training.py:34-63), so a longer profile runs the same arithmetic as separate launches:
    LayerNorm                 carca_layernorm_fwd / carca_layernorm_bwd           (carca.py:298,303,421)
    every Linear / Conv1d(k=1) carca_gemm_rows / carca_gemm_wgrad                 (carca.py:238-240,306-312,344)
    the attention core        carca_mha_core(_drop) / carca_mha_core_bwd(_drop)   (carca.py:242-260; one wave per
                              (user, head, query), any Tq / Tk up to 1024 keys, dropout on the weights inside)
    nn.Dropout sites          carca_dropout_fwd / carca_mask_mul                  (carca.py:309,312,416)
    embeddings, dot decoders  the modules' own embed_segments / score_groups and their backward halves
and the glue between them (residual adds, LeakyReLU, the decoder's d -> 1 product, sigmoid) as torch element-wise ops on
the device.  Same results as the fused path where both apply (tests/test_hip_long_profile.py runs L = 50 through both),
several times slower per user: this is the envelope, the fused kernels are the hot path.
Dropout: same counter-based keep-masks as the fused kernels (include/carca_hip.h, CarcaDropout), exported through
model._keep_dropout_masks / model._last_dropout_masks in the fused path's layout so the oracle can replay them."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import _lib, ops
from ._lib import CarcaHipError


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        d = x.shape[-1]
        x2 = ops._f32(x.detach()).reshape(-1, d)
        y = ops.layernorm_fwd(x2, w.detach(), b.detach(), d, d)
        ctx.save_for_backward(x2, w.detach())
        ctx.needs = (x.requires_grad, w.requires_grad or b.requires_grad)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        d = x2.shape[1]
        dg = db = None
        if ctx.needs[1]:
            dg = torch.zeros(d, dtype=torch.float32, device=dy.device)
            db = torch.zeros(d, dtype=torch.float32, device=dy.device)
        dx = ops.layernorm_bwd(dy.contiguous().view(-1, d), x2, w, d, d, dgamma=dg, dbeta=db)
        return dx.view(dy.shape), dg, db


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, site, sink, key):
        y = ops._f32(x.detach()).clone()
        mask = ops.dropout_fwd(y, y.shape[-1], p, seed, site)
        if sink is not None:
            sink(key, mask)
        ctx.mask, ctx.scale = mask, 1.0 / (1.0 - p)
        return y

    @staticmethod
    def backward(ctx, dy):
        cols = dy.shape[-1]
        dx = ops.mask_mul(dy.contiguous().view(-1, cols), ctx.mask, ctx.scale, cols, cols)
        return dx.view(dy.shape), None, None, None, None, None


class _AttnFn(torch.autograd.Function):
    """carca.py:242-260 on projected q, k, v with the weights' dropout (carca.py:258) inside the kernel."""

    @staticmethod
    def forward(ctx, q, k, v, q_ids, k_ids, H, causal, drop, sink, key):
        res = ops.mha_core(q.detach(), k.detach(), v.detach(), q_ids, k_ids, H, causal, False, drop=drop)
        keep = res[2] if len(res) == 3 else None
        if sink is not None and keep is not None:
            sink(key, keep)
        ctx.save_for_backward(q.detach(), k.detach(), v.detach())
        ctx.t = (q_ids, k_ids, H, causal, keep, drop[0] if keep is not None else 0.0)
        return res[0]

    @staticmethod
    def backward(ctx, d_out):
        q, k, v = ctx.saved_tensors
        q_ids, k_ids, H, causal, keep, p = ctx.t
        dq, dk, dv = ops.mha_core_bwd(q, k, v, q_ids, k_ids, H, causal, d_out, None, keep=keep, p=p)
        return dq, dk, dv, None, None, None, None, None, None, None


class _EmbedSegsFn(torch.autograd.Function):
    """Embedding.forward (carca.py:85-95 and the ablation variants) over the profile and every target group in one call of
    the module's embed_segments; backward = its embed_backward (what autograd._EmbedFn does for one segment)."""

    @staticmethod
    def forward(ctx, module, segs, *params):
        d = module.d
        dpi = ops.row_ld(d)
        es, saved = module.embed_segments(segs, ld_e=dpi)
        ctx.module, ctx.params, ctx.segs, ctx.saved, ctx.dpi = module, params, segs, saved, dpi
        return tuple(e[..., :d] for e in es)

    @staticmethod
    def backward(ctx, *des):
        from .autograd import _PackPlan, _pad_to, _standalone_pass

        module, params, segs, dpi = ctx.module, ctx.params, ctx.segs, ctx.dpi
        d = module.d
        late = module.late_grad_params(ctx.saved) if hasattr(module, "late_grad_params") else ()
        ids = [s[0] for s in segs] if hasattr(module, "items_embed") else None
        grads, gbp, det, after = _standalone_pass(module, params, _PackPlan(), ids, late)
        full = []
        for de, s in zip(des, segs):
            if de is None:
                de = torch.zeros(*s[0].shape, d, dtype=torch.float32, device=s[0].device)
            full.append(_pad_to(de, dpi).view(-1, dpi))
        module.embed_backward(full, list(segs), ctx.saved, gbp, segs[0][0].shape[1], dpi)
        if det is not None:
            det.finish()
        after()
        return (None, None) + tuple(grads)


class _DotScoreFn(torch.autograd.Function):
    """DotProduct / WeightedDotProduct (carca.py:352-399): the decoder's score_groups / score_backward halves."""

    @staticmethod
    def forward(ctx, dec, B, L, d, p2d, *os2d):
        Ts = [o.shape[0] // B for o in os2d]
        ys, sv = dec.score_groups(p2d.detach().contiguous(), [o.detach().contiguous() for o in os2d], Ts, B, L, d, d)
        ctx.dec, ctx.sv, ctx.dims = dec, sv, (B, L, d)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        B, L, d = ctx.dims
        dys = [dy.contiguous() if dy is not None else torch.zeros_like(y) for dy, y in zip(dys, ctx.sv["ys"])]
        dp, dos = ctx.dec.score_backward(dys, ctx.sv, B, L, d, d)
        return (None, None, None, None, dp[:, :d]) + tuple(o[:, :d] for o in dos)


def _linear(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    from .autograd import _LinearFn

    return _LinearFn.apply(x, w, b)


def _norm(x: Tensor, ln) -> Tensor:
    return _LayerNormFn.apply(x, ln.weight, ln.bias)


class _Sink:
    """Keep-masks of one forward in the layout of the fused path's model._last_dropout_masks."""

    def __init__(self, n_blocks: int, n_groups: int):
        self.masks = {"embed": None, "blocks": [dict() for _ in range(n_blocks)], "cross": [None] * n_groups}

    def __call__(self, key, mask):
        if key[0] == "embed":
            self.masks["embed"] = mask
        elif key[0] == "block":
            self.masks["blocks"][key[1]][key[2]] = mask
        else:
            self.masks["cross"][key[1]] = mask


def sa_block(blk, x: Tensor, ids: Tensor, seed: int = 0, site0: int = 0, sink=None, index: int = 0) -> Tensor:
    """SelfAttentionBlock.forward (carca.py:297-318) on x [B, L, d], ids [B, L] (0 = pad)."""
    a = blk.attn
    p = blk.drop_p()
    q = _norm(x, blk.norm1)
    Q, K, V = _linear(q, a.WQ.weight, a.WQ.bias), _linear(x, a.WK.weight, a.WK.bias), _linear(x, a.WV.weight, a.WV.bias)
    s = _AttnFn.apply(Q, K, V, ids, ids, a.H, 0, (p, seed, site0) if p > 0 else None, sink, ("block", index, "m_attn"))
    if blk.residual:
        s = s + q  # the NORMED tensor is the residual (carca.py:301-302)
    s = _norm(s, blk.norm2)
    f = F.leaky_relu(_linear(s, blk.ffn_1.weight[:, :, 0], blk.ffn_1.bias), blk.lrelu.negative_slope)
    if p > 0:
        f = _DropoutFn.apply(f, p, seed, site0 + 1, sink, ("block", index, "m_ffn1"))
    f = _linear(f, blk.ffn_2.weight[:, :, 0], blk.ffn_2.bias)
    if p > 0:
        f = _DropoutFn.apply(f, p, seed, site0 + 2, sink, ("block", index, "m_ffn2"))
    if blk.residual:
        f = f + s
    return f


def cross_block(dec, o: Tensor, o_ids: Tensor, p_n: Tensor, p_ids: Tensor, seed: int = 0, site: int = 2000, sink=None,
                index: int = 0) -> Tensor:
    """CrossAttentionBlock.forward (carca.py:338-349) up to the squeeze: scores [B, N] for o [B, N, d] against the
    final-normed profile p_n [B, L, d]."""
    a = dec.attn
    p = dec.drop_p()
    Q, K, V = _linear(o, a.WQ.weight, a.WQ.bias), _linear(p_n, a.WK.weight, a.WK.bias), _linear(p_n, a.WV.weight, a.WV.bias)
    causal = -1 if dec.training else None  # carca.py:339
    s = _AttnFn.apply(Q, K, V, o_ids, p_ids, a.H, causal, (p, seed, site) if p > 0 else None, sink, ("cross", index))
    if dec.residual:
        s = s + o
    logit = (s * dec.ffn.weight.view(-1)).sum(dim=-1) + dec.ffn.bias
    return torch.sigmoid(logit)


def forward(model, profile, targets, trace: Optional[dict] = None) -> List[Tensor]:
    """CARCA.forward (carca.py:411-431) for any profile length: the groups' scores [B, N_g] (not yet squeezed / joined)."""
    from .modules import CrossAttentionBlock, cached_parameters, note_training_forward

    emb, dec = model.embeds, model.decoder
    d = emb.d
    p_x = profile[0]
    B, L = p_x.shape
    ops._need_cuda(p_x)
    if L > 1024:
        raise CarcaHipError(f"forward: L={L} > 1024 profile slots (carca_mha_core keeps a query's weights in registers)")
    training = model.training
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in cached_parameters(model))
    seed = ops.new_dropout_seed() if training else 0
    sink = _Sink(len(model.encoder), len(targets)) if getattr(model, "_keep_dropout_masks", False) else None
    segs = tuple([(profile[0], profile[1], profile[2], False)] + [(t[0], t[1], t[2], True) for t in targets])
    if grad:
        if any(t is not None and t.requires_grad for s in segs for t in s[:3]):
            raise CarcaHipError("gradients with respect to the input tensors (ids/attrs/ctx) are not produced")
        note_training_forward()
    # (one embedding call takes the profile and three target groups -- CARCA_MAX_SEGS; further groups go in calls of four)
    es: List[Tensor] = []
    for c0 in range(0, len(segs), _lib.MAX_SEGS):
        chunk = segs[c0: c0 + _lib.MAX_SEGS]
        if grad:
            es += list(_EmbedSegsFn.apply(emb, chunk, *cached_parameters(emb)))
        else:
            dpi = ops.row_ld(d)
            es += [e[..., :d] for e in emb.embed_segments(chunk, ld_e=dpi)[0]]
    x = es[0]
    if trace is not None:
        trace["p_embed"] = x
        for gi in range(len(targets)):
            trace[f"o_embed{gi}"] = es[gi + 1]
    if training and model.dropout.p > 0:  # carca.py:416
        x = _DropoutFn.apply(x, float(model.dropout.p), seed, 1000, sink, ("embed",))
    for i, blk in enumerate(model.encoder):
        blk._check_mode()
        x = sa_block(blk, x, p_x, seed, 4 * i, sink, i)
        if trace is not None:
            trace[f"block{i}"] = x
    p_n = _norm(x, model.norm)  # carca.py:421
    if trace is not None:
        trace["p_final"] = p_n
    if isinstance(dec, CrossAttentionBlock):
        ys = [cross_block(dec, es[gi + 1], targets[gi][0], p_n, p_x, seed, 2000 + gi, sink, gi) for gi in range(len(targets))]
    else:
        ys = list(_DotScoreFn.apply(dec, B, L, d, p_n.reshape(B * L, d), *[e.reshape(-1, d) for e in es[1:]]))
    if sink is not None:
        model._last_dropout_masks = sink.masks
    return ys
