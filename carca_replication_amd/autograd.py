"""torch.autograd glue for the HIP backward pass.

The reference trains with plain torch.autograd over src/carca.py; here one autograd.Function wraps the
whole CARCA.forward (carca.py:411-431) and its backward is a fixed sequence of C-ABI launches
(include/carca_hip.h): attention cores (csrc/backward.hip), dense input/weight gradients
(carca_gemm_rows / carca_gemm_wgrad, csrc/gemm.hip), LayerNorm backward, embedding scatter-add.
Torch only allocates the buffers, keeps the saved tensors alive and routes the returned gradients.
Gradients were checked against the reference's own loss.backward() (fixture G2, tests/test_hip_backward.py).
"""
from __future__ import annotations

from typing import List

import torch
from torch import Tensor

from . import ops
from ._lib import MAX_BLOCKS, MAX_GROUPS, CarcaHipError


BIG_TABLE_BYTES = 64 * 2 ** 20  # embedding tables above this keep ONE gradient buffer, cleared row-wise (see below)


def _grad_buffers(model, params, extra: int, id_lists, late=()):
    """The backward's gradient storage: ONE flat fp32 buffer cut into per-parameter views (every view starts on a
    16-byte boundary), laid out for what happens to the gradients afterwards:

        [ early parameters | late parameters | staging (extra floats) | big tables ]

      early    everything whose gradient is final before the pass's LAST launch; in parameter order
      late     `late`: the parameters that last launch produces (AllEmbedding: feats_embed.weight / .bias, a kernel as
               long as the forward's feature GEMM) -- a sharded step all-reduces the early range UNDER that kernel
               (engine.train_step), then the late range
      staging  the backward's head-padded staging areas (zeroed with the rest, never reduced)
      big      embedding tables of 64 MB or more (BASELINE config 4: 1 M items x 128 = 512 MB): exchanged row-wise between
               ranks, and -- when the ids a pass can touch are known (id_lists) -- never re-zeroed as a whole: the model
               keeps the buffer across steps and a step clears only the table ROWS the previous step's scatter-add touched
               (carca_zero_rows over that step's ids, recorded by `after` with carca_concat_ids) plus the small front.
    The layout is published as model._flat_grad (dist.allreduce_flat reads it).  Returns (views, staging, after).
    Falls back to a fresh zero-filled buffer per step when there is no big table, when the ids are unknown, or when a
    previous gradient still lives in the cached buffer (accumulation without zero_grad(set_to_none=True))."""
    if not params:
        return [], None, (lambda: None)
    # (only the ITEM table is row-sparse: its gradient rows are the batch's ids.  A dense weight of that size -- feats_embed
    # over an attribute vocabulary of 40 k at g = 450 is 64 MB -- needs its whole gradient cleared every step like any other)
    table = getattr(getattr(model, "embeds", model), "items_embed", None)
    tw = table.weight if table is not None else None
    big = [i for i, p in enumerate(params) if p is tw and p.numel() * p.element_size() >= BIG_TABLE_BYTES]
    late_i = [i for i, p in enumerate(params) if any(p is q for q in late) and i not in big]
    early_i = [i for i in range(len(params)) if i not in big and i not in late_i]
    r4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
    offs, total = {}, 0
    for i in early_i:
        offs[i] = total
        total += r4(params[i].numel())
    n_early = total
    for i in late_i:
        offs[i] = total
        total += r4(params[i].numel())
    n_late = total
    total += r4(extra)
    front = total
    for i in big:
        offs[i] = total
        total += r4(params[i].numel())
    dev, dt = params[0].device, params[0].dtype
    cached = bool(big) and id_lists is not None
    c = model.__dict__.get("_grad_cache") if cached else None
    if cached:
        key = (tuple((p.data_ptr(), tuple(p.shape)) for p in params), extra, tuple(t.numel() for t in id_lists), n_early, n_late)
        if c is not None and (c["key"] != key or any(
                p.grad is not None and p.grad.untyped_storage().data_ptr() == c["flat"].untyped_storage().data_ptr()
                for p in params)):
            c = None  # other shapes, or the cached buffer still holds gradients somebody is accumulating into
            model.__dict__.pop("_grad_cache", None)
        if c is None:
            c = dict(key=key, flat=torch.zeros(total, dtype=dt, device=dev),
                     dirty=torch.zeros(sum(t.numel() for t in id_lists), dtype=torch.int32, device=dev), fresh=True)
            model.__dict__["_grad_cache"] = c
        flat = c["flat"]
        if not c["fresh"]:
            flat[:front].zero_()
            # rows to clear: the ones this rank's last pass scattered into, and -- users sharded over ranks -- the ones the
            # row exchange added for the OTHER ranks' users (engine.note_exchanged_rows; found by tools/two_rank_check.py:
            # without them a rank's step-1 gradient still carried the other rank's step-0 rows)
            lists = [c["dirty"]]
            foreign = model.__dict__.get("_grad_foreign")
            if foreign is not None:
                lists.append(foreign)
            for i in big:
                ops.zero_rows(flat[offs[i]: offs[i] + params[i].numel()].view(params[i].shape), lists)
        c["fresh"] = False
    else:
        flat = torch.zeros(total, dtype=dt, device=dev)  # one fill launch for every gradient and staging area
    views = [flat[offs[i]: offs[i] + p.numel()].view(p.shape) for i, p in enumerate(params)]
    model.__dict__["_flat_grad"] = dict(flat=flat, early=(0, n_early), late=(n_early, n_late), front=front,
                                        big=[params[i] for i in big], n_params=len(params))
    # (`big` names the parameters only: a second reference to a big table's view would make AccumulateGrad CLONE it --
    # it takes a gradient over as p.grad only while nobody else holds it -- i.e. a fresh 512 MB allocation + copy per
    # step at BASELINE config 4, and a p.grad that no longer lives in the flat buffer; tests/test_c4_scale.py)

    def after():
        if cached:
            ops.concat_ids(id_lists, c["dirty"])

    return views, flat[n_late:front], after


class _det_pass:
    """Deterministic mode (ops.set_deterministic): the pass's flat gradient buffer gets a zeroed int64 shadow of the same
    length, kept on the model across steps (the flush leaves it zero), and the library accumulates into it instead of
    issuing fp32 atomics (include/carca_hip.h: carca_det_begin).  A model with a big item table sweeps that table's
    range too (dense: the mode trades the touched-row saving for reproducibility)."""

    def __init__(self, model):
        info = model.__dict__["_flat_grad"]
        self.flat, self.info = info["flat"], info
        sh = model.__dict__.get("_det_shadow")
        if sh is None or sh.numel() != self.flat.numel() or sh.device != self.flat.device:
            sh = torch.zeros(self.flat.numel(), dtype=torch.int64, device=self.flat.device)
            model.__dict__["_det_shadow"] = sh
        self.shadow = sh
        ops.det_begin(self.flat, sh)

    def flush_staging(self):
        ops.det_flush(self.flat, self.shadow, self.info["late"][1], self.info["front"])

    def finish(self):
        ops.det_flush(self.flat, self.shadow, 0, self.flat.numel())
        ops.det_begin(None, None)


class _Tail:
    """Hands out zeroed slices of the flat buffer's tail (16-byte aligned sizes)."""

    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def take(self, n: int):
        n4 = (n + 3) // 4 * 4
        if self.buf is None or self.pos + n4 > self.buf.numel():
            return None  # caller allocates (and zeroes) its own
        out = self.buf[self.pos: self.pos + n4]
        self.pos += n4
        return out


# The embedding backward of the TARGET rows beside the encoder backward of the profile rows.  After the decoder's backward
# the pass has two independent halves: d e of the target rows is final (targets reach the loss through the decoder only,
# carca.py:338-349), and everything left on the profile side -- two SelfAttentionBlock backwards -- is a chain of
# one-workgroup-per-user launches that fill half the CUs.  The target rows' embedding backward (two thirds of the pass's
# largest product, d feats_embed) is issued on a second stream right there, with the one-workgroup-per-CU weight-gradient
# kernel held to half the chip (tuning key 10), and the profile rows' embedding backward follows the chain on the first
# stream with the other half.  The second stream sums into buffers of its own (the tail of the flat gradient buffer: zeroed
# by the same fill) that are added at the join: the weight-gradient kernels end in a read-modify-write of dW.
# Off: deterministic mode (its shadow buffer covers the real gradients), sharded steps (the early-gradients event sits
# inside the one call), the re-associated embedding path, other embeddings / decoders.
SPLIT_EMBED_BWD = "graph"  # "graph": while a hipGraph is being captured (engine.GraphedTrainStep) -- an eager step is bound by
                           # its ~40 launches' host time (1.7 ms at C2), which the split's extra launches only add to (eager
                           # steps measured 1.71-2.48 ms with it, 1.79 without); True: always; False: never
SPLIT_SIDE_CUS = 128       # CU budget of the second stream's weight-gradient launch (round 4, after both feats_embed products
                           # left the padding's rows out: 96 / rest 1.213 ms, 128 / all 1.205, 144 / all 1.213, one stream 1.238)
SPLIT_MAIN_CUS = 256       # CU budget of the FIRST stream's weight-gradient launch (the profile rows', issued behind the encoder's
                           # backward chain); 0 = the CUs the second stream's launch leaves
SPLIT_MAIN_TARGET_USERS = 0.25  # share of the FIRST target segment's users left to the first stream, whose last launch takes
                                # every CU once the second stream's is done (C2, round 4: 0.04 / 0.15 / 0.2 / 0.25 / 0.3 / 0.35
                                # -> 1.215 / 1.196 / 1.188 / 1.185 / 1.203 / 1.207 ms per graphed step; round 3: 0 / 0.04 / 0.08
                                # -> 1.654 / 1.642 / 1.655)
SPLIT_MIN_GFLOP = 20.0     # the split pays when d feats_embed is long against the launches it doubles (C2: 71 GFLOP per pass)
SPLIT_TAIL_ON_SIDE = False  # round 5 (MEASURED AND OFF, see below): the weight gradients that feed nothing -- the grouped d x d products of the decoder and the
                           # blocks, d joint_embed -- leave the FIRST stream's chain: they are issued on the second stream behind
                           # its big kernel (which ends first), gated by an event behind the encoder's backward; the first
                           # stream goes from the chain straight into its own rows' embedding backward.  False: round 4's order
SPLIT_TABLE_STREAM = False  # ... and each stream's row table (wgrad_rowtab_kernel: a function of the ids) built on a third
                            # stream beside that stream's d [z ; q] / scatter launches instead of between them and the big kernel.
                            # OFF: with it the C2-sized capture segfaults in hipStreamEndCapture on this runtime (ROCm 7.2; the
                            # fixture-sized one passes) -- the fork is NESTED, a third stream forked off the second one
# Interleaved A/B of the graphed C2 step (tools/ab_train_graph.py, round 5): round 4's order 1.1929 ms; tail on the second stream
# with the first stream's big kernel on 128 CUs and 0.35 / 0.5 / 0.15 of the first target segment's users: 1.1977 / 1.1989 /
# 1.2109; on 192 CUs: 1.2286; on 256: 1.2193-1.2276.  Two big kernels held to half a chip each for their whole length lose more
# (the one that ends first leaves its half idle) than the ~110 us of small products gain by leaving the first stream's chain:
# round 4's order -- second stream's kernel on 128 CUs beside the chain, the first stream's LAST kernel on all 256 -- stays.
SPLIT_TAIL_MAIN_CUS = 128  # CU budget of the first stream's big kernel under SPLIT_TAIL_ON_SIDE
SPLIT_TAIL_MAIN_TARGET_USERS = 0.35  # ... and the share of the first target segment's users it takes (SPLIT_MAIN_TARGET_USERS' role)
_SIDE_STREAMS = {}
_TABLE_STREAMS = {}
import os as _os0  # (A/B switches of the captured schedule from the environment: CARCA_SPLIT_TAIL / CARCA_SPLIT_TABLE = 0 | 1)

SPLIT_TAIL_ON_SIDE = {"0": False, "1": True}.get(_os0.environ.get("CARCA_SPLIT_TAIL", ""), SPLIT_TAIL_ON_SIDE)
SPLIT_TABLE_STREAM = {"0": False, "1": True}.get(_os0.environ.get("CARCA_SPLIT_TABLE", ""), SPLIT_TABLE_STREAM)


def ensure_side_streams(device) -> None:
    """The second / third stream of the captured schedule, created OUTSIDE a capture (engine.GraphedTrainStep calls this before
    it captures).  Creating them lazily worked while the first use sat in the backward; with the prep fork in the FORWARD a
    stream created by the capturing thread inside the capture crashed hipStreamEndCapture on this runtime (segfault; the same
    schedule with the streams created by an eager warm-up step passes)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    if key not in _TABLE_STREAMS:
        _TABLE_STREAMS[key] = torch.cuda.Stream(device=device)


class _SideEmbed:
    def __init__(self, emb, gbp, tail: "_Tail", device):
        self.params = emb.side_grad_params()
        self.ok = True
        self.tmp = {}
        for q in self.params:
            t = tail.take(q.numel())
            if t is None:
                self.ok = False
                return
            self.tmp[id(q)] = t[: q.numel()].view(q.shape)
        self.gbp = dict(gbp)
        self.gbp.update(self.tmp)
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        self.stream = _SIDE_STREAMS[key]
        if key not in _TABLE_STREAMS:
            _TABLE_STREAMS[key] = torch.cuda.Stream(device=device)
        self.table_stream = _TABLE_STREAMS[key].cuda_stream if SPLIT_TABLE_STREAM else None
        self.cus = ops.num_cus()

    @staticmethod
    def floats(emb) -> int:
        return sum((q.numel() + 3) // 4 * 4 for q in emb.side_grad_params())

    def launch(self, emb, des, segs, zq, L, dpi, wj_t) -> None:
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            ops.set_tuning(10, min(SPLIT_SIDE_CUS, self.cus))
            try:
                emb.embed_backward(des, segs, zq, self.gbp, L, dpi, wj_t=wj_t, skip_joint=True, table_stream=self.table_stream)
            finally:
                ops.set_tuning(10, 0)

    def tail(self, fn) -> None:
        """fn() on the second stream once the FIRST stream has got to where it is now (the end of the encoder's backward)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            fn()

    def main_cus(self) -> int:
        if SPLIT_TAIL_ON_SIDE:  # (the second stream still has its tail to run when its big kernel ends: the halves stay halves)
            return min(SPLIT_TAIL_MAIN_CUS, self.cus)
        if SPLIT_MAIN_CUS > 0:
            return min(SPLIT_MAIN_CUS, self.cus)
        return max(8, self.cus - min(SPLIT_SIDE_CUS, self.cus))

    def join(self, gbp) -> None:
        torch.cuda.current_stream().wait_stream(self.stream)
        # one add per run of parameters whose gradients lie back to back in the flat buffer (their temporaries do: same
        # order, same 16-byte rounding) -- feats_embed.weight / .bias and joint_embed.weight / .bias at C2: two launches
        r4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
        runs = []
        for q in self.params:
            real, tmp = gbp[id(q)], self.tmp[id(q)]
            if runs and runs[-1][0] + 4 * runs[-1][2] == real.data_ptr() and runs[-1][1] + 4 * runs[-1][2] == tmp.data_ptr():
                runs[-1][2] += r4(q.numel())
                runs[-1][3] = q.numel() - r4(q.numel())
            else:
                runs.append([real.data_ptr(), tmp.data_ptr(), r4(q.numel()), q.numel() - r4(q.numel()), real, tmp])
        for _, _, n, slack, real, tmp in runs:
            n += slack  # (the last parameter's rounding is not part of its view)
            if n == real.numel():
                real.add_(tmp)
            else:
                real.reshape(-1).as_strided((n,), (1,)).add_(tmp.reshape(-1).as_strided((n,), (1,)))


class _PackPlan:
    """Every transposed weight copy of a backward pass (ONE pack launch) and every head-padded gradient staging area
    (zeroed by the gradient buffer's fill, ONE unpack launch at the end)."""

    def __init__(self):
        self.fw, self.gi, self.greal = [], [], []

    def add_attn(self, attn, extra_fwd: List[ops.PackItem]) -> "_Packs":
        h = _Packs(self, attn, len(self.fw), len(self.gi))
        hp = (h.dh, h.dhp)
        # Bt operands of the input-gradient GEMMs: Bt[n = input feature][k = head-padded output feature] = W[k][n]
        self.fw += [ops.PackItem(m.weight, h.dpi, h.dpo, col_heads=hp, transposed=True)
                    for m in (attn.WQ, attn.WK, attn.WV)] + extra_fwd
        # head-padded staging for d WQ/WK/WV [DPO, d] and their biases [DPO]
        self.gi += [ops.PackItem(m.weight, h.dpo, h.d, row_heads=hp) for m in (attn.WQ, attn.WK, attn.WV)]
        self.gi += [ops.PackItem(m.bias, 1, h.dpo, col_heads=hp) for m in (attn.WQ, attn.WK, attn.WV)]
        self.greal += [m.weight for m in (attn.WQ, attn.WK, attn.WV)] + [m.bias for m in (attn.WQ, attn.WK, attn.WV)]
        return h

    def add_staging(self, item: ops.PackItem, param) -> int:
        self.gi.append(item)
        self.greal.append(param)
        return len(self.gi) - 1

    def staging_floats(self) -> int:
        return ops.PackedWeights.size_of(self.gi) if self.gi else 0

    def build(self, device, tail: "_Tail") -> None:
        if self.fw:
            self.wT = ops.PackedWeights(self.fw, device)
            self.wT.pack()
        if self.gi:
            zeroed = tail.take(self.staging_floats())
            self.g = ops.PackedWeights(self.gi, device, buf=zeroed)
            if zeroed is None:
                self.g.buf.zero_()

    def unpack(self, gbp) -> None:
        if self.gi:
            self.g.unpack_into([gbp[id(p)] for p in self.greal], accumulate=True)


class _Packs:
    """One attention-bearing module's slice of the plan."""

    def __init__(self, plan: _PackPlan, attn, fw_base: int, g_base: int):
        self.plan, self.fw_base, self.g_base = plan, fw_base, g_base
        self.d, self.H = attn.d, attn.H
        self.dpi, self.dhp, self.dpo = ops.padded_dims(attn.d, attn.H)
        self.dh = attn.d // attn.H

    def wT(self, i: int) -> Tensor:
        return self.plan.wT.view(self.fw_base + i)

    def g(self, i: int) -> Tensor:
        return self.plan.g.view(self.g_base + i)


def _cross_decoder_backward(model, st, ys, dys, gbp, wg, cp, d_wpad):
    """Backward of the final LayerNorm + sigmoid/ffn head + cross-attention (carca.py:421, 340-347) as one host call:
    returns (d encoder output, [d o_g])."""
    dec = model.decoder
    d, H = model.embeds.d, dec.attn.H
    dpi = st["dpi"]
    ngroups = st["ngroups"]
    cs = st["csave"]
    cpd = cs["p"]
    masks = cs.get("m_attn") if cpd > 0 else None
    groups = [(cs["qh"][gi], ys[gi], dys[gi], st["segs"][gi + 1][0], st["es"][gi + 1].view(-1, dpi),
               masks[gi] if masks is not None else None) for gi in range(ngroups)]
    g = lambda p: gbp[id(p)]  # noqa: E731
    return ops.cross_score_bwd(
        groups, st["p_x"], cs["kh"], cs["vh"], st["p_normed"], st["enc_out"], (cp.wT(0), cp.wT(1), cp.wT(2)),
        st["cw"].ffn_w_pad, dec.ffn.weight.detach(), model.norm.weight.detach(),
        dict(g_ffn_w=g(dec.ffn.weight), g_ffn_b=g(dec.ffn.bias), g_ffn_w_pad=d_wpad, g_wq=cp.g(0), g_wk=cp.g(1),
             g_wv=cp.g(2), g_bq=cp.g(3), g_bk=cp.g(4), g_bv=cp.g(5), g_norm_w=g(model.norm.weight),
             g_norm_b=g(model.norm.bias)),
        st["B"], st["L"], d, H, dec.residual, st["training"], cpd, wg)


import os as _os

# MEASURED AND OFF (round 5): graphed C2 step 1.1912 ms with it, 1.1919 without -- the fill and the pack (14 us of kernels) were
# not on the critical path for long enough to matter, and the fork is one more thing for a capture to go wrong on.
EARLY_PREP = {"0": False, "1": True, "graph": "graph"}.get(_os.environ.get("CARCA_EARLY_PREP", ""), False)  # the backward's zero fill (gradients + staging) and its pack launch (transposed weight copies) read nothing
                      # the forward produces: "graph" = while a hipGraph is captured they are issued on the second stream at the
                      # START of the forward and joined in front of the backward's first kernel (14 us off the step's critical
                      # path); False = at the start of the backward, as eager steps always do


GRAPH_WARMUP = [False]  # set by engine.GraphedTrainStep around its eager warm-up steps


def _prepare_backward(model, params, st) -> dict:
    """What _CarcaFn.backward sets up before its first kernel: the pack plan (one launch), the flat gradient buffer (one
    fill), the second stream's buffers."""
    from .modules import CrossAttentionBlock

    emb, dec = model.embeds, model.decoder
    d, dpi, B = emb.d, st["dpi"], st["B"]
    dev = st["p_x"].device
    plan = _PackPlan()
    cpk, wpad_idx = None, None
    if isinstance(dec, CrossAttentionBlock):
        cpk = plan.add_attn(dec.attn, [])
        wpad_idx = plan.add_staging(ops.PackItem(dec.ffn.weight, 1, cpk.dpo, col_heads=(cpk.dh, cpk.dhp)),
                                    dec.ffn.weight)
    bpks = []
    for blk in model.encoder:
        dpi_b = ops.padded_dims(d, blk.attn.H)[0]
        bpks.append(plan.add_attn(blk.attn, [ops.PackItem(blk.ffn_1.weight[:, :, 0], dpi_b, dpi_b, transposed=True),
                                             ops.PackItem(blk.ffn_2.weight[:, :, 0], dpi_b, dpi_b, transposed=True)]))
    # the embedding's own transposed copy (joint_embed^T for d [z ; q]) rides in the same pack launch
    emb_wt_idx = None
    if hasattr(emb, "backward_pack_items") and not isinstance(st["emb_saved"], str):
        items = emb.backward_pack_items(dpi)
        if items:
            emb_wt_idx = len(plan.fw)
            plan.fw += items
    # (every id a scatter-add of this pass can touch: known for the embeddings with an item table)
    id_lists = [sg[0] for sg in st["segs"]] if hasattr(emb, "items_embed") and len(st["segs"]) <= 4 else None
    late = emb.late_grad_params(st["emb_saved"]) if hasattr(emb, "late_grad_params") else ()
    split_on = torch.cuda.is_current_stream_capturing() if SPLIT_EMBED_BWD == "graph" else bool(SPLIT_EMBED_BWD)
    may_side = (bool(SPLIT_EMBED_BWD) and st["is_ca"] and emb_wt_idx is not None and not ops.deterministic()
                and ops.early_event is None and hasattr(emb, "side_grad_params") and len(st["segs"]) >= 2)
    if may_side:  # (worth it only for a long d feats_embed product: every latency-bound launch of the call runs twice)
        wf = emb.feats_embed.weight
        rows = sum(sg[0].numel() for sg in st["segs"])
        # ... and while the encoder's backward leaves CUs idle: its kernels run one or two workgroups per user
        may_side = (2.0 * rows * wf.shape[0] * wf.shape[1] >= SPLIT_MIN_GFLOP * 1e9 and 2 * B <= ops.num_cus())
    want_side = may_side and split_on
    r4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
    # (the second stream's buffers are reserved whenever the split COULD be taken: the cached gradient buffer of a model
    # with a big item table is keyed on this size, and the eager warm-up steps of GraphedTrainStep must leave the capture
    # the buffer they used -- a fresh one inside the capture would be zero-filled as a whole by every replay)
    extra = r4(plan.staging_floats()) + (_SideEmbed.floats(emb) if may_side else 0)
    grads, tail_buf, after_pass = _grad_buffers(model, params, extra, id_lists, late)
    det = _det_pass(model) if ops.deterministic() else None
    tail = _Tail(tail_buf)
    plan.build(dev, tail)
    gbp = {id(p): g for p, g in zip(params, grads)}
    side = _SideEmbed(emb, gbp, tail, dev) if want_side else None
    if side is not None and not side.ok:
        side = None
    return dict(plan=plan, cpk=cpk, wpad_idx=wpad_idx, bpks=bpks, emb_wt_idx=emb_wt_idx, grads=grads, after_pass=after_pass,
                det=det, gbp=gbp, side=side)


class _CarcaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, profile, targets, *params):
        from .modules import CrossAttentionBlock

        p_x, p_a, p_c = profile
        emb, dec = model.embeds, model.decoder
        is_ca = isinstance(dec, CrossAttentionBlock)
        d = emb.d
        dpi, _, _ = ops.padded_dims(d, model._heads())
        B, L = p_x.shape
        # train.py:86-88 passes torch.split views: ids are copied (tiny), the dense attrs/ctx views are walked
        # in place by the kernels (CarcaRowSeg.attrs_bstride / CarcaWgradSeg.x_bstride)
        c_ = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
        segs = [(c_(p_x), p_a, p_c, False)] + [(c_(o_x), o_a, o_c, True) for (o_x, o_a, o_c) in targets]
        p_x = segs[0][0]
        if (is_ca and not emb.__dict__.get("_fold_train") and model._fusable() and len(model.encoder) <= MAX_BLOCKS
                and len(targets) <= MAX_GROUPS):
            # the reference architecture: the whole training forward is ONE host call (carca_forward with its training
            # extras: saved tensors, per-block outputs, dropout sites)
            for blk in model.encoder:
                blk._check_mode()
            dec._check_mode()
            # EARLY_PREP: what the backward sets up before its first kernel depends on the weights and the ids alone
            early_box: list = []
            # ("graph": inside the capture AND in the eager warm-up steps GraphedTrainStep runs right before it -- the first
            # pass through this fork must not be the captured one: taken for the first time inside a capture it crashed
            # hipStreamEndCapture on this runtime, taken once eagerly before it the same capture passes)
            want_early = (GRAPH_WARMUP[0] or torch.cuda.is_current_stream_capturing()) if EARLY_PREP == "graph" else bool(EARLY_PREP)

            def fork_prep():
                # (forked BEHIND the forward's first launch -- its pack kernel --, not at the very start of a capture: a
                # stream that waits on an event recorded into a still EMPTY capture crashed hipStreamEndCapture here)
                key = p_x.device.index if p_x.device.index is not None else torch.cuda.current_device()
                if key not in _SIDE_STREAMS:
                    _SIDE_STREAMS[key] = torch.cuda.Stream(device=p_x.device)
                ps = _SIDE_STREAMS[key]
                ps.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(ps):
                    pst = dict(dpi=dpi, B=B, p_x=p_x, segs=segs, emb_saved=None, is_ca=True)
                    prep = _prepare_backward(model, params, pst)
                    ev = torch.cuda.Event()
                    ev.record(ps)
                early_box.append((prep, ev))

            hook = fork_prep if (want_early and p_x.is_cuda and not ops.deterministic() and ops.early_event is None) else None
            tr: dict = {}
            ys = model._forward_fused((p_x, p_a, p_c), [sg[:3] for sg in segs[1:]], train=tr, after_pack=hook)
            early = early_box[0] if early_box else None
            st = dict(p_x=p_x, segs=segs, es=tr["es"], emb_saved=tr["zq"], blocks=tr["blocks"], enc_out=tr["enc_out"],
                      training=model.training, B=B, L=L, m_embed=tr["m_embed"], p_emb=tr["p_emb"], dpi=dpi, is_ca=True,
                      p_normed=tr["p_normed"], csave=tr["csave"], cw=tr["cw"], keep=tr["keep"], ngroups=len(ys))
            if getattr(model, "_keep_dropout_masks", False):  # test hook (see below)
                model._last_dropout_masks = dict(
                    embed=tr["m_embed"], blocks=[{k: v for k, v in b.items() if k.startswith("m_")} for b in tr["blocks"]],
                    cross=tr["csave"].get("m_attn"))
            ctx.model = model
            ctx.params = params
            ctx.st = st
            ctx.prep = ctx.prep_event = None
            if early is not None:  # (the backward's fill + pack, issued on the second stream before the forward's first kernel)
                ctx.prep, ctx.prep_event = early
            joint = getattr(ys, "joint", None)
            if joint is not None and len(ys) > 1:
                # one output tensor [B, sum N] whose column blocks are the groups' scores: autograd then hands the
                # backward ONE gradient tensor, read in place through a row stride (no cat forward, no split copies back)
                st["Ns"] = [int(y.shape[1]) for y in ys]
                ctx.save_for_backward(joint)
                return joint
            ctx.save_for_backward(*ys)
            return tuple(ys)
        if emb.__dict__.get("_fold_train"):  # CARCA.fold_embedding(True, training=True): the re-associated embedding
            es, emb_saved = emb._embed_segments_folded(segs, ld_e=dpi), "folded"
        else:
            es, emb_saved = emb.embed_segments(segs, ld_e=dpi)
        x = es[0]
        # one seed per forward; every dropout site hashes (seed, site id, element index)  (include/carca_hip.h)
        seed = ops.new_dropout_seed() if model.training else 0
        p_emb = float(model.dropout.p) if model.training else 0.0
        m_embed = ops.dropout_fwd(x, d, p_emb, seed, 1000) if p_emb > 0 else None  # carca.py:416
        # every module's weights are repacked at a training step: one pack launch for all of them
        repack: list = []
        sws = [blk.weights_struct(x.device, repack) for blk in model.encoder]
        cw = dec.weights_struct(x.device, model.norm, repack) if is_ca else None
        ops.pack_many(repack)
        blocks = []
        for i, blk in enumerate(model.encoder):
            blk._check_mode()
            bp = blk.drop_p()
            y, saved = ops.sa_block_fwd(x, p_x, sws[i], d, blk.attn.H, blk.residual, save=True,
                                        drop=(bp, seed, 4 * i) if bp > 0 else None)
            saved["x_in"] = x
            saved["p"] = bp
            blocks.append(saved)
            x = y
        st = dict(p_x=p_x, segs=segs, es=es, emb_saved=emb_saved, blocks=blocks, enc_out=x, training=model.training,
                  B=B, L=L, m_embed=m_embed, p_emb=p_emb, dpi=dpi, is_ca=is_ca)
        if is_ca:
            dec._check_mode()
            H = dec.attn.H
            groups = [(es[gi + 1], segs[gi + 1][0]) for gi in range(len(targets))]
            dp_ = dec.drop_p()
            ys, p_normed, csave = ops.cross_score_fwd(x, p_x, groups, cw, d, H, dec.residual, model.training, save=True,
                                                      drop=(dp_, seed, 2000) if dp_ > 0 else None)
            csave["p"] = dp_
            st.update(p_normed=p_normed, csave=csave, cw=cw)
        else:  # dot decoders: stand-alone final LayerNorm (carca.py:421) + row dots (carca.py:352-399)
            p_n = ops.layernorm_fwd(x.view(B * L, -1), model.norm.weight, model.norm.bias, d, dpi)
            ys, dsave = dec.score_groups(p_n, [e.view(-1, dpi) for e in es[1:]], [e.shape[1] for e in es[1:]], B, L, d,
                                         dpi)
            csave = {}
            st.update(dsave=dsave)
        if getattr(model, "_keep_dropout_masks", False):  # test hook: lets a test replay the reference with these masks
            model._last_dropout_masks = dict(embed=m_embed, blocks=[{k: v for k, v in b.items() if k.startswith("m_")}
                                                                    for b in blocks], cross=csave.get("m_attn"))
        st["ngroups"] = len(ys)
        ctx.model = model
        ctx.params = params
        ctx.st = st
        ctx.save_for_backward(*ys)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        model, params, st = ctx.model, ctx.params, ctx.st
        emb, dec = model.embeds, model.decoder
        d = emb.d
        dpi = st["dpi"]
        B, L = st["B"], st["L"]
        p_x = st["p_x"]
        dev = p_x.device
        # every transposed weight copy and every staging area of this pass: one pack launch, one zero fill (shared with
        # the gradients), one unpack launch -- issued HERE, or already under the forward (EARLY_PREP, _prepare_backward)
        prep = ctx.prep if getattr(ctx, "prep", None) is not None else _prepare_backward(model, params, st)
        if getattr(ctx, "prep_event", None) is not None:
            torch.cuda.current_stream().wait_event(ctx.prep_event)
        ctx.prep = ctx.prep_event = None
        plan, cpk, wpad_idx, bpks, emb_wt_idx = prep["plan"], prep["cpk"], prep["wpad_idx"], prep["bpks"], prep["emb_wt_idx"]
        grads, after_pass, det, gbp, side = prep["grads"], prep["after_pass"], prep["det"], prep["gbp"], prep["side"]
        ys = ctx.saved_tensors
        ngroups = st["ngroups"]
        if "Ns" in st:  # joint output: per-group views of the one score / gradient tensor
            dy_all = dys[0] if dys[0] is not None else torch.zeros_like(ys[0])
            if dy_all.stride(1) != 1 or dy_all.stride(0) != ys[0].stride(0):
                dy_all = dy_all.contiguous()
            cuts = [sum(st["Ns"][:i]) for i in range(ngroups + 1)]
            ys = [ys[0][:, cuts[i]: cuts[i + 1]] for i in range(ngroups)]
            dys = [dy_all[:, cuts[i]: cuts[i + 1]] for i in range(ngroups)]
        else:
            dys = [dys[gi].contiguous() if dys[gi] is not None else torch.zeros_like(ys[gi]) for gi in range(ngroups)]
        # the small weight-gradient products feed nothing downstream: collected, then issued as ONE grouped launch
        wg = ops.WgradGroup()
        if st["is_ca"]:
            dx, des_t = _cross_decoder_backward(model, st, ys, dys, gbp, wg, cpk, plan.g.view(wpad_idx).view(-1))
            if side is not None:  # d e of the target rows is final: their embedding backward starts on the second stream
                segs_all, zq_all = st["segs"], st["emb_saved"]
                rows0 = segs_all[0][0].numel()
                side_segs, side_des, main_extra, zq_side = list(segs_all[1:]), list(des_t), None, rows0
                nb_main = int((SPLIT_TAIL_MAIN_TARGET_USERS if SPLIT_TAIL_ON_SIDE else SPLIT_MAIN_TARGET_USERS) *
                              segs_all[1][0].shape[0])
                if 0 < nb_main < segs_all[1][0].shape[0] and len(segs_all) + 1 <= ops._lib.MAX_SEGS:
                    # balance: the FIRST users of the first target segment stay with the profile rows (their [z ; q] rows
                    # follow the profile's in the saved buffer: one call, two segments)
                    x, a, c, tg = segs_all[1]
                    T = x.shape[1]
                    head = lambda t: None if t is None else t[:nb_main]  # noqa: E731
                    rest = lambda t: None if t is None else t[nb_main:]  # noqa: E731
                    main_extra = ((head(x), head(a), head(c), tg), des_t[0][: nb_main * T])
                    side_segs[0] = (rest(x), rest(a), rest(c), tg)
                    side_des[0] = des_t[0][nb_main * T:]
                    zq_side = rows0 + nb_main * T
                side.launch(emb, side_des, side_segs, zq_all[zq_side:], L, dpi, plan.wT.view(emb_wt_idx))
        else:
            dp, des_t = dec.score_backward(dys, st["dsave"], B, L, d, dpi)
            # final LayerNorm (carca.py:421)
            enc_out = st["enc_out"].view(-1, dpi)
            dx = ops.layernorm_bwd(dp, enc_out, model.norm.weight.detach(), d, dpi, dgamma=gbp[id(model.norm.weight)],
                                   dbeta=gbp[id(model.norm.bias)])

        # ---------------- encoder blocks, last to first (carca.py:297-318) --------------------------------
        for blk, sv, bp in zip(reversed(list(model.encoder)), reversed(st["blocks"]), reversed(bpks)):
            # one host call per block (carca_sa_block_bwd): seven launches + five products appended to wg
            x_in = sv["x_in"].view(-1, dpi)
            g = lambda p: gbp[id(p)]  # noqa: E731
            dx = ops.sa_block_bwd(
                dx, p_x, sv, x_in, (bp.wT(0), bp.wT(1), bp.wT(2), bp.wT(3), bp.wT(4)),
                (blk.norm1.weight.detach(), blk.norm2.weight.detach()),
                dict(g_w1=g(blk.ffn_1.weight), g_b1=g(blk.ffn_1.bias), g_w2=g(blk.ffn_2.weight), g_b2=g(blk.ffn_2.bias),
                     g_wq=bp.g(0), g_wk=bp.g(1), g_wv=bp.g(2), g_bq=bp.g(3), g_bk=bp.g(4), g_bv=bp.g(5),
                     g_ln1_w=g(blk.norm1.weight), g_ln1_b=g(blk.norm1.bias), g_ln2_w=g(blk.norm2.weight),
                     g_ln2_b=g(blk.norm2.bias)),
                B, L, d, blk.attn.H, blk.residual, sv["p"], wg)

        # ---------------- embedding (carca.py:85-95 and its ablations) -------------------------------------
        if st["p_emb"] > 0:  # CARCA.dropout on the profile embedding (carca.py:416)
            dx = ops.mask_mul(dx, st["m_embed"], 1.0 / (1.0 - st["p_emb"]), d, dpi)
        des = [dx] + des_t                      # d e per segment, [rows, dpi]; profile rows still unmasked
        if side is not None and SPLIT_TAIL_ON_SIDE:
            wj_t = plan.wT.view(emb_wt_idx)
            m_des, m_segs = [dx], list(st["segs"][:1])
            if main_extra is not None:
                m_segs.append(main_extra[0])
                m_des.append(main_extra[1])
            all_des, all_segs = m_des + side_des, m_segs + side_segs

            def tail():  # (second stream, behind its big kernel: nothing downstream reads these gradients before the join)
                wg.launch()
                plan.unpack(gbp)
                emb.embed_backward(all_des, all_segs, st["emb_saved"], gbp, L, dpi, wj_t=wj_t, only_joint=True)

            side.tail(tail)
            ops.set_tuning(10, side.main_cus())
            try:  # first stream: its own rows' d [z ; q], scatter-add and d feats_embed, straight behind the chain
                emb.embed_backward(m_des, m_segs, st["emb_saved"], gbp, L, dpi, wj_t=wj_t, skip_joint=True,
                                   table_stream=side.table_stream)
            finally:
                ops.set_tuning(10, 0)
            side.join(gbp)
            if det is not None:
                det.finish()
            after_pass()
            ctx.st = None
            return (None, None, None) + tuple(grads)
        wg.launch()
        if det is not None:  # (the staging areas are about to be READ: their accumulated sums out of the shadow first)
            det.flush_staging()
        plan.unpack(gbp)  # head-padded staging areas -> the real WQ / WK / WV / ffn gradients
        if side is not None:
            ops.set_tuning(10, side.main_cus())
            try:
                wj_t = plan.wT.view(emb_wt_idx)
                # (d joint_embed over ALL rows here, in one launch: the second stream's segments ride along joint_only)
                m_des, m_segs = [dx], list(st["segs"][:1])
                if main_extra is not None:
                    m_segs.append(main_extra[0])
                    m_des.append(main_extra[1])
                n_own = len(m_segs)
                m_segs += side_segs
                m_des += side_des
                emb.embed_backward(m_des, m_segs, st["emb_saved"], gbp, L, dpi, wj_t=wj_t,
                                   joint_only=[i >= n_own for i in range(len(m_segs))])
            finally:
                ops.set_tuning(10, 0)
            side.join(gbp)
        elif emb_wt_idx is not None:
            emb.embed_backward(des, st["segs"], st["emb_saved"], gbp, L, dpi, wj_t=plan.wT.view(emb_wt_idx))
        else:
            emb.embed_backward(des, st["segs"], st["emb_saved"], gbp, L, dpi)
        if det is not None:
            det.finish()
        after_pass()
        ctx.st = None
        return (None, None, None) + tuple(grads)


def carca_forward_with_grad(model, profile, targets) -> List[Tensor]:
    from .modules import cached_parameters, note_training_forward

    note_training_forward()  # packed-weight caches: see modules._WEIGHT_EPOCH
    params = cached_parameters(model)
    if any(t is not None and t.requires_grad for t in profile) or \
            any(t is not None and t.requires_grad for grp in targets for t in grp):
        raise CarcaHipError("gradients with respect to the input tensors (ids/attrs/ctx) are not produced")
    out = _CarcaFn.apply(model, tuple(profile), [tuple(g) for g in targets], *params)
    if isinstance(out, Tensor):  # the groups' scores as column blocks of one tensor (modules.JointScores)
        from .modules import JointScores

        ys = JointScores()
        ys.joint = out
        off = 0
        for grp in targets:
            ys.append(out[:, off: off + grp[0].shape[1]])
            off += grp[0].shape[1]
        return ys
    return list(out)


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, ids_mask, eps, denom):
        loss, dy = ops.bce_fwd(y_pred.detach(), y_true, ids_mask, eps, want_grad=True, denom=denom)
        ctx.save_for_backward(dy)
        ctx.shape = y_pred.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dy,) = ctx.saved_tensors
        return dy.view(ctx.shape) * g, None, None, None, None


def bce_with_grad(y_pred, y_true, mask, eps, denom=None, ids=None):
    """ids: the int32 target ids the mask was made from (mask = ids != 0, utils.py:6-7); when given, the float mask and
    its two conversions (three tiny launches) are skipped -- the kernel tests ids != 0 itself."""
    if ids is not None and ids.dtype == torch.int32:
        return _BceFn.apply(y_pred, y_true, ids, eps, denom)
    return _BceFn.apply(y_pred, y_true, mask != 0, eps, denom)


# ----------------------------------------------------------------------------------------------------------------
# torch.autograd through the STAND-ALONE modules (abstract.py:8-50: every block / embedding / encoding of the reference
# is an ordinary differentiable nn.Module).  Training of the hot path goes through _CarcaFn above; these wrap the same
# per-module backward entry points (carca_sa_block_bwd, carca_cross_score_bwd, carca_embed_bwd, ...) for a caller who
# composes the modules himself.  Gradients: parameters, and the activations x / o / p (not ids, attrs, ctx).
# ----------------------------------------------------------------------------------------------------------------
def _standalone_pass(module, params, plan: _PackPlan, id_lists=None, late=()):
    """Gradient buffers (one flat zero-filled buffer, autograd._grad_buffers) + the pack plan's copies for one module."""
    grads, tail_buf, after = _grad_buffers(module, params, plan.staging_floats(), id_lists, late)
    det = _det_pass(module) if ops.deterministic() else None
    plan.build(params[0].device, _Tail(tail_buf))
    return grads, {id(p): g for p, g in zip(params, grads)}, det, after


def _pad_to(t: Tensor, width: int) -> Tensor:
    if t.shape[-1] == width and t.is_contiguous():
        return t
    out = t.new_zeros(*t.shape[:-1], width)
    out[..., : t.shape[-1]] = t
    return out


class _SaBlockFn(torch.autograd.Function):
    """SelfAttentionBlock.forward (carca.py:297-318) with its saved tensors; backward = carca_sa_block_bwd."""

    @staticmethod
    def forward(ctx, module, x, ids, *params):
        d, H = module.attn.d, module.attn.H
        dpi, _, _ = ops.padded_dims(d, H)
        B, L = x.shape[0], x.shape[1]
        xp = _pad_to(x.detach()[..., :d], dpi)
        p = module.drop_p()
        y, saved = ops.sa_block_fwd(xp, ids, module.weights_struct(x.device), d, H, module.residual, save=True,
                                    drop=(p, ops.new_dropout_seed(), 0) if p > 0 else None)
        ctx.module, ctx.params, ctx.saved, ctx.ids, ctx.xp, ctx.p = module, params, saved, ids, xp, p
        ctx.geom = (B, L, d, H, dpi, x.shape[-1])
        return y[..., :d]

    @staticmethod
    def backward(ctx, dy):
        module, params, sv = ctx.module, ctx.params, ctx.saved
        B, L, d, H, dpi, x_cols = ctx.geom
        plan = _PackPlan()
        bp = plan.add_attn(module.attn, [ops.PackItem(module.ffn_1.weight[:, :, 0], dpi, dpi, transposed=True),
                                         ops.PackItem(module.ffn_2.weight[:, :, 0], dpi, dpi, transposed=True)])
        grads, gbp, det, _ = _standalone_pass(module, params, plan)
        g = lambda q: gbp[id(q)]  # noqa: E731
        wg = ops.WgradGroup()
        dx = ops.sa_block_bwd(
            _pad_to(dy, dpi).view(-1, dpi), ctx.ids, sv, ctx.xp.view(-1, dpi), (bp.wT(0), bp.wT(1), bp.wT(2), bp.wT(3), bp.wT(4)),
            (module.norm1.weight.detach(), module.norm2.weight.detach()),
            dict(g_w1=g(module.ffn_1.weight), g_b1=g(module.ffn_1.bias), g_w2=g(module.ffn_2.weight), g_b2=g(module.ffn_2.bias),
                 g_wq=bp.g(0), g_wk=bp.g(1), g_wv=bp.g(2), g_bq=bp.g(3), g_bk=bp.g(4), g_bv=bp.g(5),
                 g_ln1_w=g(module.norm1.weight), g_ln1_b=g(module.norm1.bias), g_ln2_w=g(module.norm2.weight),
                 g_ln2_b=g(module.norm2.bias)),
            B, L, d, H, module.residual, ctx.p, wg)
        wg.launch()
        if det is not None:
            det.flush_staging()
        plan.unpack(gbp)
        if det is not None:
            det.finish()
        dxo = dx.view(B, L, dpi)[..., :d]
        if x_cols != d:
            dxo = _pad_to(dxo, x_cols)
        return (None, dxo, None) + tuple(grads)


def sa_block_with_grad(module, x, mask):
    from .modules import cached_parameters, note_training_forward

    note_training_forward()
    ops._need_cuda(x, mask)
    return _SaBlockFn.apply(module, x, mask != 0, *cached_parameters(module))


class _CrossFn(torch.autograd.Function):
    """CrossAttentionBlock.forward (carca.py:338-349) on an already normed profile; backward = carca_cross_score_bwd
    without its final-LayerNorm stage."""

    @staticmethod
    def forward(ctx, module, o, o_ids, p, p_ids, *params):
        d, H = module.attn.d, module.attn.H
        dpi, _, _ = ops.padded_dims(d, H)
        B, N = o.shape[0], o.shape[1]
        L = p.shape[1]
        op_, pp = _pad_to(o.detach()[..., :d], dpi), _pad_to(p.detach()[..., :d], dpi)
        cw = module.weights_struct(o.device, None)
        pd = module.drop_p()
        (y,), p_normed, csave = ops.cross_score_fwd(pp, p_ids, [(op_, o_ids)], cw, d, H, module.residual, module.training,
                                                    save=True, drop=(pd, ops.new_dropout_seed(), 0) if pd > 0 else None)
        ctx.module, ctx.params, ctx.csave, ctx.cw, ctx.pd = module, params, csave, cw, pd
        ctx.t = (op_, o_ids, p_ids, p_normed, module.training)
        ctx.geom = (B, N, L, d, H, dpi, o.shape[-1], p.shape[-1])
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        module, params, cs = ctx.module, ctx.params, ctx.csave
        (y,) = ctx.saved_tensors
        B, N, L, d, H, dpi, o_cols, p_cols = ctx.geom
        op_, o_ids, p_ids, p_normed, training = ctx.t
        plan = _PackPlan()
        cp = plan.add_attn(module.attn, [])
        wpad = plan.add_staging(ops.PackItem(module.ffn.weight, 1, cp.dpo, col_heads=(cp.dh, cp.dhp)), module.ffn.weight)
        grads, gbp, det, _ = _standalone_pass(module, params, plan)
        g = lambda q: gbp[id(q)]  # noqa: E731
        masks = cs.get("m_attn") if ctx.pd > 0 else None
        wg = ops.WgradGroup()
        dp, des = ops.cross_score_bwd(
            [(cs["qh"][0], y, dy.contiguous(), o_ids, op_.view(-1, dpi), masks[0] if masks is not None else None)], p_ids,
            cs["kh"], cs["vh"], p_normed, None, (cp.wT(0), cp.wT(1), cp.wT(2)), ctx.cw.ffn_w_pad, module.ffn.weight.detach(), None,
            dict(g_ffn_w=g(module.ffn.weight), g_ffn_b=g(module.ffn.bias), g_ffn_w_pad=plan.g.view(wpad).view(-1), g_wq=cp.g(0),
                 g_wk=cp.g(1), g_wv=cp.g(2), g_bq=cp.g(3), g_bk=cp.g(4), g_bv=cp.g(5)),
            B, L, d, H, module.residual, training, ctx.pd, wg)
        wg.launch()
        if det is not None:
            det.flush_staging()
        plan.unpack(gbp)
        if det is not None:
            det.finish()
        do = des[0].view(B, N, dpi)[..., :d]
        dpo_ = dp.view(B, L, dpi)[..., :d]
        return (None, do if o_cols == d else _pad_to(do, o_cols), None, dpo_ if p_cols == d else _pad_to(dpo_, p_cols), None) + \
            tuple(grads)


def cross_with_grad(module, o, o_mask, p, p_mask):
    from .modules import cached_parameters, note_training_forward

    note_training_forward()
    ops._need_cuda(o, o_mask, p, p_mask)
    y = _CrossFn.apply(module, o, o_mask != 0, p, p_mask != 0, *cached_parameters(module))
    return y.squeeze()  # bare squeeze, as carca.py:346


class _EmbedFn(torch.autograd.Function):
    """Embedding.forward (carca.py:85-95 and the ablation variants) for one segment; backward = the module's
    embed_backward (carca_embed_bwd for AllEmbedding)."""

    @staticmethod
    def forward(ctx, module, x, a, c, target, *params):
        d = module.d
        dpi = ops.row_ld(d)
        seg = (x, a, c, bool(target))
        (e,), saved = module.embed_segments([seg], ld_e=dpi)
        ctx.module, ctx.params, ctx.seg, ctx.saved, ctx.dpi = module, params, seg, saved, dpi
        return e[..., :d]

    @staticmethod
    def backward(ctx, de):
        module, params, seg, dpi = ctx.module, ctx.params, ctx.seg, ctx.dpi
        x = seg[0]
        late = module.late_grad_params(ctx.saved) if hasattr(module, "late_grad_params") else ()
        ids = [x] if hasattr(module, "items_embed") else None
        grads, gbp, det, after = _standalone_pass(module, params, _PackPlan(), ids, late)
        module.embed_backward([_pad_to(de, dpi).view(-1, dpi)], [seg], ctx.saved, gbp, x.shape[1], dpi)
        if det is not None:
            det.finish()
        after()
        return (None, None, None, None, None) + tuple(grads)


def embed_with_grad(module, x, a, c, target):
    from .modules import cached_parameters, note_training_forward

    if any(t is not None and t.requires_grad for t in (a, c)):
        raise CarcaHipError("gradients with respect to attrs / ctx are not produced")
    note_training_forward()
    return _EmbedFn.apply(module, x, a, c, target, *cached_parameters(module))


class _AddPosFn(torch.autograd.Function):
    """Encoding.forward (carca.py:25-31, 54-60): x + table[:T]; d table[t] = sum over users of dy[:, t]."""

    @staticmethod
    def forward(ctx, x, table):
        ctx.needs = (x.requires_grad, table.requires_grad)
        return ops.add_positions(x.detach(), table.detach())

    @staticmethod
    def backward(ctx, dy):
        dtab = None
        if ctx.needs[1]:
            B, T, d = dy.shape
            dtab = torch.zeros(T, d, dtype=torch.float32, device=dy.device)
            ops.colsum(dy.contiguous().view(B * T, d), d, dtab, T=T)
        return (dy if ctx.needs[0] else None), dtab


def add_positions_with_grad(x, table):
    return _AddPosFn.apply(x, table)


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the row GEMM (MultiHeadAttention's stand-alone projections, carca.py:238-240)."""

    @staticmethod
    def forward(ctx, x, w, b):
        rows = x.detach().reshape(-1, x.shape[-1])
        (y,) = ops.gemm_rows([dict(a0=ops._f32(rows).contiguous())], w.detach(), w.shape[0], w.shape[1], w.shape[0],
                             bias=b.detach())
        ctx.save_for_backward(x.detach(), w.detach())
        ctx.needs = (x.requires_grad, w.requires_grad, b.requires_grad)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        n_out, n_in = w.shape
        dy2 = dy.contiguous().view(-1, n_out)
        dx = dw = db = None
        if ctx.needs[0]:
            (dx2,) = ops.gemm_rows([dict(a0=dy2)], w.t().contiguous(), n_in, n_out, n_in)
            dx = dx2.view(*x.shape)
        if ctx.needs[1] or ctx.needs[2]:
            dw = torch.zeros_like(w)
            db = torch.zeros(n_out, dtype=torch.float32, device=w.device)
            ops.gemm_wgrad([dict(dy=dy2, x=x.reshape(-1, n_in).contiguous())], n_out, n_in, dw, db)
        return dx, dw, db


class _MhaCoreFn(torch.autograd.Function):
    """The attention core of MultiHeadAttention.forward (carca.py:242-260) on projected q, k, v; differentiable in the
    merged heads AND in the returned weights (return_w hands them out before dropout, carca.py:262-263)."""

    @staticmethod
    def forward(ctx, q, k, v, q_ids, k_ids, H, causal, want_w, drop, module):
        res = ops.mha_core(q.detach(), k.detach(), v.detach(), q_ids, k_ids, H, causal, want_w, drop=drop)
        out, w = res[0], res[1]
        keep = res[2] if len(res) == 3 else None  # nn.Dropout on the weights (carca.py:258): the kernel's keep-mask
        if module is not None:
            module.__dict__["last_keep_mask"] = keep
        ctx.save_for_backward(q.detach(), k.detach(), v.detach())
        ctx.t = (q_ids, k_ids, H, causal, keep, drop[0] if keep is not None else 0.0)
        if want_w:
            return out, w
        ctx.mark_non_differentiable()
        return out, None

    @staticmethod
    def backward(ctx, d_out, d_w):
        q, k, v = ctx.saved_tensors
        q_ids, k_ids, H, causal, keep, p = ctx.t
        if d_out is None and d_w is None:
            return (None,) * 10
        dq, dk, dv = ops.mha_core_bwd(q, k, v, q_ids, k_ids, H, causal, d_out, d_w, keep=keep, p=p)
        return dq, dk, dv, None, None, None, None, None, None, None


def mha_with_grad(module, query, key, value, q_mask, k_mask, causal, return_w, drop=None):
    q = _LinearFn.apply(query, module.WQ.weight, module.WQ.bias)
    k = _LinearFn.apply(key, module.WK.weight, module.WK.bias)
    v = _LinearFn.apply(value, module.WV.weight, module.WV.bias)
    out, w = _MhaCoreFn.apply(q, k, v, q_mask != 0, k_mask != 0, module.H, causal, bool(return_w), drop, module)
    return (w, out) if return_w else out
