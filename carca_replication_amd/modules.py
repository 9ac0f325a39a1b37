"""Host-side mirror of the reference's model surface (src/carca.py, src/abstract.py, src/utils.py).

Same class names, constructor signatures, attribute names (hence identical ``state_dict`` keys and
shapes), same parameter-creation order (hence identical weights under the
same ``torch.manual_seed``) and the same ``forward`` contracts -- but every forward runs the
hand-written gfx950 kernels behind the C ABI (include/carca_hip.h).  There is NO eager/ATen
implementation of the model math in this file: called with CPU tensors, or without the HIP
library, the modules raise ``CarcaHipError``.

reference                                     here
---------                                     ----
get_mask                utils.py:6-7          folded into every kernel as (ids != 0)
AllEmbedding.forward    carca.py:85-95        ops.embed_fwd        (csrc/embed.hip)
SelfAttentionBlock      carca.py:297-318      ops.sa_block_fwd     (csrc/sa_block.hip)
CARCA.norm + CrossAttentionBlock carca.py:421,338-349  ops.cross_score_fwd (csrc/cross_score.hip)
CARCA.forward           carca.py:411-431      CARCA.forward below: 1 pack + 3 embed + n_blocks + 1 launches
BinaryCrossEntropy      carca.py:441-444      ops.bce_fwd          (csrc/loss_metrics.hip)
AttrCtx/Attr/Id/MLPId embeddings carca.py:98-198   the same row GEMM with terms removed (ops.gemm_rows)
DotProduct / WeightedDotProduct  carca.py:352-399  ops.dot_score_fwd (+ layernorm_fwd, slot_decay_scale, l2norm; csrc/decoders.hip)
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from typing import Iterable, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib, ops
from ._lib import CarcaHipError

# ------------------------------------------------------------------------------------------------
# plug-in interfaces (abstract.py:8-50)
# ------------------------------------------------------------------------------------------------


class Model(nn.Module, ABC):
    """abstract.py:8-14"""

    @abstractmethod
    def forward(self, profile: Tuple[Tensor, Tensor, Tensor], targets: List[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
        ...


class Embedding(nn.Module, ABC):
    """abstract.py:17-23"""

    @abstractmethod
    def forward(self, x: Tensor, a: Tensor, c: Tensor, mask: Tensor, target: bool) -> Tensor:
        ...


class Encoding(nn.Module, ABC):
    """abstract.py:26-32"""

    @abstractmethod
    def forward(self, x: Tensor) -> Tensor:
        ...


class Encoder(nn.Module, ABC):
    """abstract.py:35-41"""

    @abstractmethod
    def forward(self, x: Tensor, mask: Tensor) -> Tensor:
        ...


class Decoder(nn.Module, ABC):
    """abstract.py:44-50"""

    @abstractmethod
    def forward(self, o: Tensor, o_mask: Tensor, p: Tensor, p_mask: Tensor) -> Tensor:
        ...


def get_mask(input: Tensor) -> Tensor:
    """utils.py:6-7.  Kept for callers (train.py:45,92); the kernels derive the mask from the ids."""
    return torch.where(input == 0.0, 0.0, 1.0)


def to(*tensors: Tensor, device: str) -> Tuple[Tensor, ...]:
    """utils.py:10-11 (None entries -- the attribute slots of an ids-only batch -- pass through)"""
    return tuple(t if t is None else t.to(device) for t in tensors)


# ------------------------------------------------------------------------------------------------
# encodings (carca.py:15-60): an additive [T, d] table, folded into the embed kernel's epilogue
# ------------------------------------------------------------------------------------------------


def _no_standalone_grad(x: Tensor, module: nn.Module) -> None:
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters())):
        raise CarcaHipError(f"{type(module).__name__}.forward on its own is built for inference (torch.no_grad()): the "
                            "backward pass of the hot path runs through CARCA.forward")


class IdentityEncoding(Encoding):
    def position_table(self, T: int) -> Optional[Tensor]:
        return None

    def forward(self, x: Tensor) -> Tensor:
        return x


class LearnableEncoding(Encoding):
    def __init__(self, d: int, max_len: int):
        super().__init__()
        self.max_len = max_len
        self.encoding = nn.Embedding(max_len, d)
        nn.init.xavier_uniform_(self.encoding.weight)

    def position_table(self, T: int) -> Tensor:
        if T > self.max_len:
            raise CarcaHipError(f"sequence length {T} exceeds LearnableEncoding.max_len={self.max_len}")
        return self.encoding.weight[:T]

    def forward(self, x: Tensor) -> Tensor:
        """Stand-alone use of the ABC (abstract.py:31, carca.py:25-31): x [B, T, d] + encoding[:T].  Inside CARCA the
        table rides in the embedding GEMM's epilogue instead.  Differentiable in x and in the table."""
        table = self.position_table(x.shape[1])
        if torch.is_grad_enabled() and (x.requires_grad or table.requires_grad):
            from .autograd import add_positions_with_grad

            return add_positions_with_grad(x, table)
        return ops.add_positions(x, table)


class PositionalEncoding(Encoding):
    def __init__(self, d_model: int, max_len: int):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def position_table(self, T: int) -> Tensor:
        if T > self.pe.shape[1]:
            raise CarcaHipError(f"sequence length {T} exceeds PositionalEncoding.max_len={self.pe.shape[1]}")
        return self.pe[0, :T]

    def forward(self, x: Tensor) -> Tensor:
        """Stand-alone use of the ABC (abstract.py:31, carca.py:54-60): x [B, T, d] + pe[:, :T]."""
        if torch.is_grad_enabled() and x.requires_grad:
            from .autograd import add_positions_with_grad

            return add_positions_with_grad(x, self.position_table(x.shape[1]))
        return ops.add_positions(x, self.position_table(x.shape[1]))


# ------------------------------------------------------------------------------------------------
# AllEmbedding (carca.py:66-95)
# ------------------------------------------------------------------------------------------------


def _position_table(enc: Encoding, T: int) -> Optional[Tensor]:
    if hasattr(enc, "position_table"):
        return enc.position_table(T)
    raise CarcaHipError(f"encoding {type(enc).__name__} has no position_table(); cannot be fused")


def _segs_pos(enc: Encoding, segs) -> Optional[Tensor]:
    """Position table [T, d] for the profile segment(s) of `segs`, or None (targets never get one, carca.py:91)."""
    if any(not tgt for (_, _, _, tgt) in segs):
        T = next(x.shape[1] for (x, _, _, tgt) in segs if not tgt)
        return _position_table(enc, T)
    return None


def _pos_grad(enc: Encoding, de_profile: Tensor, ids: Tensor, d: int, L: int, gbp) -> None:
    """d LearnableEncoding.encoding.weight[t] += sum over users of (d e * mask)[t]  (carca.py:25-31)."""
    if hasattr(enc, "encoding"):
        ops.colsum(de_profile, d, gbp[id(enc.encoding.weight)], ids=ids, T=L)


def _rows(x: Tensor) -> int:
    return x.numel()


class AllEmbedding(Embedding):
    def __init__(self, n_items: int, d: int, g: int, n_ctx: int, n_attrs: int, enc: Encoding):
        super().__init__()
        self.d = d
        self.enc = enc
        self.items_embed = nn.Embedding(num_embeddings=n_items, embedding_dim=d, padding_idx=0)
        self.feats_embed = nn.Linear(in_features=n_ctx + n_attrs, out_features=g)
        self.joint_embed = nn.Linear(in_features=g + d, out_features=d)
        for m in (self.items_embed, self.feats_embed, self.joint_embed):
            nn.init.xavier_uniform_(m.weight)
        with torch.no_grad():
            self.items_embed.weight[0].zero_()
        nn.init.zeros_(self.feats_embed.bias)
        nn.init.zeros_(self.joint_embed.bias)

    def __getstate__(self):  # keep device-side caches (attribute table, folded weights) out of checkpoints
        state = dict(super().__getstate__())
        state.pop("_attr_table", None)
        state.pop("_fold_cache", None)
        state.pop("_split_cache", None)
        state.pop("_wj_t", None)
        state.pop("_fold_train", None)
        state.pop("_wf_t", None)
        state.pop("_ztab_cache", None)
        return state

    def register_attr_table(self, attrs: Optional[Tensor]) -> None:
        """API-compatible extension: keep the item-attribute matrix [n_items, n_attrs] (row 0 = pad,
        exactly `load_attrs`' output, data.py:28-35) on the device.  Afterwards `a` may be None in forward():
        the rows are gathered by item id inside the feature GEMM, so no dense [B, T, n_attrs] batch tensor is
        built, shipped over PCIe or read from HBM.  Valid because the dataset always sets a = attrs[x]
        (data.py:119-132,167-187).  Pass None to unregister."""
        if attrs is None:
            self.__dict__.pop("_attr_table", None)
            return
        if attrs.dim() != 2 or attrs.shape[1] > self.feats_embed.in_features:
            raise CarcaHipError("register_attr_table: expected [n_items, n_attrs] with n_attrs <= feats_embed.in_features")
        self.__dict__["_attr_table"] = attrs.detach().to(self.items_embed.weight.device, torch.float32).contiguous()

    def attr_table(self) -> Optional[Tensor]:
        return self.__dict__.get("_attr_table")

    def folded_weights(self):
        """(W_c [d, ldw], bias_c [d]) with W_c = W_jq W_f and bias_c = W_jq b_f + b_j, composed on the device with
        carca_gemm_rows and cached per weight version (inference only: see CarcaForwardDesc.fold_wc)."""
        prm = (self.feats_embed.weight, self.feats_embed.bias, self.joint_embed.weight, self.joint_embed.bias)
        key = (_WEIGHT_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in prm)
        cache = self.__dict__.get("_fold_cache")
        if cache is not None and cache[0] == key:
            return cache[1], cache[2]
        Wf, bf, Wj, bj = (p.detach() for p in prm)
        d, g, F = self.d, Wf.shape[0], Wf.shape[1]
        wf_t = ops.PackedWeights([ops.PackItem(Wf, F, g, transposed=True)], Wf.device)  # Bt[n = F][k = g]
        wf_t.pack()
        wjq = Wj[:, d:]  # [d, g] view, row stride d + g
        ldw = ((F + 3) // 4) * 4
        (wc,) = ops.gemm_rows([dict(a0=wjq)], wf_t.view(0), F, g, ldw)
        (bc,) = ops.gemm_rows([dict(a0=wjq, add=bj.view(d, 1))], bf.view(1, g), 1, g, 4)
        bias_c = bc[:, 0].contiguous()
        self.__dict__["_fold_cache"] = (key, wc, bias_c)
        self.__dict__["_wf_t"] = wf_t  # [F, g] transposed copy, reused by the re-associated backward of the same step
        return wc, bias_c

    def z_table(self) -> Tensor:
        """sqrt(d) E W_jz^T, [n_items, d]: the item term of joint_embed (carca.py:87-89) per ITEM instead of per batch row,
        composed on the device with carca_gemm_rows and cached per weight version (inference only: CarcaForwardDesc.z_table
        -- the joint product then runs over q's columns and adds row `id` of this table; no gather launch, no z columns)."""
        prm = (self.items_embed.weight, self.joint_embed.weight)
        key = (_WEIGHT_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in prm)
        cache = self.__dict__.get("_ztab_cache")
        if cache is not None and cache[0] == key:
            return cache[1]
        E, Wj = (p.detach() for p in prm)
        d = self.d
        (zt,) = ops.gemm_rows([dict(a0=E)], Wj[:, :d], d, d, d, alpha=float(d) ** 0.5)
        self.__dict__["_ztab_cache"] = (key, zt)
        return zt

    def _pos(self, T: int) -> Optional[Tensor]:
        return _position_table(self.enc, T)

    def bind_split_weights(self, n_attrs: int) -> None:
        """Opt-in split-precision feature GEMM (ops.set_feature_gemm_precision): hand the library the packed 16-bit planes
        of feats_embed.weight[:, :n_attrs], prepared once per weight version (same key as every other packed copy)."""
        mode = ops.feature_gemm_mode()
        if mode == 0:
            return
        W = self.feats_embed.weight
        key = (_WEIGHT_EPOCH[0], W.data_ptr(), W._version, mode, int(n_attrs))
        cache = self.__dict__.get("_split_cache")
        if cache is None or cache[0] != key:
            cache = (key, ops.split_pack(W, n_attrs, mode))
            self.__dict__["_split_cache"] = cache
        ops.split_bind(W.detach(), cache[1], mode, n_attrs)

    def embed_segments(self, segs, ld_e: int):
        """segs: [(x, a, c, is_target)] -> ([e [B,T,ld_e]], zq).  One fused call for all segments."""
        table = self.attr_table()
        self.bind_split_weights(table.shape[1] if table is not None else segs[0][1].shape[-1])
        pos = _segs_pos(self.enc, segs)
        call = [(x, a, c, (not tgt) and pos is not None) for (x, a, c, tgt) in segs]
        return ops.embed_fwd(call, self.items_embed.weight, self.feats_embed.weight, self.feats_embed.bias,
                             self.joint_embed.weight, self.joint_embed.bias, pos, ld_e, attrs_table=self.attr_table())

    # ---- opt-in re-association for TRAINING (CARCA.fold_embedding(True, training=True)) -------------------------
    # AllEmbedding is linear (carca.py:86-89): e = sqrt(d) E[x] W_jz^T + [a;c] (W_jq W_f)^T + (W_jq b_f + b_j).  Evaluated in
    # that order the F -> g product never happens: the forward needs one F -> d GEMM (14 instead of 71 GFLOP at C2) and
    # the backward one F -> d weight-gradient product G = (de*mask)^T [a;c], from which
    #   d W_f = W_jq^T G,  d b_f = W_jq^T cs,  d W_jq = G W_f^T + cs (x) b_f,  cs = colsum(de*mask) = d b_j
    # are small products.  Same algebra, different fp32 summation order (~1e-6 relative).
    def _embed_segments_folded(self, segs, ld_e: int):
        d = self.d
        wc, bias_c = self.folded_weights()  # recomposed at every training step (the cache key carries the epoch)
        E, Wj = self.items_embed.weight.detach(), self.joint_embed.weight.detach()
        table = self.attr_table()
        pos = _segs_pos(self.enc, segs)
        n_ctx = segs[0][2].shape[-1]
        n_attrs = table.shape[1] if table is not None else segs[0][1].shape[-1]
        es = [torch.empty(x.shape[0], x.shape[1], ld_e, dtype=torch.float32, device=E.device) for (x, _, _, _) in segs]
        outs = [e.view(-1, ld_e) for e in es]
        ops.gemm_rows([dict(a0=E, a0_gather=True, ids=x, out=o) for (x, _, _, _), o in zip(segs, outs)], Wj[:, :d], d, d,
                      ld_e, bias=bias_c, alpha=float(d) ** 0.5)

        def src(a, x):
            return dict(a0=a) if a is not None else dict(a0=table, a0_gather=True)

        ops.gemm_rows([dict(a1=c if n_ctx else None, ids=x, add=o, out=o, add_pos=(not tgt) and pos is not None,
                            T=x.shape[1], **src(a, x)) for (x, a, c, tgt), o in zip(segs, outs)],
                      wc[:, :n_attrs], d, n_attrs, ld_e, bt1=wc[:, n_attrs:] if n_ctx else None, K1=n_ctx, pos=pos,
                      mask_rows=True)
        return es

    def _embed_backward_folded(self, des, segs, gbp, L: int, dpi: int) -> None:
        d = self.d
        E, Wj, Wf, bf = (p.detach() for p in (self.items_embed.weight, self.joint_embed.weight, self.feats_embed.weight,
                                               self.feats_embed.bias))
        g_feats, F = Wf.shape
        table = self.attr_table()
        n_ctx = segs[0][2].shape[-1]
        n_attrs = F - n_ctx
        ids_seg = [sg[0] for sg in segs]
        nseg = len(des)
        if not segs[0][3]:  # (segment 0 is the profile: the only one with a position term, carca.py:91)
            _pos_grad(self.enc, des[0], ids_seg[0], d, L, gbp)
        g_joint_w, g_joint_b = gbp[id(self.joint_embed.weight)], gbp[id(self.joint_embed.bias)]
        # d W_jz = sqrt(d) (de*mask)^T E[x];  d b_j = cs
        ops.gemm_wgrad([dict(dy=des[i], x=E, x_gather=True, ids=ids_seg[i]) for i in range(nseg)], d, d, g_joint_w[:, :d],
                       g_joint_b, mask_rows=True)
        g_joint_w[:, :d].mul_(float(d) ** 0.5)
        # d E[x] += sqrt(d) (de*mask) W_jz
        wjz_t = ops.PackedWeights([ops.PackItem(Wj[:, :d], d, dpi, transposed=True)], des[0].device)
        wjz_t.pack()
        dz = ops.gemm_rows([dict(a0=des[i], ids=ids_seg[i]) for i in range(nseg)], wjz_t.view(0), d, d, d, mask_rows=True)
        g_items = gbp[id(self.items_embed.weight)]
        for i in range(nseg):
            ops.embed_scatter(dz[i], ids_seg[i], d, float(d) ** 0.5, g_items)
        # G = (de*mask)^T [a ; c]   [d, F]
        G = torch.zeros(d, F, dtype=torch.float32, device=des[0].device)

        def xsrc(i):
            a = segs[i][1]
            return dict(x=a) if a is not None else dict(x=table, x_gather=True)

        ops.gemm_wgrad([dict(dy=des[i], ids=ids_seg[i], x1=segs[i][2] if n_ctx else None, **xsrc(i)) for i in range(nseg)],
                       d, n_attrs, G, None, mask_rows=True, K1=n_ctx)
        wjq = Wj[:, d:]  # [d, g] view
        cs = g_joint_b    # colsum(de*mask), complete in stream order
        ops.gemm_wgrad([dict(dy=wjq, x=G)], g_feats, F, gbp[id(self.feats_embed.weight)], None)
        ops.gemm_wgrad([dict(dy=wjq, x=cs.view(d, 1))], g_feats, 1, gbp[id(self.feats_embed.bias)].view(g_feats, 1), None)
        # d W_jq = G W_f^T + cs (x) b_f: 90 rows against K = 4102 is one long dependent chain per block as a row GEMM (150 us),
        # so it runs as a weight-gradient product over the F "rows" of the two transposed operands instead
        g_t = ops.PackedWeights([ops.PackItem(G, F, dpi, transposed=True)], G.device)  # G^T [F, dpi]
        g_t.pack()
        wf_t = self.__dict__["_wf_t"].view(0)                                          # W_f^T [F, g] of this step's forward
        ops.gemm_wgrad([dict(dy=g_t.view(0), x=wf_t)], d, g_feats, g_joint_w[:, d:], None)
        g_joint_w[:, d:].addmm_(cs.view(d, 1), bf.view(1, g_feats))

    def late_grad_params(self, saved):
        """The parameters whose gradient the backward's LAST launch produces (autograd._grad_buffers lays them out behind
        the others so that a sharded step can reduce everything else under that launch)."""
        return () if isinstance(saved, str) else (self.feats_embed.weight, self.feats_embed.bias)

    def side_grad_params(self):
        """The dense parameters embed_backward accumulates into with a read-modify-write at the end of a kernel: a second,
        concurrent embed_backward call of the same pass (autograd._SideEmbed) needs buffers of its own for them.  (The
        item table's rows are scatter-added with atomics: shared; d joint_embed stays in one call over all rows.)"""
        return (self.feats_embed.weight, self.feats_embed.bias)

    def backward_pack_items(self, dpi: int):
        """The transposed weight copy embed_backward needs (Bt[n = input feature of joint_embed][k = output feature]): the
        backward pass packs it in ITS pack launch and hands the view back as embed_backward(..., wj_t=)."""
        return [ops.PackItem(self.joint_embed.weight, self.d + self.feats_embed.weight.shape[0], dpi, transposed=True)]

    def embed_backward(self, des, segs, zq, gbp, L: int, dpi: int, wj_t=None, joint_only=None, skip_joint=False,
                       only_joint=False, table_stream=None) -> None:
        """Backward of embed_segments (carca.py:85-95): des[i] = d e of segment i, [rows, dpi], NOT yet masked;
        accumulates into the gradient buffers gbp[id(param)].  One host call (carca_embed_bwd): position-encoding
        gradient, d joint_embed, d [z ; q], item-row scatter-add, d feats_embed.  joint_only / skip_joint: see
        autograd._SideEmbed (the target rows' share on a second stream)."""
        if isinstance(zq, str):  # the forward took the re-associated path
            return self._embed_backward_folded(des, segs, gbp, L, dpi)
        d = self.d
        g_feats = self.feats_embed.weight.shape[0]
        table = self.attr_table()
        n_attrs = table.shape[1] if table is not None else segs[0][1].shape[-1]
        n_ctx = segs[0][2].shape[-1]
        if wj_t is None:  # (a caller without a pack plan: own copy, own launch)
            pw = self.__dict__.get("_wj_t")  # Bt[n = input feature of joint_embed][k = output feature]: repacked every step
            if pw is None or pw.buf.device != des[0].device or pw.items[0].dst_cols != dpi:
                pw = ops.PackedWeights(self.backward_pack_items(dpi), des[0].device)
                self.__dict__["_wj_t"] = pw
            pw.items[0].src = self.joint_embed.weight
            pw.pack()
            wj_t = pw.view(0)
        enc_w = self.enc.encoding.weight if hasattr(self.enc, "encoding") else None
        ops.embed_bwd(des, segs, zq, wj_t,
                      dict(g_items=gbp[id(self.items_embed.weight)], g_feats_w=gbp[id(self.feats_embed.weight)],
                           g_feats_b=gbp[id(self.feats_embed.bias)], g_joint_w=gbp[id(self.joint_embed.weight)],
                           g_joint_b=gbp[id(self.joint_embed.bias)]),
                      table, d, g_feats, n_attrs, n_ctx, L,
                      # (targets carry no position term; an only_joint call leaves it to the call that owns the profile rows)
                      gbp[id(enc_w)] if (enc_w is not None and not segs[0][3] and not only_joint) else None,
                      joint_only=joint_only, skip_joint=skip_joint, only_joint=only_joint, table_stream=table_stream)

    def _embed_unmasked(self, x: Tensor, a: Optional[Tensor], c: Tensor, target: bool) -> Tensor:
        """carca.py:86-92 WITHOUT line 94: W_j [sqrt(d) E[x] ; W_f [a;c] + b_f] + b_j (+ position term) for every slot, id 0
        included (E[0] is the zero row of padding_idx = 0, carca.py:73) -- what a caller-supplied mask that keeps a slot
        whose id is 0 multiplies.  Three row GEMMs; inference only."""
        d, E = self.d, self.items_embed.weight.detach()
        Wf, bf, Wj, bj = (p.detach() for p in (self.feats_embed.weight, self.feats_embed.bias, self.joint_embed.weight,
                                               self.joint_embed.bias))
        g = Wf.shape[0]
        table = self.attr_table()
        n_ctx = c.shape[-1]
        n_attrs = table.shape[1] if table is not None else a.shape[-1]
        pos = None if target else _position_table(self.enc, x.shape[1])
        ids = ops._ids32(x.reshape(-1))
        src = dict(a0=ops._f32(a).reshape(-1, n_attrs)) if a is not None else dict(a0=table, a0_gather=True, ids=ids)
        (q,) = ops.gemm_rows([dict(a1=ops._f32(c).reshape(-1, n_ctx) if n_ctx else None, **src)], Wf[:, :n_attrs], g, n_attrs,
                             (g + 3) // 4 * 4, bt1=Wf[:, n_attrs:] if n_ctx else None, K1=n_ctx, bias=bf)
        e = torch.empty(x.shape[0], x.shape[1], d, dtype=torch.float32, device=E.device)
        o = e.view(-1, d)
        ops.gemm_rows([dict(a0=E, a0_gather=True, ids=ids, out=o)], Wj[:, :d], d, d, d, bias=bj, alpha=float(d) ** 0.5)
        ops.gemm_rows([dict(a0=q, ids=ids, add=o, out=o, add_pos=pos is not None, T=x.shape[1])], Wj[:, d:], d, g, d, pos=pos)
        return e

    def forward(self, x: Tensor, a: Tensor, c: Tensor, mask: Tensor, target: bool) -> Tensor:
        """abstract.py:22 / carca.py:85-95.  The fused kernels take the mask as x != 0 -- what get_mask(x) is at every call
        site of the reference (carca.py:413-426).  Any OTHER mask is honoured as carca.py:94 does (`e * mask.unsqueeze(2)`):
        see _apply_caller_mask."""
        return _apply_caller_mask(self, x, a, c, mask, target)


# ------------------------------------------------------------------------------------------------
# ablation variants (carca.py:98-198, 352-399) -- SURVEY.md section 8 row f4.  Same constructors and state_dict as
# the reference; the embeddings are AllEmbedding's row GEMM with terms removed, the decoders a row-dot epilogue of
# the final LayerNorm.  Each class brings the two halves the engine calls:
#   embeddings: embed_segments(segs, ld_e) -> (es, saved) and embed_backward(des, segs, saved, gbp, L, dpi)
#   decoders:   score_groups(p, os, B, L, d, dpi) -> (ys, saved) and score_backward(dys, saved, ...) -> (dp, d os)
# ------------------------------------------------------------------------------------------------
def _standalone_embed(module, x, a, c, target):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        from .autograd import embed_with_grad

        return embed_with_grad(module, x, a, c, target)
    (e,), _ = module.embed_segments([(x, a, c, target)], ld_e=module.d)
    return e


def _apply_caller_mask(module, x, a, c, mask, target):
    """Embedding.forward(x, a, c, mask, target) of the ABC (abstract.py:22) with the caller's `mask` honoured
    (carca.py:94,120,149,166,197: `e * mask.unsqueeze(2)`).  The kernels mask with x != 0; so
      * mask is None or equals get_mask(x) (every call of the reference): the fused path as it is;
      * mask is zero wherever x is 0 (it drops or re-weights valid slots): the fused result times the mask -- exact, and
        differentiable like the reference's multiply;
      * mask keeps a slot whose id is 0: the reference embeds that slot (item row 0 + its attributes / context) and scales
        it; AllEmbedding runs its unmasked row products and multiplies (inference); with gradients enabled, or for the
        ablation embeddings, this raises instead of returning another answer than the reference's.
    Stand-alone module surface only (CARCA.forward builds its masks itself): the comparison costs one host read."""
    if mask is None:
        return _standalone_embed(module, x, a, c, target)
    ops._need_cuda(x, mask)
    if mask.shape != x.shape:
        raise ValueError(f"mask {tuple(mask.shape)} must have the shape of the ids {tuple(x.shape)}")
    valid = x != 0
    m = mask.to(torch.float32)
    flags = torch.stack([(m == valid.to(torch.float32)).all(), (m[~valid] == 0).all()]).tolist()  # (one host read)
    if flags[0]:
        return _standalone_embed(module, x, a, c, target)
    if flags[1]:
        return _standalone_embed(module, x, a, c, target) * m.unsqueeze(2)
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())
    if grad or not hasattr(module, "_embed_unmasked"):
        raise CarcaHipError(f"{type(module).__name__}.forward: `mask` keeps slots whose id is 0; the fused kernels leave those "
                            "rows out (x != 0) -- supported for AllEmbedding under torch.no_grad() only")
    return module._embed_unmasked(x, a, c, target) * m.unsqueeze(2)


class _FeatsEmbedding(Embedding):
    """AttrCtxEmbedding / AttrEmbedding: e = (W_j (W_f [a(;c)] + b_f) + b_j (+pos)) * mask  (carca.py:112-121,141-150)."""
    _use_ctx = True

    def _build(self, d: int, g: int, n_in: int, enc: Encoding):
        self.d, self.enc = d, enc
        self.feats_embed = nn.Linear(in_features=n_in, out_features=g)
        self.joint_embed = nn.Linear(in_features=g, out_features=d)
        for m in (self.feats_embed, self.joint_embed):
            nn.init.xavier_uniform_(m.weight)
        for m in (self.feats_embed, self.joint_embed):
            nn.init.zeros_(m.bias)

    def _split(self, segs):
        n_ctx = segs[0][2].shape[-1] if self._use_ctx else 0
        n_attrs = self.feats_embed.in_features - n_ctx
        return n_attrs, n_ctx

    def embed_segments(self, segs, ld_e: int):
        d, Wf, Wj = self.d, self.feats_embed.weight, self.joint_embed.weight
        g = Wf.shape[0]
        n_attrs, n_ctx = self._split(segs)
        g_ld = (g + 3) // 4 * 4
        pos = _segs_pos(self.enc, segs)
        qs = ops.gemm_rows([dict(a0=a, a1=c if n_ctx else None) for (_, a, c, _) in segs], Wf[:, :n_attrs], g, n_attrs,
                           g_ld, bt1=Wf[:, n_attrs:] if n_ctx else None, K1=n_ctx, bias=self.feats_embed.bias)
        es = ops.gemm_rows([dict(a0=qs[i], ids=x, T=x.shape[1], add_pos=(not tgt) and pos is not None)
                            for i, (x, _, _, tgt) in enumerate(segs)], Wj, d, g, ld_e, bias=self.joint_embed.bias,
                           pos=pos, mask_rows=True)
        return [e.view(x.shape[0], x.shape[1], ld_e) for e, (x, _, _, _) in zip(es, segs)], qs

    def embed_backward(self, des, segs, qs, gbp, L: int, dpi: int) -> None:
        d, Wf, Wj = self.d, self.feats_embed.weight, self.joint_embed.weight
        g = Wf.shape[0]
        n_attrs, n_ctx = self._split(segs)
        ids_seg = [sg[0] for sg in segs]
        nseg = len(des)
        if not segs[0][3]:  # (a target segment carries no position term)
            _pos_grad(self.enc, des[0], ids_seg[0], d, L, gbp)
        ops.gemm_wgrad([dict(dy=des[i], x=qs[i], ids=ids_seg[i]) for i in range(nseg)], d, g, gbp[id(Wj)],
                       gbp[id(self.joint_embed.bias)], mask_rows=True)
        wj_t = ops.PackedWeights([ops.PackItem(Wj, g, dpi, transposed=True)], des[0].device)
        wj_t.pack()
        dqs = ops.gemm_rows([dict(a0=des[i], ids=ids_seg[i]) for i in range(nseg)], wj_t.view(0), g, d, g,
                            mask_rows=True)
        ops.gemm_wgrad([dict(dy=dqs[i], x=segs[i][1], x1=segs[i][2] if n_ctx else None) for i in range(nseg)], g,
                       n_attrs, gbp[id(Wf)], gbp[id(self.feats_embed.bias)], K1=n_ctx)

    def forward(self, x, a, c, mask, target):
        return _apply_caller_mask(self, x, a, c, mask, target)


class AttrCtxEmbedding(_FeatsEmbedding):
    def __init__(self, d: int, g: int, n_ctx: int, n_attrs: int, enc: Encoding):
        super().__init__()
        self._build(d, g, n_ctx + n_attrs, enc)


class AttrEmbedding(_FeatsEmbedding):
    _use_ctx = False

    def __init__(self, d: int, g: int, n_attrs: int, enc: Encoding):
        super().__init__()
        self._build(d, g, n_attrs, enc)


class IdEmbedding(Embedding):
    """e = (E[x] * sqrt(d) (+pos)) * mask  (carca.py:164-171): the gather rides a d x d row GEMM against sqrt(d) I, whose
    single non-zero term per output reproduces the reference's one rounding exactly."""

    def __init__(self, n_items: int, d: int, enc: Encoding):
        super().__init__()
        self.d, self.enc = d, enc
        self.items_embed = nn.Embedding(num_embeddings=n_items, embedding_dim=d, padding_idx=0)
        nn.init.xavier_uniform_(self.items_embed.weight)
        with torch.no_grad():
            self.items_embed.weight[0].zero_()

    def _scaled_identity(self, device) -> Tensor:
        eye = self.__dict__.get("_eye")
        if eye is None or eye.device != device:
            eye = torch.eye(self.d, dtype=torch.float32, device=device) * (float(self.d) ** 0.5)
            self.__dict__["_eye"] = eye
        return eye

    def __getstate__(self):
        state = dict(super().__getstate__())
        state.pop("_eye", None)
        return state

    def embed_segments(self, segs, ld_e: int):
        d, E = self.d, self.items_embed.weight
        pos = _segs_pos(self.enc, segs)
        es = ops.gemm_rows([dict(a0=E, a0_gather=True, ids=x, T=x.shape[1], add_pos=(not tgt) and pos is not None)
                            for (x, _, _, tgt) in segs], self._scaled_identity(E.device), d, d, ld_e, pos=pos,
                           mask_rows=True)
        return [e.view(x.shape[0], x.shape[1], ld_e) for e, (x, _, _, _) in zip(es, segs)], None

    def embed_backward(self, des, segs, saved, gbp, L: int, dpi: int) -> None:
        d = self.d
        ids_seg = [sg[0] for sg in segs]
        if not segs[0][3]:  # (a target segment carries no position term)
            _pos_grad(self.enc, des[0], ids_seg[0], d, L, gbp)
        for i in range(len(des)):  # rows with id 0 are skipped: that is the e * mask of carca.py:170
            ops.embed_scatter(des[i], ids_seg[i], d, float(d) ** 0.5, gbp[id(self.items_embed.weight)])

    def forward(self, x, a, c, mask, target):
        return _apply_caller_mask(self, x, a, c, mask, target)


class MLPIdEmbedding(Embedding):
    """e = (W (E[x] * sqrt(d)) + b (+pos)) * mask with a g-wide item table (carca.py:190-198)."""

    def __init__(self, n_items: int, d: int, g: int, enc: Encoding):
        super().__init__()
        self.d, self.enc = d, enc
        self.items_embed = nn.Embedding(num_embeddings=n_items, embedding_dim=g, padding_idx=0)
        self.feats_embed = nn.Linear(in_features=g, out_features=d)
        nn.init.xavier_uniform_(self.items_embed.weight)
        nn.init.xavier_uniform_(self.feats_embed.weight)
        nn.init.zeros_(self.feats_embed.bias)
        with torch.no_grad():
            self.items_embed.weight[0].zero_()

    def embed_segments(self, segs, ld_e: int):
        d, E, W = self.d, self.items_embed.weight, self.feats_embed.weight
        g = E.shape[1]
        pos = _segs_pos(self.enc, segs)
        es = ops.gemm_rows([dict(a0=E, a0_gather=True, ids=x, T=x.shape[1], add_pos=(not tgt) and pos is not None)
                            for (x, _, _, tgt) in segs], W, d, g, ld_e, bias=self.feats_embed.bias, pos=pos,
                           mask_rows=True, alpha=float(d) ** 0.5)
        return [e.view(x.shape[0], x.shape[1], ld_e) for e, (x, _, _, _) in zip(es, segs)], None

    def embed_backward(self, des, segs, saved, gbp, L: int, dpi: int) -> None:
        d, E, W = self.d, self.items_embed.weight, self.feats_embed.weight
        g = E.shape[1]
        sd = float(d) ** 0.5
        ids_seg = [sg[0] for sg in segs]
        nseg = len(des)
        if not segs[0][3]:  # (a target segment carries no position term)
            _pos_grad(self.enc, des[0], ids_seg[0], d, L, gbp)
        # d W = sqrt(d) (d e * mask)^T E[x]; the table rows are gathered inside the product
        gw = torch.zeros_like(gbp[id(W)])
        ops.gemm_wgrad([dict(dy=des[i], x=E.detach(), x_gather=True, ids=ids_seg[i]) for i in range(nseg)], d, g, gw,
                       gbp[id(self.feats_embed.bias)], mask_rows=True)
        gbp[id(W)].add_(gw, alpha=sd)
        w_t = ops.PackedWeights([ops.PackItem(W, g, dpi, transposed=True)], des[0].device)
        w_t.pack()
        dzs = ops.gemm_rows([dict(a0=des[i], ids=ids_seg[i]) for i in range(nseg)], w_t.view(0), g, d, g,
                            mask_rows=True)
        for i in range(nseg):
            ops.embed_scatter(dzs[i], ids_seg[i], g, sd, gbp[id(E)])

    def forward(self, x, a, c, mask, target):
        return _apply_caller_mask(self, x, a, c, mask, target)


class _DotDecoder(Decoder):
    """Row-dot decoders over the final-normed profile (carca.py:352-399).  Neither looks at the masks."""

    def _standalone(self, o, p):
        if torch.is_grad_enabled() and (o.requires_grad or p.requires_grad):
            raise CarcaHipError(f"training through a stand-alone {type(self).__name__} is not built; train through "
                                f"CARCA.forward")
        B, L, d = p.shape
        ys, _ = self.score_groups(p.reshape(B * L, d), [o.reshape(-1, d)], [o.shape[1]], B, L, d, d)
        return ys[0]


class DotProduct(_DotDecoder):
    def __init__(self) -> None:
        super().__init__()
        self.sig = nn.Sigmoid()

    def score_groups(self, p2d: Tensor, os2d: List[Tensor], Ts: List[int], B: int, L: int, d: int, dpi: int):
        """y_g = sigmoid(p . o): slot t against target t in train mode, last slot against all in eval (carca.py:361-367)."""
        slot = self.training
        ys = [ops.dot_score_fwd(p2d, o, B, L, T, d, slot, 0) for o, T in zip(os2d, Ts)]
        return ys, dict(p=p2d, os=os2d, Ts=Ts, ys=ys, slot=slot)

    def score_backward(self, dys, sv, B: int, L: int, d: int, dpi: int):
        dp = torch.zeros(B * L, dpi, dtype=torch.float32, device=sv["p"].device)
        dos = [ops.dot_score_bwd(sv["p"], o, y, dy, dp, B, L, T, d, sv["slot"], 0, dpi)
               for o, y, dy, T in zip(sv["os"], sv["ys"], dys, sv["Ts"])]
        return dp, dos

    def forward(self, o, o_mask, p, p_mask):
        return self._standalone(o, p)


class WeightedDotProduct(_DotDecoder):
    def __init__(self, gamma: float, seq_len: int, normalize: bool, device: str):
        super().__init__()
        self.norm = normalize
        self.gamma = float(gamma)
        self.W = (gamma ** torch.arange(0, seq_len, device=device).unsqueeze(0).repeat(seq_len, 1)).tril().unsqueeze(-1)
        self.sig = nn.Sigmoid()

    def score_groups(self, p2d: Tensor, os2d: List[Tensor], Ts: List[int], B: int, L: int, d: int, dpi: int):
        """p' = p[t] * sum_{j<=t} gamma^j (what the reference's repeat / tril / sum does, carca.py:385-386); optional L2
        normalisation of p' and o; y = sigmoid(p' . o) or (p' . o + 1) / 2 (carca.py:388-397)."""
        slot = self.training
        ld = max(dpi, d)
        pw = ops.slot_decay_scale(p2d, B, L, d, self.gamma, ld)
        if self.norm:
            pq = ops.l2norm_fwd(pw, d, ld)
            oq = [ops.l2norm_fwd(o, d, ld) for o in os2d]
        else:
            pq, oq = pw, os2d
        link = 1 if self.norm else 0
        ys = [ops.dot_score_fwd(pq, o, B, L, T, d, slot, link) for o, T in zip(oq, Ts)]
        return ys, dict(pw=pw, pq=pq, os=os2d, oq=oq, Ts=Ts, ys=ys, slot=slot, link=link)

    def score_backward(self, dys, sv, B: int, L: int, d: int, dpi: int):
        dpq = torch.zeros(B * L, dpi, dtype=torch.float32, device=sv["pq"].device)
        doq = [ops.dot_score_bwd(sv["pq"], o, y, dy, dpq, B, L, T, d, sv["slot"], sv["link"], dpi)
               for o, y, dy, T in zip(sv["oq"], sv["ys"], dys, sv["Ts"])]
        if self.norm:
            dpw = ops.l2norm_bwd(sv["pw"], dpq, d, dpi)
            dos = [ops.l2norm_bwd(o, g, d, dpi) for o, g in zip(sv["os"], doq)]
        else:
            dpw, dos = dpq, doq
        return ops.slot_decay_scale(dpw, B, L, d, self.gamma, dpi), dos  # the slot weights are diagonal: own transpose

    def forward(self, o, o_mask, p, p_mask):
        return self._standalone(o, p)


# ------------------------------------------------------------------------------------------------
# attention (carca.py:204-349)
# ------------------------------------------------------------------------------------------------


class MultiHeadAttention(nn.Module):
    """Parameter holder with the reference's names (carca.py:204-226).

    Its arithmetic (carca.py:228-265) runs fused inside SelfAttentionBlock / CrossAttentionBlock.
    """

    def __init__(self, embed_dim: int, num_heads: int, dropout: float):
        super().__init__()
        assert embed_dim % num_heads == 0.0, "Embedding dim must be divisible by number of heads"
        self.d = embed_dim
        self.H = num_heads
        self.WQ = nn.Linear(in_features=embed_dim, out_features=embed_dim)
        self.WK = nn.Linear(in_features=embed_dim, out_features=embed_dim)
        self.WV = nn.Linear(in_features=embed_dim, out_features=embed_dim)
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(p=dropout)
        for m in (self.WQ, self.WK, self.WV):
            nn.init.xavier_uniform_(m.weight)
        for m in (self.WQ, self.WK, self.WV):
            nn.init.zeros_(m.bias)

    def pack_items(self, dpi: int, dhp: int, dpo: int) -> List[ops.PackItem]:
        dh = self.d // self.H
        mats = [ops.PackItem(m.weight, dpo, dpi, row_heads=(dh, dhp), frag16=True) for m in (self.WQ, self.WK, self.WV)]
        vecs = [ops.PackItem(m.bias, 1, dpo, col_heads=(dh, dhp)) for m in (self.WQ, self.WK, self.WV)]
        return mats + vecs

    def forward(self, query, key, value, q_mask, k_mask, causal: int = None, return_w: bool = False):
        """Stand-alone MultiHeadAttention.forward (carca.py:228-265): three row GEMMs for the projections and
        carca_mha_core for the attention (the blocks' fused kernels never come here).  Returns the merged heads
        [B, Tq, d], or (weights [H*B, Tq, Tk] before dropout, output) with return_w -- the reference's order.
        Differentiable (autograd.mha_with_grad: carca_mha_core_bwd + row / weight-gradient GEMMs).
        Train mode with p > 0 (carca.py:258, self.dropout on the weights): the masks are drawn inside carca_mha_core_drop
        from a seed taken from torch's CPU generator (torch's Philox stream cannot be reproduced in a kernel); the keep-mask
        of the last call, uint8 [B, H, Tq, Tk], stays on the module as `last_keep_mask` for replay (tests/test_hip_dropout.py)."""
        ops._need_cuda(query, key, value, q_mask, k_mask)
        drop = None
        self.__dict__["last_keep_mask"] = None
        if self.training and self.dropout.p > 0:
            if self.dropout.p >= 1:
                raise CarcaHipError("MultiHeadAttention: dropout p must be < 1")
            drop = (float(self.dropout.p), ops.new_dropout_seed(), 0)
        if torch.is_grad_enabled() and (any(t.requires_grad for t in (query, key, value)) or
                                        any(p.requires_grad for p in self.parameters())):
            from .autograd import mha_with_grad

            return mha_with_grad(self, query, key, value, q_mask, k_mask, causal, return_w, drop=drop)
        d = self.d
        proj = []
        for x, lin in ((query, self.WQ), (key, self.WK), (value, self.WV)):
            x2 = ops._f32(x).reshape(-1, x.shape[-1])
            (y,) = ops.gemm_rows([dict(a0=x2)], lin.weight.detach(), d, d, d, bias=lin.bias.detach())
            proj.append(y.view(x.shape[0], x.shape[1], d))
        res = ops.mha_core(proj[0], proj[1], proj[2], q_mask != 0, k_mask != 0, self.H, causal, return_w, drop=drop)
        out, w = res[0], res[1]
        if len(res) == 3:
            self.__dict__["last_keep_mask"] = res[2]
        return (w, out) if return_w else out


# Packed / composed weight caches are keyed on (data_ptr, _version) of their parameters -- but not every optimizer bumps
# _version (torch.optim.Adam(fused=True) updates parameters without touching it), so the key also carries an epoch
# that every training-mode forward advances: a training forward always repacks (weights change every step anyway) and
# the first inference forward after training repacks once; inference loops keep their caches.
_WEIGHT_EPOCH = [0]
USE_Z_TABLE = True  # inference: the joint embedding's item term from AllEmbedding.z_table() (False: gather + d + g columns; A/B)


def note_training_forward() -> None:
    _WEIGHT_EPOCH[0] += 1


def cached_parameters(module: nn.Module) -> List[nn.Parameter]:
    """list(module.parameters()) without walking the module tree on every call (0.25 ms per train step at C2).
    The cached list is revalidated against the tree it was built from -- every _parameters / _modules dict of the
    subtree must still hold the same objects -- so replacing a parameter or a sub-module is picked up."""
    c = module.__dict__.get("_param_cache")
    if c is not None:
        sizes, checks, params = c
        if all(len(d) == n for d, n in sizes) and all(d.get(k) is o for d, k, o in checks):
            return params
    sizes, checks = [], []
    for m in module.modules():
        for d in (m._parameters, m._modules):
            sizes.append((d, len(d)))
            checks.extend((d, k, o) for k, o in d.items())
    params = list(module.parameters())
    module.__dict__["_param_cache"] = (sizes, checks, params)
    return params


class _PackedModule:
    """Mixin: (re)packs this module's parameters into the kernels' layout when any of them changed."""

    def _packed(self, items_fn, device, defer: Optional[list] = None) -> ops.PackedWeights:
        """defer: a list the caller flushes with ops.pack_many (several modules, one launch); None = pack now."""
        key = ((_WEIGHT_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in self._pack_params()), self._pack_shape(),
               str(device))
        cache = self.__dict__.get("_pack_cache")
        if cache is not None and cache[0] == key:
            return cache[1]
        if cache is not None and cache[0][1:] == key[1:]:
            pw = cache[1]  # same geometry: refill the same device buffer
            pw.items = items_fn()
        else:
            pw = ops.PackedWeights(items_fn(), device)
        if defer is None:
            pw.pack()
        else:
            defer.append(pw)
        self.__dict__["_pack_cache"] = (key, pw)
        return pw

    def __getstate__(self):  # torch.save(model) pickles whole modules (train.py:124): drop the ctypes caches
        state = dict(super().__getstate__())
        for k in ("_pack_cache", "_final_norm_params", "_plan", "_fold_cache", "_param_cache", "_split_cache"):
            state.pop(k, None)
        return state


class SelfAttentionBlock(_PackedModule, Encoder):
    def __init__(self, d: int, H: int, p: float, residual: bool):
        super().__init__()
        self.residual = residual
        self.norm1 = nn.LayerNorm(normalized_shape=d)
        self.attn = MultiHeadAttention(embed_dim=d, num_heads=H, dropout=p)
        self.norm2 = nn.LayerNorm(normalized_shape=d)
        self.ffn_1 = nn.Conv1d(in_channels=d, out_channels=d, kernel_size=1)
        self.lrelu = nn.LeakyReLU()
        self.dropout1 = nn.Dropout(p=p)
        self.ffn_2 = nn.Conv1d(in_channels=d, out_channels=d, kernel_size=1)
        self.dropout2 = nn.Dropout(p=p)
        nn.init.xavier_uniform_(self.ffn_1.weight)
        nn.init.xavier_uniform_(self.ffn_2.weight)
        nn.init.zeros_(self.ffn_1.bias)
        nn.init.zeros_(self.ffn_2.bias)

    # -- packing -------------------------------------------------------------------------------
    def _pack_params(self):
        return cached_parameters(self)

    def _pack_shape(self):
        return (self.attn.d, self.attn.H)

    def _items(self) -> List[ops.PackItem]:
        d, H = self.attn.d, self.attn.H
        dpi, dhp, dpo = ops.padded_dims(d, H)
        vec = lambda t: ops.PackItem(t, 1, dpi)  # noqa: E731
        return ([vec(self.norm1.weight), vec(self.norm1.bias), vec(self.norm2.weight), vec(self.norm2.bias)]
                + self.attn.pack_items(dpi, dhp, dpo)
                + [ops.PackItem(self.ffn_1.weight[:, :, 0], dpi, dpi, frag16=True),
                   ops.PackItem(self.ffn_2.weight[:, :, 0], dpi, dpi, frag16=True),
                   vec(self.ffn_1.bias), vec(self.ffn_2.bias)])

    def weights_struct(self, device, defer: Optional[list] = None) -> "_lib.SaWeights":
        pw = self._packed(self._items, device, defer)
        w = _lib.SaWeights()
        names = ["ln1_w", "ln1_b", "ln2_w", "ln2_b", "wq", "wk", "wv", "bq", "bk", "bv", "w1", "w2", "b1", "b2"]
        for i, n in enumerate(names):
            setattr(w, n, pw.ptr(i))
        w._keepalive = pw
        return w

    def _check_mode(self):
        ps = {self.attn.dropout.p, self.dropout1.p, self.dropout2.p}
        if self.training and len(ps) != 1:
            raise CarcaHipError("the fused block kernel takes ONE dropout probability for its three sites "
                                "(the reference constructs them all from the same p, carca.py:280,286,289)")

    def drop_p(self) -> float:
        return float(self.attn.dropout.p) if self.training else 0.0

    def forward(self, x: Tensor, mask: Tensor) -> Tensor:
        """x [B, L, >=d], mask [B, L] (0 = pad) -> [B, L, d] (a view of a padded buffer)."""
        self._check_mode()
        if x.shape[1] > _lib.MAX_L or self.attn.d > ops.FUSED_MAX_D:
            from . import long_profile

            return long_profile.sa_block(self, x[..., : self.attn.d], mask != 0, ops.new_dropout_seed() if self.training else 0)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import sa_block_with_grad

            return sa_block_with_grad(self, x, mask)
        d = self.attn.d
        p = self.drop_p()
        y = ops.sa_block_fwd(x, mask != 0, self.weights_struct(x.device), d, self.attn.H, self.residual,
                             drop=(p, ops.new_dropout_seed(), 0) if p > 0 else None)
        return y[..., :d]


class CrossAttentionBlock(_PackedModule, Decoder):
    def __init__(self, d: int, H: int, p: float, residual: bool):
        super().__init__()
        self.residual = residual
        self.attn = MultiHeadAttention(embed_dim=d, num_heads=H, dropout=p)
        self.ffn = nn.Linear(in_features=d, out_features=1)
        self.sig = nn.Sigmoid()
        nn.init.xavier_uniform_(self.ffn.weight)
        nn.init.zeros_(self.ffn.bias)

    def _pack_params(self):
        extra = list(self.__dict__.get("_final_norm_params", ()))
        return cached_parameters(self) + extra

    def _pack_shape(self):
        return (self.attn.d, self.attn.H, len(self.__dict__.get("_final_norm_params", ())))

    def _items(self) -> List[ops.PackItem]:
        d, H = self.attn.d, self.attn.H
        dpi, dhp, dpo = ops.padded_dims(d, H)
        dh = d // H
        items = self.attn.pack_items(dpi, dhp, dpo)
        items += [ops.PackItem(self.ffn.weight, 1, dpo, col_heads=(dh, dhp)), ops.PackItem(self.ffn.weight, 1, dpi),
                  ops.PackItem(self.ffn.bias, 1, 4)]
        # decoder.ffn folded into the value projection (CarcaCaWeights.wu / cu): wu[h] = sum_i w[h,i] W_V[h,i,:],
        # cu[h] = sum_i w[h,i] b_V[h,i] -- the inference kernel scores with P . u instead of w . (P V)
        fv = self.ffn.weight.view(-1)
        items += [ops.PackItem(self.attn.WV.weight, 16, dpi, frag16=True, fold_vec=fv, fold_H=H),
                  ops.PackItem(self.attn.WV.bias.view(-1, 1), 16, 1, fold_vec=fv, fold_H=H)]
        for t in self.__dict__.get("_final_norm_params", ()):
            items.append(ops.PackItem(t, 1, dpi))
        return items

    def weights_struct(self, device, final_norm: Optional[nn.LayerNorm], defer: Optional[list] = None) -> "_lib.CaWeights":
        # the final LayerNorm of CARCA (carca.py:421) is fused into this kernel's prologue
        self.__dict__["_final_norm_params"] = (final_norm.weight, final_norm.bias) if final_norm is not None else ()
        pw = self._packed(self._items, device, defer)
        w = _lib.CaWeights()
        for i, n in enumerate(["wq", "wk", "wv", "bq", "bk", "bv", "ffn_w_pad", "ffn_w", "ffn_b"]):
            setattr(w, n, pw.ptr(i))
        w.wu, w.cu = pw.ptr(9), pw.ptr(10)
        if final_norm is not None:
            w.ln_w, w.ln_b = pw.ptr(11), pw.ptr(12)
        else:
            w.ln_w, w.ln_b = None, None
        w._keepalive = pw
        return w

    def _check_mode(self):
        pass

    def drop_p(self) -> float:
        return float(self.attn.dropout.p) if self.training else 0.0

    def forward(self, o: Tensor, o_mask: Tensor, p: Tensor, p_mask: Tensor) -> Tensor:
        """Standalone decoder call: p is already final-normed (as in carca.py:421-428)."""
        self._check_mode()
        if p.shape[1] > _lib.MAX_L or self.attn.d > ops.FUSED_MAX_D:
            from . import long_profile

            d = self.attn.d
            return long_profile.cross_block(self, o[..., :d], o_mask != 0, p[..., :d], p_mask != 0,
                                            ops.new_dropout_seed() if self.training else 0).squeeze()
        if torch.is_grad_enabled() and (o.requires_grad or p.requires_grad or
                                        any(q.requires_grad for q in self.parameters())):
            from .autograd import cross_with_grad

            return cross_with_grad(self, o, o_mask, p, p_mask)
        d, H = self.attn.d, self.attn.H
        dpi, _, _ = ops.padded_dims(d, H)
        o_pad = _pad_cols(o, dpi)
        pd = self.drop_p()
        if pd > 0:  # the keep-masks travel with the saved tensors
            (y,), _, _ = ops.cross_score_fwd(p, p_mask != 0, [(o_pad, o_mask != 0)], self.weights_struct(o.device, None),
                                             d, H, self.residual, self.training, save=True,
                                             drop=(pd, ops.new_dropout_seed(), 0))
        else:
            (y,), _ = ops.cross_score_fwd(p, p_mask != 0, [(o_pad, o_mask != 0)], self.weights_struct(o.device, None),
                                          d, H, self.residual, self.training)
        return y.squeeze()  # bare squeeze, as carca.py:346


class JointScores(list):
    """The per-group score tensors of one forward, views of `.joint` [B, sum N] (column blocks in group order)."""
    joint: Optional[Tensor] = None


def _pad_cols(t: Tensor, width: int) -> Tensor:
    """[.., w] -> contiguous [.., width] with zero pad columns (data movement only)."""
    if t.shape[-1] == width and t.is_contiguous():
        return t
    out = t.new_zeros(*t.shape[:-1], width)
    out[..., : t.shape[-1]] = t
    return out


# ------------------------------------------------------------------------------------------------
# CARCA (carca.py:401-431)
# ------------------------------------------------------------------------------------------------


class CARCA(_PackedModule, Model):
    def __init__(self, d: int, p: float, emb: Embedding, enc: Iterable[Encoder], dec: Decoder):
        super().__init__()
        self.embeds = emb
        self.dropout = nn.Dropout(p=p)
        self.encoder = enc
        self.norm = nn.LayerNorm(normalized_shape=d)
        self.decoder = dec

    def _fusable(self) -> bool:
        """The single-call inference path (carca_forward) covers AllEmbedding + SelfAttentionBlocks + CrossAttentionBlock."""
        return (isinstance(self.embeds, AllEmbedding) and isinstance(self.decoder, CrossAttentionBlock)
                and all(isinstance(b, SelfAttentionBlock) for b in self.encoder))

    def _check_built(self) -> None:
        ok_emb = hasattr(self.embeds, "embed_segments")
        ok_dec = isinstance(self.decoder, CrossAttentionBlock) or hasattr(self.decoder, "score_groups")
        if not (ok_emb and ok_dec and all(isinstance(b, SelfAttentionBlock) for b in self.encoder)):
            raise CarcaHipError("CARCA.forward is built for the reference's embeddings (All/AttrCtx/Attr/Id/MLPId), "
                                "SelfAttentionBlock encoders and its decoders (CrossAttentionBlock/DotProduct/"
                                "WeightedDotProduct); there is no ATen fallback for anything else")

    def _heads(self) -> int:
        if len(self.encoder):
            return self.encoder[0].attn.H
        return self.decoder.attn.H if isinstance(self.decoder, CrossAttentionBlock) else 1

    def forward(self, profile: Tuple[Tensor, Tensor, Tensor], targets: List[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
        self._check_built()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in cached_parameters(self))
        if profile[0].shape[1] > _lib.MAX_L or len(targets) > _lib.MAX_GROUPS or self.embeds.d > ops.FUSED_MAX_D:
            # longer than the fused kernels' 64 profile slots (or more target groups than one fused call takes, carca.py:424):
            # the same arithmetic from the row-level kernels
            from . import long_profile

            ys = long_profile.forward(self, profile, targets)
        elif needs_grad:
            from .autograd import carca_forward_with_grad

            ys = carca_forward_with_grad(self, profile, targets)
        else:
            ys = self.forward_nograd(profile, targets)
        joint = getattr(ys, "joint", None)
        if joint is not None and isinstance(self.decoder, CrossAttentionBlock) and joint.shape[0] > 1 and \
                all(y.shape[1] > 1 for y in ys):
            return joint  # = torch.cat([y.squeeze() for y in ys], -1): no size-1 dimension for the squeeze to drop
        # each group's scores are squeezed the way CrossAttentionBlock does (carca.py:346), then joined (carca.py:431);
        # the dot decoders return [B, T] unsqueezed (carca.py:361-367)
        if isinstance(self.decoder, CrossAttentionBlock):
            ys = [y.squeeze() for y in ys]
        if len(ys) == 1 and not needs_grad:
            return ys[0]  # (torch.cat of one tensor is a copy: the scores were allocated by this call, hand them out as is)
        return torch.cat(ys, dim=-1)

    def fold_embedding(self, on: bool = True, training: bool = False) -> "CARCA":
        """Opt-in shortcut: compose AllEmbedding's two Linear layers into one (include/carca_hip.h,
        CarcaForwardDesc.fold_wc).  Same algebra, ~1e-6 relative fp32 re-association.  Inference only unless
        training=True, which also re-associates the training step (AllEmbedding._embed_segments_folded: no F -> g
        product in the forward or in the backward)."""
        self.__dict__["_fold"] = bool(on)
        if isinstance(self.embeds, AllEmbedding):
            self.embeds.__dict__["_fold_train"] = bool(on) and bool(training)
        return self

    # ---- inference: one host call per forward (include/carca_hip.h: carca_forward) -----------------------------
    def _fused_ok(self, trace) -> bool:
        if trace is not None or len(self.encoder) > _lib.MAX_BLOCKS or not self._fusable():
            return False
        if self.training and (self.dropout.p > 0 or self.decoder.drop_p() > 0 or
                              any(b.drop_p() > 0 for b in self.encoder)):
            return False
        return True

    def _forward_fused(self, profile, targets, events=None, train: Optional[dict] = None, after_pack=None) -> List[Tensor]:
        """One host call for the whole forward (carca_forward).  train: None = inference (workspaces cached per shape);
        a dict = the TRAINING forward: fresh buffers, the backward's saved tensors and the dropout sites, all handed back
        through that dict (the keys autograd._CarcaFn keeps in its state)."""
        import ctypes as C

        p_x, p_a, p_c = profile
        emb, dec = self.embeds, self.decoder
        d, H = emb.d, dec.attn.H
        dev = p_x.device
        ops._need_cuda(p_x)
        dpi, _, _ = ops.padded_dims(d, H)
        B, L = p_x.shape
        g = emb.feats_embed.out_features
        table = emb.attr_table()
        n_ctx = p_c.shape[-1]
        n_attrs = table.shape[1] if table is not None else p_a.shape[-1]
        Ns = tuple(int(t[0].shape[1]) for t in targets)
        key = (str(dev), B, L, Ns, n_attrs, n_ctx, table is not None)
        rows = B * (L + sum(Ns))
        f32 = dict(dtype=torch.float32, device=dev)
        if train is None:
            plan = self.__dict__.get("_plan")
            if plan is None or plan["key"] != key:
                plan = dict(key=key, D=_lib.ForwardDesc(), zq=torch.empty(rows, d + g, **f32),
                            es=[torch.empty(B, T, dpi, **f32) for T in (L,) + Ns],
                            xw=[torch.empty(B, L, dpi, **f32) for _ in range(2)])
                self.__dict__["_plan"] = plan
        else:  # everything the backward reads must outlive this call: fresh buffers
            plan = dict(D=_lib.ForwardDesc(), zq=torch.empty(rows, d + g, **f32),
                        es=[torch.empty(B, T, dpi, **f32) for T in (L,) + Ns], xw=[None, None])
        D = plan["D"]
        segs = [(p_x, p_a, p_c)] + [tuple(t) for t in targets]
        keep = []
        pos = emb._pos(L)
        for i, (x, a, c) in enumerate(segs):
            ops._need_cuda(x, a, c)
            T = x.shape[1]
            if x.shape[0] != B or c.shape != (B, T, n_ctx) or (a is not None and a.shape != (B, T, n_attrs)):
                raise CarcaHipError("forward: ids / attrs / ctx shapes do not match")
            if a is None and table is None:
                raise CarcaHipError("forward: attrs is None and no attribute table is registered")
            x32 = ops._ids32(x)
            S = D.segs[i]
            S.ids, S.e_out, S.rows, S.T = x32.data_ptr(), plan["es"][i].data_ptr(), B * T, T
            S.add_pos = int(i == 0 and pos is not None)
            if a is not None:
                a, a_bs = ops._btk_view(a)
                S.attrs, S.attrs_bstride, S.attrs_table = a.data_ptr(), a_bs, None
            else:
                S.attrs, S.attrs_bstride, S.attrs_table = None, 0, table.data_ptr()
                S.attrs_table_rows = table.shape[0]
            if n_ctx > 0:
                c, c_bs = ops._btk_view(c)
                S.ctx, S.ctx_bstride = c.data_ptr(), c_bs
            keep += [x32, a, c]
        D.ngroups, D.B, D.L, D.d, D.g, D.H = len(targets), B, L, d, g, H
        D.n_attrs, D.n_ctx, D.n_blocks, D.ld_e = n_attrs, n_ctx, len(self.encoder), dpi
        prm = [emb.items_embed.weight, emb.feats_embed.weight, emb.feats_embed.bias, emb.joint_embed.weight,
               emb.joint_embed.bias]
        for t in prm:
            ops._need_cuda(t)
        D.items_w, D.feats_w, D.feats_b, D.joint_w, D.joint_b = [t.data_ptr() for t in prm]
        emb.bind_split_weights(n_attrs)
        if pos is not None:
            pos = pos.detach().contiguous()
            keep.append(pos)
        D.pos = pos.data_ptr() if pos is not None else None
        D.zq = plan["zq"].data_ptr()
        if self.__dict__.get("_fold") and not self.training:
            wc, bias_c = emb.folded_weights()
            keep += [wc, bias_c]
            D.fold_wc, D.fold_bias, D.fold_ldwc = wc.data_ptr(), bias_c.data_ptr(), wc.stride(0)
        else:
            D.fold_wc, D.fold_bias, D.fold_ldwc = None, None, 0
        D.z_table, D.ld_z_table = None, 0
        if train is None and not self.training and D.fold_wc is None and USE_Z_TABLE:
            zt = emb.z_table()  # (inference: the item term of the joint embedding per item, cached per weight version)
            keep.append(zt)
            D.z_table, D.ld_z_table = zt.data_ptr(), zt.stride(0)
        if train is None:
            D.x_work[0], D.x_work[1] = plan["xw"][0].data_ptr(), plan["xw"][1].data_ptr()
        repack: list = []  # (after a training step every module repacks: one launch for all of them)
        for i, blk in enumerate(self.encoder):
            blk._check_mode()
            D.sa[i] = blk.weights_struct(dev, repack)
            keep.append(blk.__dict__["_pack_cache"])
            D.sa_residual[i] = int(bool(blk.residual))
        D.ca = dec.weights_struct(dev, self.norm, repack)
        ops.pack_many(repack)
        if after_pack is not None:  # (the training forward: a point BEHIND its first launch, ahead of everything else --
            after_pack()            # where autograd forks the backward's fill + pack onto a second stream)
        D.ca_residual, D.training = int(bool(dec.residual)), int(bool(self.training))
        # the groups' scores are written as column blocks of ONE [B, sum N] tensor: what carca.py:431's torch.cat builds,
        # without the copy (and without the split / re-gather of its gradient in the backward pass)
        ys = JointScores()
        ys.joint = torch.empty(B, sum(Ns), dtype=torch.float32, device=dev)
        off = 0
        for gi, N in enumerate(Ns):
            y = ys.joint[:, off: off + N]
            ys.append(y)
            D.y[gi], D.N[gi] = y.data_ptr(), N
            off += N
        D.ldy = sum(Ns)
        D.p_normed = None
        if train is not None:
            _, _, dpo = ops.padded_dims(d, H)
            u8 = lambda *sh: torch.empty(*sh, dtype=torch.uint8, device=dev)  # noqa: E731
            mk = lambda n, w_: torch.empty(n, w_, **f32)  # noqa: E731
            seed = ops.new_dropout_seed() if self.training else 0
            p_emb = float(self.dropout.p) if self.training else 0.0
            p_blk = max([b.drop_p() for b in self.encoder], default=0.0)
            if any(b.drop_p() != p_blk for b in self.encoder):
                raise CarcaHipError("forward: the blocks' dropout probabilities differ (one value per call)")
            p_ca = dec.drop_p()
            blocks, x_prev = [], plan["es"][0]
            for i, blk in enumerate(self.encoder):
                y_i = torch.empty(B, L, dpi, **f32)
                D.x_out[i] = y_i.data_ptr()
                sv = dict(qn=mk(B * L, dpi), qh=mk(B * L, dpo), kh=mk(B * L, dpo), vh=mk(B * L, dpo), r=mk(B * L, dpi),
                          s2=mk(B * L, dpi), h1=mk(B * L, dpi))
                if p_blk > 0:
                    sv.update(m_attn=u8(B, blk.attn.H, L, L), m_ffn1=u8(B * L, dpi), m_ffn2=u8(B * L, dpi))
                S = D.sa_save[i]
                for k in ("qn", "qh", "kh", "vh", "r", "s2", "h1", "m_attn", "m_ffn1", "m_ffn2"):
                    setattr(S, k, sv[k].data_ptr() if k in sv else None)
                sv["x_in"], sv["p"] = x_prev, p_blk
                blocks.append(sv)
                x_prev = y_i
            p_normed = torch.empty(B, L, dpi, **f32)
            D.p_normed = p_normed.data_ptr()
            csave = dict(kh=mk(B * L, dpo), vh=mk(B * L, dpo), qh=[mk(B * N, dpo) for N in Ns], p=p_ca)
            D.ca_save.kh, D.ca_save.vh = csave["kh"].data_ptr(), csave["vh"].data_ptr()
            if p_ca > 0:
                csave["m_attn"] = [u8(B, H, N, L) for N in Ns]
            for gi in range(_lib.MAX_GROUPS):
                D.ca_save.qh[gi] = csave["qh"][gi].data_ptr() if gi < len(Ns) else None
                D.ca_save.m_attn[gi] = csave["m_attn"][gi].data_ptr() if (p_ca > 0 and gi < len(Ns)) else None
            m_embed = u8(B * L, d) if p_emb > 0 else None
            D.save_blocks = D.save_cross = 1
            D.p_embed, D.p_block, D.p_cross, D.seed = p_emb, p_blk, p_ca, seed
            D.seed_offset = ops.dropout_seed_offset_ptr()
            D.m_embed = m_embed.data_ptr() if m_embed is not None else None
            train.update(es=plan["es"], zq=plan["zq"], blocks=blocks, enc_out=x_prev, p_normed=p_normed, csave=csave,
                         cw=D.ca, m_embed=m_embed, p_emb=p_emb, keep=(keep, D))
        ev = (C.c_void_p * len(events))(*events) if events is not None else None
        D.n_events = len(events) if events is not None else 0
        _lib.check(_lib.load().carca_forward(C.byref(D), ev, ops._stream()), "forward")
        return ys

    def forward_nograd(self, profile, targets, trace: Optional[dict] = None) -> List[Tensor]:
        p_x, p_a, p_c = profile
        if p_x.shape[1] > _lib.MAX_L or len(targets) > _lib.MAX_GROUPS or self.embeds.d > ops.FUSED_MAX_D:
            from . import long_profile

            return long_profile.forward(self, profile, targets, trace)
        if self._fused_ok(trace):
            return self._forward_fused(profile, targets, events=ops.fused_events())
        d = self.embeds.d
        H = self._heads()
        dpi, _, _ = ops.padded_dims(d, H)
        segs = [(p_x, p_a, p_c, False)] + [(o_x, o_a, o_c, True) for (o_x, o_a, o_c) in targets]
        es, _ = self.embeds.embed_segments(segs, ld_e=dpi)
        x = es[0]
        seed = ops.new_dropout_seed() if self.training else 0
        if self.training and self.dropout.p > 0:  # carca.py:416
            ops.dropout_fwd(x, d, float(self.dropout.p), seed, 1000)
        if trace is not None:
            trace["p_embed"] = x[..., :d]
            for gi in range(len(targets)):
                trace[f"o_embed{gi}"] = es[gi + 1][..., :d]
        for i, blk in enumerate(self.encoder):
            blk._check_mode()
            bp = blk.drop_p()
            x = ops.sa_block_fwd(x, p_x, blk.weights_struct(x.device), d, blk.attn.H, blk.residual,
                                 drop=(bp, seed, 4 * i) if bp > 0 else None)
            if trace is not None:
                trace[f"block{i}"] = x[..., :d]
        if not isinstance(self.decoder, CrossAttentionBlock):
            # dot decoders: stand-alone final LayerNorm (carca.py:421), then a row dot per target (carca.py:352-399)
            B, L = p_x.shape
            p_n = ops.layernorm_fwd(x.view(B * L, -1), self.norm.weight, self.norm.bias, d, dpi)
            if trace is not None:
                trace["p_final"] = p_n.view(B, L, dpi)[..., :d]
            ys, _ = self.decoder.score_groups(p_n, [e.view(-1, dpi) for e in es[1:]], [e.shape[1] for e in es[1:]], B, L,
                                              d, dpi)
            return ys
        self.decoder._check_mode()
        H = self.decoder.attn.H
        groups = [(es[gi + 1], targets[gi][0]) for gi in range(len(targets))]
        dp = self.decoder.drop_p()
        if dp > 0:
            ys, p_normed, _ = ops.cross_score_fwd(x, p_x, groups, self.decoder.weights_struct(x.device, self.norm), d, H,
                                                  self.decoder.residual, self.training, save=True,
                                                  drop=(dp, seed, 2000))
        else:
            ys, p_normed = ops.cross_score_fwd(x, p_x, groups, self.decoder.weights_struct(x.device, self.norm), d, H,
                                               self.decoder.residual, self.training, want_normed=trace is not None)
        if trace is not None:
            trace["p_final"] = p_normed[..., :d]
        return ys


# ------------------------------------------------------------------------------------------------
# KNN baseline (knn.py:8-21)
# ------------------------------------------------------------------------------------------------


class KNN(Model):
    """The reference's attribute-similarity baseline: score = attrs(last profile item) . attrs(target), no parameters.
    Extension: register_attr_table(attrs) lets profile/target attribute tensors be None (rows gathered by id on the
    device), as for AllEmbedding."""

    def __init__(self):
        super().__init__()
        self._attr_table: Optional[Tensor] = None

    def register_attr_table(self, attrs: Optional[Tensor]) -> None:
        self._attr_table = None if attrs is None else attrs.detach().to(torch.float32).contiguous()

    def forward(self, profile: Tuple[Tensor, Tensor, Tensor], targets: List[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
        p_x, p_a, p_c = profile
        ys = []
        for o_x, o_a, o_c in targets:
            if p_a is None or o_a is None:
                if self._attr_table is None:
                    raise ValueError("KNN: attribute tensors are None and no table is registered (register_attr_table)")
                ys.append(ops.knn_score(None, None, p_x, o_x, table=self._attr_table))
            else:
                ys.append(ops.knn_score(p_a, o_a))
        return torch.cat(ys, dim=-1)  # knn.py:21


# ------------------------------------------------------------------------------------------------
# loss (carca.py:437-444)
# ------------------------------------------------------------------------------------------------


class BinaryCrossEntropy(nn.Module):
    def forward(self, y_pred: Tensor, y_true: Tensor, mask: Tensor, eps: float = 1e-8,
                denom: Optional[Tensor] = None) -> Tensor:
        """`denom` (extension, device float[1]): the whole batch's mask count when users are sharded (dist.py)."""
        if torch.is_grad_enabled() and y_pred.requires_grad:
            from .autograd import bce_with_grad

            return bce_with_grad(y_pred, y_true, mask, eps, denom)
        loss, _ = ops.bce_fwd(y_pred, y_true, mask != 0, eps, denom=denom)
        return loss
