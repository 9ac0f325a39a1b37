"""CPU oracle for the CARCA forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``carca_replication_amd``) never imports this module and fails
loudly when its HIP library is missing.

What it is: a functional fp32 (optionally fp64) restatement, on the CPU with
plain torch tensor ops, of the algorithm in the reference's ``src/carca.py`` /
``src/utils.py`` / ``src/train.py``.  Each function cites the reference
file:line it follows (paths relative to the reference checkout).  Parameters
are passed as a flat ``dict`` keyed exactly like the reference's ``state_dict``
(SURVEY.md section 8b), so a fixture's weights can be fed straight in.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself,
captured in the build container by ``tests/golden/make_golden.py`` and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays
every fixture through this file.

Gradients: every function is differentiable torch code, so a gradient oracle
is ``torch.autograd`` over this restatement (checked against the reference's
own ``loss.backward()`` in fixture G2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

Params = Dict[str, Tensor]

# fp32(-(2**32) + 1.0) rounds to exactly -2**32 (carca.py:251)
NEG_FILL = -(2.0 ** 32) + 1.0
LN_EPS = 1e-5  # torch.nn.LayerNorm default, carca.py:279,283,408
LRELU_SLOPE = 0.01  # torch.nn.LeakyReLU default, carca.py:285


@dataclass
class CarcaConfig:
    d: int
    H: int
    n_blocks: int
    residual_sa: bool = True
    residual_ca: bool = True
    encoding: str = "identity"  # identity | learnable | positional
    # ablation variants (training.py:76-100)
    embedding: str = "all"      # all | attrctx | attr | id | mlpid        carca.py:63-198
    decoder: str = "ca"         # ca | dot | wdot                          carca.py:321-399
    gamma: float = 0.9          # WeightedDotProduct decay                 training.py:55
    l2_norm: bool = False       # WeightedDotProduct normalisation         training.py:56


# --------------------------------------------------------------------------- #
# a1: mask                                                       utils.py:6-7  #
# --------------------------------------------------------------------------- #
def get_mask(ids: Tensor, dtype: torch.dtype = torch.float32) -> Tensor:
    """1.0 where id != 0, else 0.0 (utils.py:7)."""
    return (ids != 0).to(dtype)


# --------------------------------------------------------------------------- #
# a9: encodings                                                carca.py:15-60  #
# --------------------------------------------------------------------------- #
def sinusoid_table(d_model: int, max_len: int, dtype=torch.float32) -> Tensor:
    """Buffer ``pe`` [1, max_len, d_model] of PositionalEncoding (carca.py:47-52)."""
    pos = torch.arange(max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(pos * div)
    pe[0, :, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def position_term(params: Params, cfg: CarcaConfig, T: int) -> Optional[Tensor]:
    """Additive [T, d] table for the profile side, or None for identity.

    learnable: rows 0..T-1 of ``embeds.enc.encoding.weight`` (carca.py:26-30);
    positional: ``embeds.enc.pe[0, :T]`` (carca.py:59).
    """
    if cfg.encoding == "identity":
        return None
    if cfg.encoding == "learnable":
        return params["embeds.enc.encoding.weight"][:T]
    if cfg.encoding == "positional":
        return params["embeds.enc.pe"][0, :T]
    raise ValueError(f"Unknown encoding type: {cfg.encoding}")


# --------------------------------------------------------------------------- #
# a2: AllEmbedding.forward                                    carca.py:85-95  #
# --------------------------------------------------------------------------- #
def all_embedding(params: Params, cfg: CarcaConfig, x: Tensor, a: Tensor, c: Tensor, mask: Tensor,
                  target: bool, return_q: bool = False):
    """e = (W_j [E[x]*sqrt(d) ; W_f [a;c] + b_f] + b_j (+pos)) * mask."""
    Wf, bf = params["embeds.feats_embed.weight"], params["embeds.feats_embed.bias"]
    Wj, bj = params["embeds.joint_embed.weight"], params["embeds.joint_embed.bias"]
    E = params["embeds.items_embed.weight"]
    n_attrs = a.shape[-1]
    # feats: attrs first, then ctx (carca.py:86) -- done as two partial products
    q = a @ Wf[:, :n_attrs].T + c @ Wf[:, n_attrs:].T + bf
    z = E[x.long()] * (cfg.d ** 0.5)  # carca.py:87-88
    d = cfg.d
    e = z @ Wj[:, :d].T + q @ Wj[:, d:].T + bj  # joint: z first, then q (carca.py:89)
    if not target:
        pos = position_term(params, cfg, x.shape[1])
        if pos is not None:
            e = e + pos.unsqueeze(0)  # carca.py:91-92
    e = e * mask.unsqueeze(-1)  # carca.py:94
    if return_q:
        return e, q, z
    return e


# --------------------------------------------------------------------------- #
# f4: ablation embeddings                                    carca.py:98-198  #
# --------------------------------------------------------------------------- #
def embedding(params: Params, cfg: CarcaConfig, x: Tensor, a: Tensor, c: Tensor, mask: Tensor, target: bool) -> Tensor:
    """Dispatch on cfg.embedding; every variant ends with (+pos on the profile side) * mask like AllEmbedding."""
    kind = cfg.embedding
    if kind == "all":
        return all_embedding(params, cfg, x, a, c, mask, target)
    if kind in ("attrctx", "attr"):
        Wf, bf = params["embeds.feats_embed.weight"], params["embeds.feats_embed.bias"]
        Wj, bj = params["embeds.joint_embed.weight"], params["embeds.joint_embed.bias"]
        n_attrs = a.shape[-1]
        q = a @ Wf[:, :n_attrs].T + bf                      # AttrEmbedding: attributes only (carca.py:142)
        if kind == "attrctx":
            q = q + c @ Wf[:, n_attrs:].T                   # AttrCtxEmbedding: cat((a, c)) (carca.py:113)
        e = q @ Wj.T + bj                                   # joint_embed: g -> d, no item-id term (carca.py:114,143)
    elif kind == "id":
        e = params["embeds.items_embed.weight"][x.long()] * (cfg.d ** 0.5)          # carca.py:164-165
    elif kind == "mlpid":
        z = params["embeds.items_embed.weight"][x.long()] * (cfg.d ** 0.5)          # [.., g] (carca.py:190-191)
        e = z @ params["embeds.feats_embed.weight"].T + params["embeds.feats_embed.bias"]  # carca.py:192
    else:
        raise ValueError(f"Unknown embedding type: {kind}")
    if not target:
        pos = position_term(params, cfg, x.shape[1])
        if pos is not None:
            e = e + pos.unsqueeze(0)
    return e * mask.unsqueeze(-1)


# --------------------------------------------------------------------------- #
# f4: ablation decoders                                     carca.py:352-399  #
# --------------------------------------------------------------------------- #
def dot_decoder(cfg: CarcaConfig, o: Tensor, p: Tensor, training: bool) -> Tensor:
    """DotProduct (carca.py:361-367) and WeightedDotProduct (carca.py:383-399).  Neither looks at the masks."""
    L = p.shape[1]
    if cfg.decoder == "wdot":
        # carca.py:376-378,385-386: W[t][j] = gamma^j for j <= t; pw[b,t,j,:] = p[b,t,:] (the repeat runs along the
        # NEW axis), so the sum over j scales slot t by c_t = sum_{j<=t} gamma^j -- it does not mix slots
        # (kept as the same product-then-sum so that the fp32 rounding matches the fixtures to the last bit or two)
        w = cfg.gamma ** torch.arange(0, L)                                  # float32, like the reference's buffer
        W = torch.tril(w.unsqueeze(0).expand(L, L)).to(p.dtype)              # [t][j]
        p = (p.unsqueeze(2) * W.view(1, L, L, 1)).sum(dim=2)                 # = p[b,t,:] * c_t
        if cfg.l2_norm:
            p = torch.nn.functional.normalize(p, dim=2)
            o = torch.nn.functional.normalize(o, dim=2)
    if training:
        s = (p * o).sum(-1)               # slot t of the profile against target t (carca.py:362,391)
    else:
        s = (p[:, -1:, :] * o).sum(-1)    # last profile slot against every candidate (carca.py:364,393)
    if cfg.decoder == "wdot" and cfg.l2_norm:
        return (s + 1.0) / 2.0            # carca.py:395-396
    return torch.sigmoid(s)


# --------------------------------------------------------------------------- #
# f4: KNN baseline model                                          knn.py:8-21 #
# --------------------------------------------------------------------------- #
def knn_forward(profile: Tuple[Tensor, Tensor, Tensor], targets: List[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
    """score = attribute row of the LAST profile slot (knn.py:14) . attribute row of each target (knn.py:18)."""
    last = profile[1][:, -1:, :]
    return torch.cat([(last * o_a).sum(dim=-1) for _, o_a, _ in targets], dim=-1)


# --------------------------------------------------------------------------- #
# a3: MultiHeadAttention.forward                            carca.py:228-265  #
# --------------------------------------------------------------------------- #
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def attention_mask(q_mask: Tensor, k_mask: Tensor, causal: Optional[int]) -> Tensor:
    """bool [B, Tq, Tk]: q valid & k valid (& j <= i + causal) (carca.py:246-250)."""
    m = (q_mask.unsqueeze(2) * k_mask.unsqueeze(1)) != 0
    if causal is not None:
        Tq, Tk = m.shape[1], m.shape[2]
        i = torch.arange(Tq).unsqueeze(1)
        j = torch.arange(Tk).unsqueeze(0)
        m = m & (j <= i + causal).unsqueeze(0)
    return m


def mha(params: Params, prefix: str, H: int, query: Tensor, key: Tensor, value: Tensor, q_mask: Tensor,
        k_mask: Tensor, causal: Optional[int], drop_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Returns (weights [B,H,Tq,Tk] pre-dropout, out [B,Tq,d]).

    Head h of user b is the reference's batch index h*B+b (carca.py:242-244);
    here it is kept as [B,H,...], which is the same numbers in another order.
    ``drop_mask`` (already scaled by 1/(1-p)) multiplies the weights where the
    reference applies dropout (carca.py:258).
    """
    B, Tq, d = query.shape
    Tk = key.shape[1]
    dh = d // H
    assert d % H == 0, "Embedding dim must be divisible by number of heads"  # carca.py:208
    Q = query @ params[prefix + "WQ.weight"].T + params[prefix + "WQ.bias"]
    K = key @ params[prefix + "WK.weight"].T + params[prefix + "WK.bias"]
    V = value @ params[prefix + "WV.weight"].T + params[prefix + "WV.bias"]
    Q = Q.view(B, Tq, H, dh).transpose(1, 2)
    K = K.view(B, Tk, H, dh).transpose(1, 2)
    V = V.view(B, Tk, H, dh).transpose(1, 2)
    m = attention_mask(q_mask, k_mask, causal).unsqueeze(1)  # [B,1,Tq,Tk]
    add = torch.where(m, torch.zeros((), dtype=query.dtype), torch.full((), NEG_FILL, dtype=query.dtype))
    w = (add + Q @ K.transpose(-1, -2)) / (dh ** 0.5)  # mask added BEFORE scaling (carca.py:253-254)
    w = torch.softmax(w, dim=-1)
    w = w * m  # fully masked rows -> exact zeros (carca.py:256)
    wd = w if drop_mask is None else w * drop_mask
    out = (wd @ V).transpose(1, 2).reshape(B, Tq, d)
    return w, out


# --------------------------------------------------------------------------- #
# a4: SelfAttentionBlock.forward                            carca.py:297-318  #
# --------------------------------------------------------------------------- #
def sa_block(params: Params, cfg: CarcaConfig, i: int, x: Tensor, mask: Tensor, masks: Optional[dict] = None) -> Tensor:
    """`masks` (optional) injects dropout multipliers (keep / (1-p)) at the block's three nn.Dropout sites:
    attn{i} [B,H,L,L] (carca.py:258), ffn1_{i} and ffn2_{i} [B,L,d] (carca.py:309,312)."""
    pre = f"encoder.{i}."
    mk = masks or {}
    q = layer_norm(x, params[pre + "norm1.weight"], params[pre + "norm1.bias"])
    _, s = mha(params, pre + "attn.", cfg.H, q, x, x, mask, mask, causal=0,
               drop_mask=mk.get(f"attn{i}"))  # K,V from un-normed x
    if cfg.residual_sa:
        s = s + q  # the NORMED tensor is the residual (carca.py:301-302)
    s = layer_norm(s, params[pre + "norm2.weight"], params[pre + "norm2.bias"])
    W1, b1 = params[pre + "ffn_1.weight"][:, :, 0], params[pre + "ffn_1.bias"]  # Conv1d k=1 == Linear
    W2, b2 = params[pre + "ffn_2.weight"][:, :, 0], params[pre + "ffn_2.bias"]
    f = torch.nn.functional.leaky_relu(s @ W1.T + b1, LRELU_SLOPE)
    if f"ffn1_{i}" in mk:
        f = f * mk[f"ffn1_{i}"]
    f = f @ W2.T + b2
    if f"ffn2_{i}" in mk:
        f = f * mk[f"ffn2_{i}"]
    if cfg.residual_sa:
        f = f + s
    return f  # no re-masking (carca.py:318)


# --------------------------------------------------------------------------- #
# a6: CrossAttentionBlock.forward                           carca.py:338-349  #
# --------------------------------------------------------------------------- #
def cross_block(params: Params, cfg: CarcaConfig, o: Tensor, o_mask: Tensor, p: Tensor, p_mask: Tensor,
                training: bool, return_w: bool = False, drop_mask: Optional[Tensor] = None):
    causal = -1 if training else None  # carca.py:339
    w, s = mha(params, "decoder.attn.", cfg.H, o, p, p, o_mask, p_mask, causal, drop_mask=drop_mask)
    if cfg.residual_ca:
        s = s + o
    logit = s @ params["decoder.ffn.weight"].T + params["decoder.ffn.bias"]
    y = torch.sigmoid(logit.squeeze())  # bare squeeze: B=1 -> [N], N=1 -> [B] (carca.py:346)
    if return_w:
        return y, w
    return y


# --------------------------------------------------------------------------- #
# a5/a7: CARCA.forward                                      carca.py:411-431  #
# --------------------------------------------------------------------------- #
def carca_forward(params: Params, cfg: CarcaConfig, profile: Tuple[Tensor, Tensor, Tensor],
                  targets: Sequence[Tuple[Tensor, Tensor, Tensor]], training: bool,
                  trace: Optional[dict] = None, masks: Optional[dict] = None) -> Tensor:
    """Forward in eval mode, or train mode with the dropout multipliers given in `masks`
    (keys: embed [B,L,d] (carca.py:416), attn{i}/ffn1_{i}/ffn2_{i} per block, cross{g} [B,H,N,L]); no masks = p 0."""
    p_x, p_a, p_c = profile
    dt = p_a.dtype
    p_mask = get_mask(p_x, dt)
    mk = masks or {}
    p_e = embedding(params, cfg, p_x, p_a, p_c, p_mask, target=False)
    if trace is not None:
        trace["p_mask"], trace["p_embed"] = p_mask, p_e
    if "embed" in mk:
        p_e = p_e * mk["embed"]
    for i in range(cfg.n_blocks):
        p_e = sa_block(params, cfg, i, p_e, p_mask, masks)
        if trace is not None:
            trace[f"block{i}"] = p_e
    p_e = layer_norm(p_e, params["norm.weight"], params["norm.bias"])
    if trace is not None:
        trace["p_final"] = p_e
    ys = []
    for gi, (o_x, o_a, o_c) in enumerate(targets):
        o_mask = get_mask(o_x, dt)
        o_e = embedding(params, cfg, o_x, o_a, o_c, o_mask, target=True)
        if cfg.decoder == "ca":
            y, w = cross_block(params, cfg, o_e, o_mask, p_e, p_mask, training, return_w=True,
                               drop_mask=mk.get(f"cross{gi}"))
        else:
            y, w = dot_decoder(cfg, o_e, p_e, training), None
        if trace is not None:
            trace[f"o_embed{gi}"], trace[f"dec_w{gi}"] = o_e, w
        ys.append(y)
    return torch.cat(ys, dim=-1)


# --------------------------------------------------------------------------- #
# a8: BinaryCrossEntropy.forward                            carca.py:441-444  #
# --------------------------------------------------------------------------- #
def bce_loss(y_pred: Tensor, y_true: Tensor, mask: Tensor, eps: float = 1e-8) -> Tensor:
    loss = -(y_true * torch.log(y_pred + eps) + (1.0 - y_true) * torch.log(1.0 - y_pred + eps))
    return torch.sum(loss * mask) / torch.sum(mask)


# --------------------------------------------------------------------------- #
# M: metrics                                                   train.py:15-32  #
# --------------------------------------------------------------------------- #
def positive_rank(y_pred: Tensor) -> Tensor:
    """0-based rank of column 0 under a descending sort, for tie-free scores."""
    return (y_pred[:, 1:] > y_pred[:, :1]).sum(dim=1)


def hr_ndcg_sums(y_pred: Tensor, k: int) -> Tuple[float, float]:
    """Sum over users of HR@k and NDCG@k with the positive in column 0.

    Equals compute_HR / compute_NDCG (train.py:15-32) whenever the scores are
    tie-free (torch.sort is unstable, so ties are undefined in the reference).
    """
    r = positive_rank(y_pred)
    hit = r < k
    hr = float(hit.sum())
    ndcg = float((1.0 / torch.log2(r[hit].to(torch.float32) + 2.0)).sum())
    return hr, ndcg


def hr_ndcg_sort(y_pred: Tensor, y_true: Tensor, k: int) -> Tuple[float, float]:
    """Sort-based restatement with the reference's exact steps (train.py:15-32)."""
    _, idx = torch.sort(y_pred, descending=True)
    top = torch.gather(y_true, 1, idx)[:, :k]
    hr = float(top.sum())
    ranks = torch.nonzero(top)[:, 1]
    ndcg = float((1.0 / torch.log2(ranks + 2)).sum())
    return hr, ndcg


# --------------------------------------------------------------------------- #
# synthetic inputs                                      #
# --------------------------------------------------------------------------- #
def init_params(cfg: CarcaConfig, n_items: int, g: int, n_ctx: int, n_attrs: int, L: int, seed: int = 0,
                dtype=torch.float32) -> Params:
    """Random parameters with the reference's shapes and init families.

    xavier-uniform matrices / zero biases / unit LayerNorm (carca.py:77-83,
    220-226, 291-295, 335-336); Linear/Conv biases that the reference leaves at
    torch's default init are drawn U(-1/sqrt(fan_in), 1/sqrt(fan_in)) here only
    where the reference does so too (none on this path: all are zeroed).
    The values are NOT the reference's RNG stream; fixtures carry real weights.
    """
    gen = torch.Generator().manual_seed(seed)
    d, F = cfg.d, n_attrs + n_ctx

    def xavier(*shape):
        fan_out, fan_in = shape[0], shape[1]
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        return ((torch.rand(*shape, generator=gen) * 2 - 1) * bound).to(dtype)

    P: Params = {}
    kind = cfg.embedding
    if kind in ("all", "id", "mlpid"):
        E = xavier(n_items, g if kind == "mlpid" else d)
        E[0] = 0
        P["embeds.items_embed.weight"] = E
    if kind in ("all", "attrctx", "attr"):
        P["embeds.feats_embed.weight"] = xavier(g, n_attrs if kind == "attr" else F)
        P["embeds.feats_embed.bias"] = torch.zeros(g, dtype=dtype)
        P["embeds.joint_embed.weight"] = xavier(d, d + g if kind == "all" else g)
        P["embeds.joint_embed.bias"] = torch.zeros(d, dtype=dtype)
    if kind == "mlpid":
        P["embeds.feats_embed.weight"] = xavier(d, g)
        P["embeds.feats_embed.bias"] = torch.zeros(d, dtype=dtype)
    if cfg.encoding == "learnable":
        P["embeds.enc.encoding.weight"] = xavier(L, d)
    elif cfg.encoding == "positional":
        P["embeds.enc.pe"] = sinusoid_table(d, L, dtype)

    def attn(prefix):
        for n in ("WQ", "WK", "WV"):
            P[prefix + n + ".weight"] = xavier(d, d)
            P[prefix + n + ".bias"] = torch.zeros(d, dtype=dtype)

    for i in range(cfg.n_blocks):
        pre = f"encoder.{i}."
        for n in ("norm1", "norm2"):
            P[pre + n + ".weight"] = torch.ones(d, dtype=dtype)
            P[pre + n + ".bias"] = torch.zeros(d, dtype=dtype)
        attn(pre + "attn.")
        for n in ("ffn_1", "ffn_2"):
            P[pre + n + ".weight"] = xavier(d, d).unsqueeze(-1)
            P[pre + n + ".bias"] = torch.zeros(d, dtype=dtype)
    P["norm.weight"] = torch.ones(d, dtype=dtype)
    P["norm.bias"] = torch.zeros(d, dtype=dtype)
    if cfg.decoder == "ca":
        attn("decoder.attn.")
        P["decoder.ffn.weight"] = xavier(1, d)
        P["decoder.ffn.bias"] = torch.zeros(1, dtype=dtype)
    return P


def perturb_params(P: Params, seed: int = 1, scale: float = 0.05) -> Params:
    """Make every bias / LayerNorm parameter non-trivial so tests exercise them."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in P.items():
        if k.endswith(".bias") or ("norm" in k and k.endswith(".weight")):
            out[k] = v + scale * torch.randn(v.shape, generator=gen).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def synth_eval_batch(B: int, L: int, N: int, n_items: int, n_attrs: int, n_ctx: int, seed: int = 1234,
                     min_len: int = 3, attrs_table=None):
    """Synthetic Beauty-shaped eval batch, SURVEY.md section 8d.

    lengths ~ U{min_len..L} left-padded with 0 (data.py:113,173); ids ~ U{1..n_items-1};
    candidate 0 is the positive, 1..N-1 distinct negatives not in the profile
    (data.py:77-87); p_a = attrs[p_x], o_a = attrs[o_x]; every candidate carries the
    positive's ctx (data.py:185).
    """
    import numpy as np

    if n_items - 1 < L + N:
        raise ValueError(f"synth_eval_batch: {n_items - 1} item ids cannot give {N - 1} distinct negatives outside a "
                         f"profile of up to {L} items")
    rng = np.random.default_rng(seed)
    if attrs_table is None:
        attrs_table = rng.random((n_items, n_attrs), dtype=np.float32)
        attrs_table[0] = 0.0  # pad row (data.py:33-34)
    p_x = np.zeros((B, L), dtype=np.int32)
    o_x = np.zeros((B, N), dtype=np.int32)
    for u in range(B):
        ell = int(rng.integers(min(min_len, L), L + 1))
        p_x[u, L - ell:] = rng.integers(1, n_items, size=ell)
        o_x[u, 0] = rng.integers(1, n_items)
        seen = set(p_x[u].tolist()) | {int(o_x[u, 0])}
        negs: List[int] = []
        while len(negs) < N - 1:
            cand = int(rng.integers(1, n_items))
            if cand not in seen:
                seen.add(cand)
                negs.append(cand)
        o_x[u, 1:] = negs
    p_c = rng.random((B, L, n_ctx), dtype=np.float32) * (p_x != 0)[..., None]
    o_c = np.repeat(rng.random((B, 1, n_ctx), dtype=np.float32), N, axis=1)
    p_a = attrs_table[p_x]
    o_a = attrs_table[o_x]
    t = torch.from_numpy
    return (t(p_x), t(p_a), t(p_c)), (t(o_x), t(o_a), t(o_c)), t(attrs_table)
