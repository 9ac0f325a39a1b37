"""One training step / one evaluation batch of the hot path, single- or multi-GPU.

Mirrors the body of the reference's loops (src/train.py:83-97 and :41-51) without its host syncs:
nothing here calls .item(); losses and metric sums stay on the device until the caller reads them.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import dist as cdist
from . import ops
from .modules import BinaryCrossEntropy, get_mask

_loss_fn = BinaryCrossEntropy()


SPARSE_TABLE_BYTES = 64 * 2 ** 20  # item tables above this exchange (row ids, row gradients) instead of the dense grad


def _sparse_tables(model, p_x, o_x):
    emb = getattr(getattr(model, "embeds", None), "items_embed", None)
    if emb is None or emb.weight.numel() * emb.weight.element_size() < SPARSE_TABLE_BYTES:
        return None
    return {emb.weight: torch.cat([p_x.reshape(-1), o_x.reshape(-1)])}


def train_step(model, optim, batch, sharded: bool = False) -> torch.Tensor:
    """batch = (p_x, p_a, p_c, o_x, o_a, o_c, y_true) as the reference's DataLoader yields (train.py:84).

    With sharded=True the batch holds THIS rank's users; the loss is normalised by the global mask
    count and gradients are summed over ranks, which reproduces the single-process step exactly.
    Returns the (device) loss: the global batch loss's local share when sharded.
    """
    p_x, p_a, p_c, o_x, o_a, o_c, y_true = batch
    half = o_x.shape[1] // 2
    pos = tuple(t[:, :half] for t in (o_x, o_a, o_c))  # train.py:86-88
    neg = tuple(t[:, half:] for t in (o_x, o_a, o_c))
    if o_x.shape[1] == 2 * half:  # the kernels want dense [B, L] ids per group: ONE copy for both halves instead of two
        ids2 = o_x.view(o_x.shape[0], 2, half).transpose(0, 1).contiguous()
        pos, neg = (ids2[0],) + pos[1:], (ids2[1],) + neg[1:]
    optim.zero_grad(set_to_none=True)
    y = model(profile=(p_x, p_a, p_c), targets=[pos, neg])
    denom = cdist.global_mask_count(o_x) if sharded else None
    if o_x.dtype == torch.int32:  # BinaryCrossEntropy(y, y_true, get_mask(o_x)) without materialising the float mask
        from .autograd import bce_with_grad

        loss = bce_with_grad(y, y_true, None, 1e-8, denom, ids=o_x)
    else:
        loss = _loss_fn(y, y_true, get_mask(o_x), denom=denom)
    loss.backward()
    if sharded:
        cdist.allreduce_gradients(model.parameters(), sparse_rows=_sparse_tables(model, p_x, o_x))
    optim.step()
    return loss.detach()


@torch.no_grad()
def eval_batch(model, batch, k: int = 10, sums: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scores one (p_x, p_a, p_c, o_x, o_a, o_c, y_true) eval batch (train.py:42-51); accumulates
    [HR@k sum, NDCG@k sum, ties, loss sum, users] into `sums` (device float[5]) with no host sync."""
    p_x, p_a, p_c, o_x, o_a, o_c, y_true = batch
    y = model(profile=(p_x, p_a, p_c), targets=[(o_x, o_a, o_c)])
    y2 = y.reshape(p_x.shape[0], -1)
    if sums is None:
        sums = torch.zeros(5, dtype=torch.float32, device=y.device)
    ops.rank_metrics(y2, k, sums=sums[:3])
    loss = _loss_fn(y2, y_true, get_mask(o_x))
    sums[3] += loss
    sums[4] += p_x.shape[0]
    return y, sums
