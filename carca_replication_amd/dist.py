"""User-sharded data parallelism for N GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-process (scripts/training.py:173, src/train.py:73).  Every op of the hot path
is per-user (batch dimension only, carca.py:411-431), so users shard with NO data-path collective:

  eval   rank r scores users [r*U/R, (r+1)*U/R); one all-reduce of the 4 metric sums at the end
         (mirrors train.py:49-53).
  train  every rank holds a full weight replica; per step
           1. all-reduce(sum) of the loss normaliser sum(mask)  (carca.py:443) -- one float;
           2. local forward/backward with the loss normalised by the GLOBAL count;
           3. ONE all-reduce(sum) of the flat fp32 gradient bucket.
         Summing the ranks' gradients of (local loss sum / global count) is exactly the gradient of
         the single-process batch loss, so R ranks x B/R users reproduce the reference's B-user step.

xGMI on an 8-GPU MI355X node is a point-to-point mesh (7 links x ~153 GB/s per GPU).  The C2 gradient
is 12.4 MB: one bucket, one RCCL call (latency-bound; splitting it only adds launches), issued on the flat
gradient buffer the backward pass already produced (autograd.py: no gather / scatter copies).  Buckets are
capped at `bucket_mb` so that bigger models pipeline their reduce-scatter/all-gather phases over all 7 links.
A C4-sized item table (1 M items x 128 = 512 MB dense) is not all-reduced at all: a step touches at most
B*3L rows per rank, so ranks all-gather (row ids, row gradients) -- ~10 MB per rank -- and add them locally
(`sparse_rows`), which is exactly the dense sum.

Works on CPU tensors with the gloo backend too (that is how tests/test_dist_cpu.py covers it here).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the launcher's env (RANK/WORLD_SIZE/MASTER_*). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, disjoint, exhaustive split of n users: rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_mask_count(ids: torch.Tensor) -> torch.Tensor:
    """sum over ALL ranks of the number of non-pad target slots: the loss normaliser of carca.py:443."""
    cnt = torch.count_nonzero(ids).to(torch.float32).reshape(1)
    if world_size() > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return cnt


def _buckets(grads: List[torch.Tensor], cap_bytes: int) -> List[List[torch.Tensor]]:
    out, cur, size = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and size + nbytes > cap_bytes:
            out.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nbytes
    if cur:
        out.append(cur)
    return out


def _shared_flat(grads: List[torch.Tensor]) -> Optional[torch.Tensor]:
    """The one flat buffer all gradients are views of, in order, each starting on a 4-element boundary
    (autograd._zeros_like_params hands the backward's results out that way; autograd keeps the storage), or None."""
    if not grads:
        return None
    st = grads[0].untyped_storage()
    off = 0
    for g in grads:
        if (g.untyped_storage().data_ptr() != st.data_ptr() or not g.is_contiguous() or g.storage_offset() != off or
                g.dtype != grads[0].dtype):
            return None
        off += (g.numel() + 3) // 4 * 4
    if off * grads[0].element_size() > st.nbytes():
        return None
    return grads[0].new_empty(0).set_(st, 0, (off,), (1,))


def allgather_row_gradients(grad: torch.Tensor, ids: torch.Tensor) -> None:
    """Sum over ranks of a gradient that is non-zero only in the rows `ids` touches (an embedding table):
    every rank contributes (ids, grad[ids]) with duplicates zeroed, all-gathers them and adds the lot into its own
    table.  Same result as a dense all-reduce(sum); traffic ~ world x len(ids) x (d + 1) floats instead of the table.
    `ids` must have the same length on every rank (it has: B x 3L slots of the batch, pads included)."""
    world = world_size()
    if world == 1:
        return
    flat_ids = ids.reshape(-1).to(torch.int64)
    srt, _ = torch.sort(flat_ids)
    first = torch.ones_like(srt, dtype=torch.bool)
    first[1:] = srt[1:] != srt[:-1]
    rows = grad.index_select(0, srt) * first.unsqueeze(1).to(grad.dtype)  # each touched row once, duplicates as zeros
    all_ids = [torch.empty_like(srt) for _ in range(world)]
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    h1 = dist.all_gather(all_ids, srt, async_op=True)
    h2 = dist.all_gather(all_rows, rows, async_op=True)
    h1.wait()
    h2.wait()
    grad.index_fill_(0, srt, 0.0)  # own contribution comes back with everybody else's
    grad.index_add_(0, torch.cat(all_ids), torch.cat(all_rows))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, average: bool = False,
                        sparse_rows: Optional[dict] = None) -> None:
    """Sum (or average) .grad over ranks: flat fp32 buckets, asynchronously issued, then waited.
    sparse_rows: {parameter: ids} for embedding tables whose gradient is exchanged row-wise instead
    (allgather_row_gradients)."""
    world = world_size()
    if world == 1:
        return
    params = [p for p in params if p.grad is not None]
    sparse_rows = sparse_rows or {}
    sparse = [p for p in params if any(p is q for q in sparse_rows)]
    dense = [p.grad for p in params if not any(p is q for q in sparse_rows)]
    cap = int(bucket_mb * 2 ** 20)
    work = []
    flat_all = _shared_flat([p.grad for p in params]) if not sparse else None
    if flat_all is not None:  # the backward's own flat buffer: reduce it in place, in chunks of the bucket size
        step = max(1, cap // flat_all.element_size())
        for off in range(0, flat_all.numel(), step):
            chunk = flat_all[off: off + step]
            work.append((dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True), chunk, None))
    else:
        for bucket in _buckets(dense, cap):
            flat = torch.cat([g.reshape(-1) for g in bucket])
            work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
    for p in sparse:
        ids = next(v for q, v in sparse_rows.items() if q is p)
        allgather_row_gradients(p.grad, ids)
        if average:
            p.grad.div_(world)
    for h, flat, bucket in work:
        h.wait()
        if average:
            flat.div_(world)
        if bucket is None:
            continue
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off: off + n].view_as(g))
            off += n


def allreduce_sums(sums: torch.Tensor) -> torch.Tensor:
    """Metric sums [HR, NDCG, loss, users, ...] -> summed over ranks (train.py:49-53 across shards)."""
    if world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums
