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
is 12.4 MB: one bucket, one RCCL call (latency-bound; splitting it only adds launches).  Buckets are
capped at `bucket_mb` so that a C4-sized model (item table 512 MB) pipelines reduce-scatter/all-gather
phases over all 7 links instead of serialising one giant ring pass.

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


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, average: bool = False) -> None:
    """Sum (or average) .grad over ranks in flat fp32 buckets, asynchronously issued, then waited."""
    world = world_size()
    if world == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    work = []
    for bucket in _buckets(grads, int(bucket_mb * 2 ** 20)):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        work.append((h, flat, bucket))
    for h, flat, bucket in work:
        h.wait()
        if average:
            flat.div_(world)
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
