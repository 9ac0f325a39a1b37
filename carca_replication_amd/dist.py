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
is 12.4 MB, reduced IN PLACE in the flat buffer the backward pass produced it in (autograd._grad_buffers: no gather /
scatter copies) as two RCCL calls: everything but feats_embed.{weight, bias} (5 MB) as soon as the backward's last
launch -- the 0.6 ms weight-gradient kernel of feats_embed -- has been issued, so that it runs UNDER that kernel
(engine.train_step), then feats_embed's 7.4 MB.  Ranges are cut at `bucket_mb` so that bigger models pipeline their
reduce-scatter/all-gather phases over all 7 links.
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
    if _active():
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return cnt


_VALIDATED: set = set()


def validate_global_batch(n_local: int, global_batch: Optional[int], device=None) -> None:
    """`global_batch` (users of the whole step over all ranks, shards as shard_range cuts them) checked ON EVERY RANK: the
    ranks gather their user counts -- one tiny collective, once per (count, global_batch) pair -- and every one of them raises
    the same ValueError when a shard is larger than ceil(global_batch / world) or the counts do not add up.  Checked by the
    violating rank alone (what engine._row_exchange_len does on its own) the others would walk into the step's next
    collective and wait there until the backend's timeout."""
    if global_batch is None or not _active():
        return
    key = (int(n_local), int(global_batch), dist.get_world_size())
    if key in _VALIDATED:
        return
    world = dist.get_world_size()
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device if device is not None else "cpu")
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    counts = [int(t.item()) for t in every]
    per_rank = -(-int(global_batch) // world)
    if max(counts) > per_rank or sum(counts) != int(global_batch):
        raise ValueError(f"global_batch={global_batch} over {world} ranks allows at most {per_rank} users per rank and needs "
                         f"them to add up; the ranks hold {counts} (dist.shard_range)")
    _VALIDATED.add(key)


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


# Test hook: run the collectives of a sharded step even in a 1-rank group (a 1-GPU box can then execute the HIP flat
# gradient buffer -> in-place all-reduce -> optimizer chain end to end); `last_reduce` says which path a step took.
FORCE_COLLECTIVES = False
last_reduce: dict = {}


def _active() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def flat_layout(model, params: List[torch.nn.Parameter]) -> Optional[dict]:
    """model._flat_grad (autograd._grad_buffers: ONE flat buffer, [early | late | staging | big tables]) when every
    .grad of `params` still is a view of it, else None."""
    info = getattr(model, "__dict__", {}).get("_flat_grad")
    if not info or info["n_params"] != len(params):
        return None
    st = info["flat"].untyped_storage().data_ptr()
    if any(p.grad is None or p.grad.untyped_storage().data_ptr() != st for p in params):
        return None
    return info


def allreduce_range(flat: torch.Tensor, lo: int, hi: int, bucket_mb: float = 64.0) -> list:
    """Asynchronous in-place all-reduce(sum) of flat[lo:hi], in chunks of at most bucket_mb; returns the work handles."""
    step = max(1, int(bucket_mb * 2 ** 20) // flat.element_size())
    return [dist.all_reduce(flat[off: min(off + step, hi)], op=dist.ReduceOp.SUM, async_op=True)
            for off in range(lo, hi, step)]


def allgather_row_gradients(grad: torch.Tensor, ids: torch.Tensor, pad_to: Optional[int] = None) -> Optional[torch.Tensor]:
    """Sum over ranks of a gradient that is non-zero only in the rows `ids` touches (an embedding table):
    every rank contributes (ids, grad[ids]) with duplicates zeroed, all-gathers them and adds the lot into its own
    table.  Same result as a dense all-reduce(sum); traffic ~ world x len(ids) x (d + 1) floats instead of the table.
    Ranks may hold different numbers of ids (B % world != 0, a short last batch): the lists are padded with id 0 --
    the pad row, whose gradient is zero by construction (nn.Embedding(padding_idx=0), carca.py:73).
    pad_to: the list length every rank pads to, known on the host without asking the other ranks (engine.train_step
      derives it from the GLOBAL batch size: ceil(B / world) users x ids per user): no host sync on the step.  None =
      the ranks agree on the longest list with an all-reduce(max) whose result the host must read (one sync per step).
    Returns every rank's ids, concatenated (the rows the exchanged gradients land in: what a touched-row optimizer has
    to mark, engine._mark_touched_rows), or None outside a process group."""
    if not _active():
        return None
    world = dist.get_world_size()
    flat_ids = ids.reshape(-1).to(torch.int64)
    if pad_to is None:
        n_max = torch.tensor([flat_ids.numel()], dtype=torch.int64, device=flat_ids.device)
        dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
        pad_to = int(n_max.item())
    if flat_ids.numel() > pad_to:
        raise ValueError(f"allgather_row_gradients: {flat_ids.numel()} ids on this rank, but the ranks agreed on {pad_to}")
    pad = int(pad_to) - flat_ids.numel()
    if pad > 0:
        flat_ids = torch.cat([flat_ids, flat_ids.new_zeros(pad)])
    srt, _ = torch.sort(flat_ids)
    first = torch.ones_like(srt, dtype=torch.bool)
    first[1:] = srt[1:] != srt[:-1]
    rows = grad.index_select(0, srt) * first.unsqueeze(1).to(grad.dtype)  # each touched row once, duplicates as zeros
    all_ids = torch.empty(world * srt.numel(), dtype=srt.dtype, device=srt.device)
    all_rows = torch.empty(world * rows.shape[0], rows.shape[1], dtype=rows.dtype, device=rows.device)
    h1 = dist.all_gather_into_tensor(all_ids, srt, async_op=True)
    h2 = dist.all_gather_into_tensor(all_rows, rows, async_op=True)
    h1.wait()
    h2.wait()
    grad.index_fill_(0, srt, 0.0)  # own contribution comes back with everybody else's
    grad.index_add_(0, all_ids, all_rows)
    return all_ids


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, average: bool = False,
                        sparse_rows: Optional[dict] = None, flat_info: Optional[dict] = None,
                        early_work: Optional[list] = None, sparse_pad_to: Optional[int] = None) -> dict:
    """Sum (or average) .grad over ranks.
    flat_info (flat_layout): the gradients are views of the backward's own flat buffer -- its [early | late] ranges are
      reduced IN PLACE (no gather / scatter copies), asynchronously, in chunks of bucket_mb; early_work: handles of an
      early-range reduction the caller already started (engine.train_step starts it under the backward's last kernel).
    Otherwise: flat fp32 buckets of copies.
    sparse_rows: {parameter: ids} for embedding tables whose gradient is exchanged row-wise instead
    (allgather_row_gradients, padded to sparse_pad_to ids per rank); with flat_info these are its `big` tables.
    Returns {id(parameter): every rank's ids} for the row-exchanged tables."""
    global last_reduce
    gathered: dict = {}
    if not _active():
        return gathered
    world = dist.get_world_size()
    params = [p for p in params if p.grad is not None]
    sparse_rows = sparse_rows or {}
    sparse = [p for p in params if any(p is q for q in sparse_rows)]
    cap = int(bucket_mb * 2 ** 20)
    work, flat_handles, flat_range = [], [], None
    if flat_info is not None and all(any(p is q for q in sparse_rows) for p in flat_info["big"]):
        flat = flat_info["flat"]
        handles = list(early_work) if early_work is not None else allreduce_range(flat, *flat_info["early"], bucket_mb)
        handles += allreduce_range(flat, *flat_info["late"], bucket_mb)
        flat_handles, flat_range = handles, flat[flat_info["early"][0]: flat_info["late"][1]]
        last_reduce = dict(path="flat-inplace", floats=flat_info["late"][1], early_overlapped=early_work is not None,
                           launches=len(handles), sparse_tables=len(sparse))
    else:
        dense = [p.grad for p in params if not any(p is q for q in sparse_rows)]
        for bucket in _buckets(dense, cap):
            flat = torch.cat([g.reshape(-1) for g in bucket])
            work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        last_reduce = dict(path="bucket-copies", launches=len(work), sparse_tables=len(sparse), early_overlapped=False)
    for p in sparse:
        ids = next(v for q, v in sparse_rows.items() if q is p)
        gathered[id(p)] = allgather_row_gradients(p.grad, ids, pad_to=sparse_pad_to)
        if average:
            p.grad.div_(world)
    for h in flat_handles:
        h.wait()
    if average and flat_range is not None:
        flat_range.div_(world)
    for h, flat, bucket in work:
        h.wait()
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off: off + n].view_as(g))
            off += n
    return gathered


def allreduce_sums(sums: torch.Tensor) -> torch.Tensor:
    """Metric sums [HR, NDCG, loss, users, ...] -> summed over ranks (train.py:49-53 across shards)."""
    if _active():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums
