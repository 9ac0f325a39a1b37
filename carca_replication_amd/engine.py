"""One training step / one evaluation batch of the hot path, single- or multi-GPU.

Mirrors the body of the reference's loops (src/train.py:83-97 and :41-51) without its host syncs:
nothing here calls .item(); losses and metric sums stay on the device until the caller reads them.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import dist as cdist
from . import ops
from .modules import BinaryCrossEntropy, get_mask, note_training_forward

_loss_fn = BinaryCrossEntropy()


SPARSE_TABLE_BYTES = 64 * 2 ** 20  # item tables above this exchange (row ids, row gradients) instead of the dense grad


def _sparse_tables(model, p_x, o_x):
    emb = getattr(getattr(model, "embeds", None), "items_embed", None)
    if emb is None or emb.weight.numel() * emb.weight.element_size() < SPARSE_TABLE_BYTES:
        return None
    return {emb.weight: torch.cat([p_x.reshape(-1), o_x.reshape(-1)])}


def as_batch7(batch):
    """(p_x, p_a, p_c, o_x, o_a, o_c, y_true) from what a loader yields: the reference's 7-tuple (train.py:84), or the
    ids-only 5-tuple (p_x, p_c, o_x, o_c, y_true) of CARCADataset(with_attrs=False) -- attribute slots None, filled by the
    model from its registered attribute table."""
    batch = tuple(batch)
    if len(batch) == 5:
        p_x, p_c, o_x, o_c, y_true = batch
        return p_x, None, p_c, o_x, None, o_c, y_true
    if len(batch) != 7:
        raise ValueError(f"a batch is 7 tensors (train.py:84) or 5 without the attribute tensors, got {len(batch)}")
    return batch


def _row_exchange_len(p_x, o_x, global_batch: Optional[int]) -> Optional[int]:
    """ids per rank the row exchange of a big item table pads to, from the host's own knowledge: the largest shard of
    dist.shard_range (ceil(global_batch / world) users) x ids per user.  None = unknown (the ranks ask each other)."""
    if global_batch is None:
        return None
    world = max(cdist.world_size(), 1)
    per_rank = -(-int(global_batch) // world)
    if p_x.shape[0] > per_rank:
        # (checked on the host BEFORE the step's first collective: a rank that found this out inside the row exchange
        # would raise alone while the others wait in all_gather until the backend's timeout)
        raise ValueError(f"this rank holds {p_x.shape[0]} users but global_batch={global_batch} over {world} ranks allows "
                         f"at most {per_rank} per rank (dist.shard_range)")
    return per_rank * (p_x.shape[1] + o_x.shape[1])


def train_step(model, optim, batch, sharded: bool = False, global_batch: Optional[int] = None) -> torch.Tensor:
    """batch = (p_x, p_a, p_c, o_x, o_a, o_c, y_true) as the reference's DataLoader yields (train.py:84).

    With sharded=True the batch holds THIS rank's users; the loss is normalised by the global mask
    count and gradients are summed over ranks, which reproduces the single-process step exactly.
    global_batch: users of the whole step over all ranks (shards as dist.shard_range cuts them); lets a model with a
    big item table size its row exchange without a host sync (dist.allgather_row_gradients).
    Returns the (device) loss: the global batch loss's local share when sharded.
    """
    batch = as_batch7(batch)
    p_x, o_x = batch[0], batch[3]
    gathered = None
    if not (sharded and cdist._active()):
        loss = _forward_backward(model, optim, batch, None)
    else:
        # 1. the loss normaliser of the WHOLE batch (carca.py:443), 2. local forward / backward with it, 3. gradients
        # summed over ranks, in place in the backward's flat buffer -- the early range started under the backward's last
        # kernel: an event recorded right before that launch (CarcaEmbedBwdDesc.ev_early) gates a side stream, RCCL's
        # own stream queues behind the side stream, and the call returns to issue the late range behind the kernel.
        # (global_batch validated ahead of the step's first collective -- by EVERY rank together the first time a pair
        # (local users, global_batch) is seen, so that all of them raise; by this rank alone from then on)
        cdist.validate_global_batch(p_x.shape[0], global_batch, p_x.device if p_x.is_cuda else None)
        _row_exchange_len(p_x, o_x, global_batch)
        denom = cdist.global_mask_count(o_x)
        # (deterministic mode: the early range's sums sit in the fixed-point shadow until the pass ends -- no early start)
        ev = _early_event() if p_x.is_cuda and not ops.deterministic() else None
        ops.early_event = ev
        try:
            loss = _forward_backward(model, optim, batch, denom)
        finally:
            ops.early_event = None
        params = list(model.parameters())
        info = cdist.flat_layout(model, [p for p in params if p.grad is not None])
        early = None
        if info is not None and ev is not None and info["late"][1] > info["late"][0]:
            side = _side_stream(p_x.device)
            from . import _lib

            _lib.check(_lib.load().carca_stream_wait_event(side.cuda_stream, ev.handle), "stream_wait_event")
            with torch.cuda.stream(side):
                early = cdist.allreduce_range(info["flat"], *info["early"])
        gathered = cdist.allreduce_gradients(params, sparse_rows=_sparse_tables(model, p_x, o_x), flat_info=info,
                                             early_work=early, sparse_pad_to=_row_exchange_len(p_x, o_x, global_batch))
        note_exchanged_rows(model, gathered)
    _mark_touched_rows(model, optim, p_x, o_x, gathered)
    optim.step()
    return loss


_EV, _SIDE = [], {}


def _early_event():
    if not _EV:
        _EV.append(ops.HipEvent())
    return _EV[0]


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def note_exchanged_rows(model, gathered: Optional[dict], into: Optional[torch.Tensor] = None) -> None:
    """A big item table's cached gradient buffer is cleared ROW-WISE before the next pass (autograd._grad_buffers) -- of
    the rows this rank scattered into.  The row exchange of a sharded step adds the other ranks' rows to it: tell the
    cache (model._grad_foreign, int32 ids).  into: a fixed buffer to copy the ids into (the hipGraph step: the captured
    clearing launch reads that buffer), padded with the pad row's id 0."""
    if not gathered:
        return
    ids = torch.cat([v.reshape(-1) for v in gathered.values() if v is not None]).to(torch.int32)
    if into is None:
        model.__dict__["_grad_foreign"] = ids
        return
    if ids.numel() > into.numel():
        raise ValueError(f"note_exchanged_rows: {ids.numel()} exchanged ids, the captured step was sized for {into.numel()}")
    into.zero_()
    into[: ids.numel()].copy_(ids)


def _mark_touched_rows(model, optim, p_x, o_x, gathered: Optional[dict] = None) -> None:
    """Big item tables: tell the optimizer which rows this step's gradient can touch (optim.Adam.mark_rows: rows never
    touched are skipped, bit-exactly).  With users sharded over ranks the rows are the union over ranks, because the
    exchanged row gradients land in every replica: `gathered` = {id(table): every rank's ids}, which the row exchange
    itself hands back (dist.allreduce_gradients) -- no second collective."""
    tables = _sparse_tables(model, p_x, o_x)
    if not tables or not hasattr(optim, "mark_rows"):
        return
    for w, ids in tables.items():
        if gathered and gathered.get(id(w)) is not None:
            ids = gathered[id(w)]
        optim.mark_rows(w, ids)


def _forward_backward(model, optim, batch, denom: Optional[torch.Tensor]) -> torch.Tensor:
    """denom: the loss normaliser (device float[1]: the all-reduced mask count of a sharded step), None = this batch's own."""
    p_x, p_a, p_c, o_x, o_a, o_c, y_true = batch
    half = o_x.shape[1] // 2
    pos = tuple(t if t is None else t[:, :half] for t in (o_x, o_a, o_c))  # train.py:86-88
    neg = tuple(t if t is None else t[:, half:] for t in (o_x, o_a, o_c))
    if o_x.shape[1] == 2 * half:  # the kernels want dense [B, L] ids per group: ONE copy for both halves instead of two
        ids2 = o_x.view(o_x.shape[0], 2, half).transpose(0, 1).contiguous()
        pos, neg = (ids2[0],) + pos[1:], (ids2[1],) + neg[1:]
    optim.zero_grad(set_to_none=True)
    y = model(profile=(p_x, p_a, p_c), targets=[pos, neg])
    if o_x.dtype == torch.int32 and y.dim() == 2 and y.shape == o_x.shape and y.is_contiguous():
        # BinaryCrossEntropy(y, y_true, get_mask(o_x)) (carca.py:441-444) and its derivative from ONE kernel, the backward
        # pass seeded with that derivative: no float mask, no autograd node for the loss, no ones-fill / multiply for
        # d loss / d loss = 1
        loss, dy = ops.bce_fwd(y.detach(), y_true, o_x, 1e-8, want_grad=True, denom=denom)
        torch.autograd.backward(y, dy.view_as(y))
        return loss
    loss = _loss_fn(y, y_true, get_mask(o_x), denom=denom)
    loss.backward()
    return loss.detach()


class GraphedTrainStep:
    """train_step with its forward + backward (~40 launches, 1.3-2.0 ms of host time at C2 against 1.1-1.9 ms of GPU
    time) captured ONCE into a hipGraph and replayed per batch; the optimizer's single launch is issued behind each replay.

        step = GraphedTrainStep(model, optim, example_batch)   # shapes and dtypes are fixed from here on
        for batch in loader: loss = step(batch)                  # device loss, overwritten by the next call

    The batch is copied into the graph's own input tensors `.inputs`; a loader that fills those tensors itself and passes
    them back skips the copy (dense C2 batches are 315 MB: ~0.1 ms of HBM time per step; ids-only batches over a registered
    attribute table are a few hundred KB).
    sharded=True (users sharded over ranks): the graph holds forward + backward normalised by a device scalar that every
    call refills with the all-reduced mask count; the gradient all-reduce (in place) and the optimizer run behind each
    replay, outside the graph.  The EARLY range of the gradients (everything but feats_embed.{weight, bias}) is reduced
    under the graph's last kernel, as in the eager sharded step: the capture holds an external event-record node right in
    front of that kernel (CarcaEmbedBwdDesc.ev_early, hipEventRecordExternal), every replay records the event there, and
    a side stream outside the graph waits on it and issues the all-reduce of the early range.
    Dropout: the seeds are launch arguments, which a replay repeats; the graph's first node increments a device counter
    that every dropout kernel adds to its seed (ops.set_dropout_seed_offset), so replay t draws the masks an eager step
    with seed + t would."""

    def __init__(self, model, optim, example_batch, warmup: int = 3, sharded: bool = False,
                 global_batch: Optional[int] = None):
        self.model, self.optim, self.sharded, self.global_batch = model, optim, sharded, global_batch
        example_batch = as_batch7(example_batch)
        if sharded:  # (the shapes are fixed from here on: one check, by every rank together, before anything is captured)
            cdist.validate_global_batch(example_batch[0].shape[0], global_batch,
                                        example_batch[0].device if example_batch[0].is_cuda else None)
        self.inputs = tuple(t if t is None else t.clone() for t in example_batch)
        self.denom = torch.ones(1, dtype=torch.float32, device=self.inputs[0].device) if sharded else None
        if sharded:
            self.denom.copy_(cdist.global_mask_count(self.inputs[3]))
        self.foreign = None
        if sharded and cdist._active() and _sparse_tables(model, self.inputs[0], self.inputs[3]):
            # (a big item table: the exchanged rows of every rank, cleared by the captured pass -- see note_exchanged_rows)
            per_rank = _row_exchange_len(self.inputs[0], self.inputs[3], global_batch) or \
                (self.inputs[0].numel() + self.inputs[3].numel())
            self.foreign = torch.zeros(cdist.world_size() * per_rank, dtype=torch.int32, device=self.inputs[0].device)
            model.__dict__["_grad_foreign"] = self.foreign
        # (the early-gradients event: not in deterministic mode, whose sums sit in the fixed-point shadow until the pass ends)
        self.ev_early = ops.HipEvent() if (sharded and self.inputs[0].is_cuda and not ops.deterministic()) else None
        self.replays = torch.zeros(1, dtype=torch.int64, device=self.inputs[0].device)
        # (a model without dropout draws no masks: no counter node in its graph -- 4.6 us per replay)
        has_dropout = any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules())
        ops.set_dropout_seed_offset(self.replays)
        ops.early_event = self.ev_early
        try:
            if self.inputs[0].is_cuda:  # (the captured backward's side streams must exist before the capture begins)
                from . import autograd

                autograd.ensure_side_streams(self.inputs[0].device)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # lazily built state (code objects, workspaces, descriptor rings) first
                from . import autograd as _ag

                _ag.GRAPH_WARMUP[0] = True  # (the warm-up steps take the captured pass's prep fork: see autograd.EARLY_PREP)
                try:
                    for _ in range(warmup):
                        _forward_backward(model, optim, self.inputs, self.denom)
                finally:
                    _ag.GRAPH_WARMUP[0] = False
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.scope = 0
            with torch.cuda.graph(self.graph):
                # (what the library allocates for the captured kernels -- partial tiles, row tables, descriptor copies:
                # ~170-200 MB at C2 -- belongs to this capture and is handed back by close())
                self.scope = ops.capture_scope()
                if has_dropout:
                    self.replays.add_(1)
                self.loss = _forward_backward(model, optim, self.inputs, self.denom)
            self.library_bytes = ops.capture_bytes(self.scope)
            if self.ev_early is not None:
                from . import _lib

                if not _lib.load().carca_early_event_recorded():  # (this HIP runtime adds no external event node)
                    self.ev_early = None
                    self.ev_early_note = _lib.load().carca_last_error().decode(errors="replace")
        finally:
            ops.set_dropout_seed_offset(None)
            ops.early_event = None
        # The replayed backward writes into the gradient tensors of the capture.  An eager step in between (train() sends
        # the short last batch of an epoch through train_step) rebinds every p.grad to fresh memory: remember the graph's
        # own tensors and hand them back to the parameters before each optimizer step.
        self.params = [p for p in model.parameters() if p.grad is not None]
        self.grads = [p.grad for p in self.params]
        # (a model with a big item table: the captured kernels hold pointers into the gradient cache of the capture --
        # keep it alive even if an eager step of another shape replaces the model's current one)
        self._grad_cache = model.__dict__.get("_grad_cache")

    def close(self) -> None:
        """Destroys the graph and frees the library memory its kernels were given (idempotent).  Replays still in flight
        are waited for first: the graph's kernels read that memory."""
        if getattr(self, "graph", None) is not None:
            torch.cuda.synchronize()
            self.graph = None
            ops.capture_release(self.scope)
            self.scope = 0

    def __del__(self):
        # (the collector may run this INSIDE another capture: a synchronize or a free there would invalidate it -- leave the
        # graph's memory to an explicit close(), or to the end of the process)
        try:
            if not torch.cuda.is_current_stream_capturing():
                self.close()
        except Exception:
            pass

    def __call__(self, batch) -> torch.Tensor:
        ops.poll_errors()  # (a kernel-side failure of an EARLIER replay that no launch status carries; one host read)
        for dst, src in zip(self.inputs, as_batch7(batch)):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if self.sharded:
            self.denom.copy_(cdist.global_mask_count(self.inputs[3]))
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                p.grad = g
        gathered = None
        if self.sharded:
            info = cdist.flat_layout(self.model, self.params)
            early = None
            if cdist._active() and info is not None and self.ev_early is not None and info["late"][1] > info["late"][0]:
                # the replay just queued records ev_early in front of its last kernel: the early range goes out under it
                side = _side_stream(self.inputs[0].device)
                from . import _lib

                _lib.check(_lib.load().carca_stream_wait_event(side.cuda_stream, self.ev_early.handle), "stream_wait_event")
                with torch.cuda.stream(side):
                    early = cdist.allreduce_range(info["flat"], *info["early"])
            gathered = cdist.allreduce_gradients(
                self.params, sparse_rows=_sparse_tables(self.model, self.inputs[0], self.inputs[3]),
                flat_info=info, early_work=early,
                sparse_pad_to=_row_exchange_len(self.inputs[0], self.inputs[3], self.global_batch))
            if self.foreign is not None:
                self.model.__dict__["_grad_foreign"] = self.foreign
                note_exchanged_rows(self.model, gathered, into=self.foreign)
        note_training_forward()  # the optimizer below rewrites the weights: packed inference copies are stale
        _mark_touched_rows(self.model, self.optim, self.inputs[0], self.inputs[3], gathered)
        self.optim.step()
        return self.loss


@torch.no_grad()
def eval_batch(model, batch, k: int = 10, sums: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Scores one (p_x, p_a, p_c, o_x, o_a, o_c, y_true) eval batch (train.py:42-51); accumulates
    [HR@k sum, NDCG@k sum, ties, loss sum, users] into `sums` (device float[5]) with no host sync -- one launch
    (carca_eval_metrics) for batches of up to 16384 scores, else the rank kernel, the loss kernel and two adds."""
    p_x, p_a, p_c, o_x, o_a, o_c, y_true = as_batch7(batch)
    y = model(profile=(p_x, p_a, p_c), targets=[(o_x, o_a, o_c)])
    y2 = y.reshape(p_x.shape[0], -1)
    if sums is None:
        sums = torch.zeros(5, dtype=torch.float32, device=y.device)
    if (o_x.dtype == torch.int32 and y2.is_contiguous() and y2.numel() <= ops.EVAL_METRICS_MAX and sums.is_contiguous()
            and y_true.shape == y2.shape):
        # HR / NDCG / loss / user count of the batch in ONE launch (the positive is candidate 0, data.py:165,190)
        ops.eval_metrics(y2, y_true, o_x, k, sums)
        return y, sums
    ops.rank_metrics(y2, k, sums=sums[:3])
    if o_x.dtype == torch.int32 and y2.is_contiguous():  # the loss kernel masks by ids != 0 itself (no float mask)
        loss, _ = ops.bce_fwd(y2, y_true, o_x, 1e-8)
    else:
        loss = _loss_fn(y2, y_true, get_mask(o_x))
    sums[3] += loss
    sums[4] += p_x.shape[0]
    return y, sums
