"""Train / evaluate driver with the reference's contract (src/train.py of r-papso/carca-replication).

Same function names, arguments, return values, console lines and `;`-separated CSV log as the reference
(train.py:56-152), so `scripts/training.py:176-186` drives it unchanged.  What differs underneath:
  * every batch runs the HIP path (engine.train_step / engine.eval_batch);
  * HR@k / NDCG@k come from a sort-free rank count on the device (carca_rank_metrics) and all per-batch
    `.item()` syncs (train.py:21,32,47,97) are gone: sums stay on the device until an epoch / evaluation ends;
  * the best checkpoint is written with torch.save(model) like the reference (train.py:124) but reloaded with
    weights_only=False -- the reference's bare torch.load (train.py:142) raises on torch >= 2.6.
"""
from __future__ import annotations

import os
from datetime import datetime
from typing import Tuple, Union

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import _LRScheduler
from torch.utils.data import DataLoader

from . import engine, ops
from .modules import Model, to


def _positive_columns(y_true: torch.Tensor) -> torch.Tensor:
    return y_true.argmax(dim=1).to(torch.int32)


def compute_HR(y_pred: torch.Tensor, y_true: torch.Tensor, k: int) -> float:
    """Number of rows whose positive (the 1 in y_true) ranks in the top k (train.py:15-21), tie-free scores."""
    sums, _ = ops.rank_metrics(y_pred, k, pos=_positive_columns(y_true))
    return float(sums[0])


def compute_NDCG(y_pred: torch.Tensor, y_true: torch.Tensor, k: int) -> float:
    """Sum over rows of 1 / log2(rank + 2) for positives ranked in the top k (train.py:24-32)."""
    sums, _ = ops.rank_metrics(y_pred, k, pos=_positive_columns(y_true))
    return float(sums[1])


def evaluate(model: Model, loader: DataLoader, device: str, k: int) -> Tuple[float, float, float]:
    """(HR@k, NDCG@k, mean batch loss) over the loader (train.py:35-53); one host sync at the end."""
    model = model.eval().to(device)
    sums = torch.zeros(5, dtype=torch.float32, device=device)
    n_batches = 0
    with torch.no_grad():
        for batch in loader:
            engine.eval_batch(model, to(*batch, device=device), k=k, sums=sums)
            n_batches += 1
    hr, ndcg, _ties, loss_sum, users = (float(v) for v in sums.cpu())
    return hr / users, ndcg / users, loss_sum / max(n_batches, 1)


def train(model: Model, train_loader: DataLoader, val_loader: DataLoader, test_loader: DataLoader, device: str,
          optim: Optimizer, epochs: int, top_k: int = 10, verbose: int = 1, early_stop: int = 10,
          datadir: str = "model", scheduler: Union[_LRScheduler, None] = None, graphed: bool = False) -> Model:
    """src/train.py:56-152.  graphed (extension, off by default): batches of the first batch's shape replay their forward +
    backward from one hipGraph (engine.GraphedTrainStep); any other shape (a short last batch) takes the eager step."""
    os.makedirs(datadir, exist_ok=True)
    model = model.train().to(device)
    best, stale = 0.0, 0
    t0 = datetime.now()
    log = open(f"./{datadir}/{t0.year}-{t0.month}-{t0.day}T{t0.hour}-{t0.minute}-{t0.second}.csv", "a")
    now = lambda: datetime.now().strftime("%H:%M:%S")  # noqa: E731
    epoch = 0
    captured = None
    for epoch in range(1, epochs + 1):
        loss_sum = torch.zeros((), dtype=torch.float32, device=device)
        for i, batch in enumerate(train_loader, start=1):
            dev_batch = to(*batch, device=device)
            if graphed:
                sig = tuple((tuple(t.shape), t.dtype) for t in dev_batch)
                if captured is None:
                    captured = (sig, engine.GraphedTrainStep(model, optim, dev_batch))
            if graphed and captured[0] == sig:
                loss_sum += captured[1](dev_batch)
            else:
                loss_sum += engine.train_step(model, optim, dev_batch)
            if verbose == 2:  # the reference prints a running mean per batch; that costs a sync per batch here too
                print(f"{now()} - Batch {i:03d}: Loss = {(float(loss_sum) / i):.4f}")
        mean_loss = float(loss_sum) / len(train_loader)
        if verbose in (1, 2):
            print(f"{now()} - Epoch {epoch:03d}: Train Loss = {mean_loss:.4f}")
            log.write(f"{now()};{epoch};train;{mean_loss};;\n")
        if scheduler is not None:
            scheduler.step()

        HR, NDCG, loss = evaluate(model, val_loader, device, top_k)
        model = model.train().to(device)
        if NDCG > best:
            for f in os.listdir(datadir):
                if f.endswith(".pth"):
                    os.remove(os.path.join(datadir, f))
            best, stale = NDCG, 0
            torch.save(model, os.path.join(datadir, f"{epoch:03d}_{HR:.4f}_{NDCG:.4f}.pth"))
        else:
            stale += 1
        if verbose in (1, 2):
            print(f"{now()} - Epoch {epoch:03d}: Val Loss = {loss:.4f} HR = {HR:.4f}, NDCG = {NDCG:.4f}")
            log.write(f"{now()};{epoch};val;{loss};{HR};{NDCG}\n")
        if stale >= early_stop:
            print(f"No improvement in {stale} epochs, early stopping...")
            break
        log.flush()

    saved = [f for f in os.listdir(datadir) if f.endswith(".pth")]
    if saved:
        model = torch.load(os.path.join(datadir, saved[0]), weights_only=False).to(device)
    if test_loader is not None:
        HR, NDCG, loss = evaluate(model, test_loader, device, top_k)
        print(f"{now()} - Epoch {epoch:03d}: Test Loss = {loss:.4f} HR = {HR:.4f}, NDCG = {NDCG:.4f}")
        log.write(f"{now()};{epoch};test;{loss};{HR};{NDCG}\n")
    log.close()
    return model
