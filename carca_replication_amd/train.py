"""Train / evaluate driver with the reference's contract (src/train.py of r-papso/carca-replication).

Same function names, arguments, return values, console lines and `;`-separated CSV log as the reference
(train.py:56-152), so `scripts/training.py:176-186` drives it unchanged.  What differs underneath:
  * every batch runs the HIP path (engine.train_step / engine.eval_batch);
  * HR@k / NDCG@k come from a sort-free rank count on the device (carca_rank_metrics) and all per-batch
    `.item()` syncs (train.py:21,32,47,97) are gone: sums stay on the device until an epoch / evaluation ends;
  * the best checkpoint keeps the reference's file name (`<epoch>_<HR>_<NDCG>.pth`, train.py:118-124) but holds
    state_dicts -- model, optimizer, scheduler, epoch, metrics -- instead of a pickled module: it reloads with
    torch.load(weights_only=True) (the reference's bare torch.load of a pickled module, train.py:141-142, raises on
    torch >= 2.6), survives refactors of the module classes, and `resume=` continues a run from it.
    load_checkpoint() still accepts the reference's pickled-module files.
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


CHECKPOINT_FORMAT = "carca-state-dict-v1"


def save_checkpoint(path: str, model: Model, optim: Optimizer = None, scheduler=None, **meta) -> None:
    """state_dict checkpoint: tensors + plain python values only, so that
    torch.load(path, weights_only=True) reads it back."""
    ck = {"format": CHECKPOINT_FORMAT, "model": model.state_dict(), "meta": dict(meta)}
    if optim is not None:
        ck["optimizer"] = optim.state_dict()
    if scheduler is not None:
        ck["scheduler"] = scheduler.state_dict()
    tmp = path + ".tmp"
    torch.save(ck, tmp)
    os.replace(tmp, path)  # (an interrupted save never leaves a truncated best checkpoint behind)


def load_checkpoint(path: str, model: Model, optim: Optimizer = None, scheduler=None, device=None,
                    allow_pickle: bool = False) -> dict:
    """Loads `path` into model (and optimizer / scheduler when given and stored); returns the checkpoint's meta dict.
    The file is read with torch.load(weights_only=True): nothing in it is executed, and a truncated or foreign file
    raises.  allow_pickle=True (opt-in, for files the caller trusts) also accepts what the reference writes --
    torch.save(model), a pickled module (train.py:124) -- and copies its parameters; unpickling runs code from the file."""
    if allow_pickle:
        try:
            ck = torch.load(path, map_location=device, weights_only=True)
        except Exception:
            ck = torch.load(path, map_location=device, weights_only=False)
    else:
        ck = torch.load(path, map_location=device, weights_only=True)
    if isinstance(ck, torch.nn.Module):
        model.load_state_dict(ck.state_dict())
        return {}
    if not isinstance(ck, dict) or ck.get("format") != CHECKPOINT_FORMAT:
        raise ValueError(f"{path}: not a {CHECKPOINT_FORMAT} checkpoint")
    model.load_state_dict(ck["model"])
    if optim is not None and "optimizer" in ck:
        optim.load_state_dict(ck["optimizer"])
    if scheduler is not None and "scheduler" in ck:
        scheduler.load_state_dict(ck["scheduler"])
    return ck.get("meta", {})


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
    """(HR@k, NDCG@k, mean batch loss) over the loader (train.py:35-53); one host sync at the end.
    The loader yields the reference's 7-tuples, ids-only 5-tuples (CARCADataset(with_attrs=False)) or is a
    device_data.DeviceLoader (batches built in HBM; the model needs its attribute table: register_attr_table)."""
    model = model.eval().to(device)
    sums = torch.zeros(5, dtype=torch.float32, device=device)
    n_batches = 0
    with torch.no_grad():
        for batch in loader:
            engine.eval_batch(model, to(*engine.as_batch7(batch), device=device), k=k, sums=sums)
            n_batches += 1
    hr, ndcg, _ties, loss_sum, users = (float(v) for v in sums.cpu())
    if torch.device(device).type == "cuda":
        ops.poll_errors()  # (the epoch's only host sync just happened: a kernel-side failure of any batch surfaces here)
    return hr / users, ndcg / users, loss_sum / max(n_batches, 1)


def train(model: Model, train_loader: DataLoader, val_loader: DataLoader, test_loader: DataLoader, device: str,
          optim: Optimizer, epochs: int, top_k: int = 10, verbose: int = 1, early_stop: int = 10,
          datadir: str = "model", scheduler: Union[_LRScheduler, None] = None, graphed: bool = False,
          resume: Union[str, None] = None, resume_trusted: bool = False) -> Model:
    """src/train.py:56-152.  Extensions, off by default:
    graphed: batches of the first batch's shape replay their forward + backward from one hipGraph
      (engine.GraphedTrainStep); any other shape (a short last batch) takes the eager step;
    resume: path of a checkpoint written by an earlier run (save_checkpoint): model, optimizer and scheduler state are
      restored and the epoch count continues after the stored one, with the stored best NDCG as the bar to beat;
    resume_trusted: the file named by `resume` may be a whole-module pickle as the reference writes them (train.py:124):
      load_checkpoint(allow_pickle=True) -- unpickling executes code from the file, so only for files the caller trusts."""
    os.makedirs(datadir, exist_ok=True)
    model = model.train().to(device)
    best, stale, first_epoch = 0.0, 0, 1
    best_path = None  # the checkpoint holding the best weights so far: a resumed run starts with the one it resumes from
    if resume is not None:
        meta = load_checkpoint(resume, model, optim, scheduler, device=device, allow_pickle=resume_trusted)
        best, first_epoch = float(meta.get("NDCG", 0.0)), int(meta.get("epoch", 0)) + 1
        stale, best_path = int(meta.get("stale", 0)), resume
    t0 = datetime.now()
    log = open(f"./{datadir}/{t0.year}-{t0.month}-{t0.day}T{t0.hour}-{t0.minute}-{t0.second}.csv", "a")
    now = lambda: datetime.now().strftime("%H:%M:%S")  # noqa: E731
    epoch = first_epoch - 1
    captured = None
    for epoch in range(first_epoch, epochs + 1):
        loss_sum = torch.zeros((), dtype=torch.float32, device=device)
        for i, batch in enumerate(train_loader, start=1):
            dev_batch = to(*engine.as_batch7(batch), device=device)
            if graphed:
                sig = tuple(None if t is None else (tuple(t.shape), t.dtype) for t in dev_batch)
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
            best_path = os.path.join(datadir, f"{epoch:03d}_{HR:.4f}_{NDCG:.4f}.pth")
            save_checkpoint(best_path, model, optim, scheduler,
                            epoch=epoch, HR=float(HR), NDCG=float(NDCG), val_loss=float(loss), stale=0)
        else:
            stale += 1
        if verbose in (1, 2):
            print(f"{now()} - Epoch {epoch:03d}: Val Loss = {loss:.4f} HR = {HR:.4f}, NDCG = {NDCG:.4f}")
            log.write(f"{now()};{epoch};val;{loss};{HR};{NDCG}\n")
        if stale >= early_stop:
            print(f"No improvement in {stale} epochs, early stopping...")
            break
        log.flush()

    if best_path is not None and os.path.exists(best_path):
        # the best epoch's weights (train.py:141-142) -- of THIS run, or the checkpoint it resumed from when no epoch
        # beat it; the optimizer keeps the last epoch's state
        load_checkpoint(best_path, model, device=device)
        model = model.to(device)
    if test_loader is not None:
        HR, NDCG, loss = evaluate(model, test_loader, device, top_k)
        print(f"{now()} - Epoch {epoch:03d}: Test Loss = {loss:.4f} HR = {HR:.4f}, NDCG = {NDCG:.4f}")
        log.write(f"{now()};{epoch};test;{loss};{HR};{NDCG}\n")
    log.close()
    return model
