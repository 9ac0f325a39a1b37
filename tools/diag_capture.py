"""Diagnosis of a capture crash: the small graphed train step of tests/test_hip_graph.py, with prints."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from carca_replication_amd import autograd, engine  # noqa: E402
from tests.test_hip_graph import _setup  # noqa: E402

print("EARLY_PREP =", autograd.EARLY_PREP, flush=True)
mode = os.environ.get("DIAG_MODE", "")
if mode == "noprepfn":  # the helper bypassed: prepare inline by calling it under another name (same code) -- sanity
    pass
fresh, batch = _setup(0.0)
if os.environ.get("DIAG_PATCH"):
    from carca_replication_amd import ops

    ops.new_dropout_seed = lambda: 123456789
model, opt = fresh()
print("building step", flush=True)
step = engine.GraphedTrainStep(model, opt, batch)
print("built", flush=True)
for _ in range(3):
    loss = step(batch)
torch.cuda.synchronize()
print("ok", float(loss), flush=True)
