"""Load the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    groups = {"cfg": {}, "dim": {}, "param": {}, "in": {}, "out": {}}
    for k in z.files:
        head, rest = k.split("/", 1)
        v = z[k]
        if head in ("cfg", "dim"):
            groups[head][rest] = v.item()
        elif v.dtype.kind in "fiub" and head in ("param", "in", "out"):
            groups[head][rest] = torch.from_numpy(np.ascontiguousarray(v))
        else:
            groups[head][rest] = v
    return SimpleNamespace(cfg=groups["cfg"], dim=groups["dim"], params=groups["param"], ins=groups["in"],
                           outs=groups["out"])


def oracle_config(cfg):
    from oracle.carca_oracle import CarcaConfig

    return CarcaConfig(d=int(cfg["d"]), H=int(cfg["H"]), n_blocks=int(cfg["n_blocks"]),
                       residual_sa=bool(cfg.get("residual_sa", True)), residual_ca=bool(cfg.get("residual_ca", True)),
                       encoding=str(cfg.get("encoding", "identity")), embedding=str(cfg.get("embedding", "all")),
                       decoder=str(cfg.get("decoder", "ca")), gamma=float(cfg.get("gamma", 0.9)),
                       l2_norm=bool(cfg.get("l2_norm", False)))


G1_NAMES = ["d90h3", "d90h2", "d128h4", "d64h2"]
G7_NAMES = ["learnable", "positional", "nores"]
G9_NAMES = ["attrctx", "attr", "id", "mlpid", "dot", "wdot", "wdotnorm", "iddot"]
