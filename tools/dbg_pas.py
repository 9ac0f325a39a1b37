import os, sys, torch
sys.path.insert(0, os.getcwd())
from carca_replication_amd import ops
from oracle.carca_oracle import synth_eval_batch
from tests.model_util import build_model
torch.manual_seed(0)
model = build_model(dict(d=90, H=3, n_blocks=2), 12102, 450, 6, 4096, 50).cuda().eval()
profile, target, _ = synth_eval_batch(128, 50, 101, 12102, 4096, 6, seed=1)
profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
def run(v):
    ops.set_tuning(0, v)
    y = model(profile=profile, targets=[target]).clone()
    plan = model.__dict__["_plan"]
    out = [y] + [t.clone() for t in plan["es"]] + [t.clone() for t in plan["xw"]]
    ops.set_tuning(0, 0)
    return out
names = ["y", "e_profile", "e_target", "xw0", "xw1"]
with torch.no_grad():
    a = run(0); b = run(0); c = run(19); d = run(19); e = run(99)
    for nm, lst in (("0 vs 0", (a, b)), ("19 vs 19", (c, d)), ("0 vs 19", (a, c)), ("0 vs 99", (a, e))):
        print(nm, {n: float((x - y).abs().max()) for n, x, y in zip(names, *lst)})
