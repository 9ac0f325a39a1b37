"""Eval throughput (users/s, one GPU) of the configurations SURVEY.md section 8d names besides C2 -- the parity-test cases of
BASELINE.json -- to see that kernel selection holds up away from the headline shape.  Synthetic data, random weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.carca_oracle import synth_eval_batch  # noqa: E402
from tests.model_util import build_model  # noqa: E402

CONFIGS = {
    "C2  B=128 d=90  g=450 H=3 n_attrs=4096 N=101": dict(B=128, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C3  B=512 (same per-user dims)              ": dict(B=512, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C4  B=128 d=128 g=640 H=4 n_attrs=4096      ": dict(B=128, L=50, N=101, d=128, g=640, H=4, nb=2, n_attrs=4096, n_ctx=6, n_items=12102),
    "C5  B=128 n_attrs=512 N=1001                ": dict(B=128, L=50, N=1001, d=90, g=450, H=3, nb=2, n_attrs=512, n_ctx=6, n_items=12102),
    "CLI B=256 d=64 g=256 H=2 3 blocks n_attrs=512": dict(B=256, L=50, N=101, d=64, g=256, H=2, nb=3, n_attrs=512, n_ctx=6, n_items=12102),
    "small n_attrs=64                             ": dict(B=128, L=50, N=101, d=90, g=450, H=3, nb=2, n_attrs=64, n_ctx=6, n_items=12102),
}
results = []
only = os.environ.get("ONLY")  # e.g. ONLY=C5 (prefix of the configuration's name) for a rocprofv3 run of one of them
for name, c in CONFIGS.items():
    if only and not name.startswith(only):
        continue
    torch.manual_seed(0)
    model = build_model(dict(d=c["d"], H=c["H"], n_blocks=c["nb"]), c["n_items"], c["g"], c["n_ctx"], c["n_attrs"], c["L"]).cuda().eval()
    profile, target, _ = synth_eval_batch(c["B"], c["L"], c["N"], c["n_items"], c["n_attrs"], c["n_ctx"], seed=1)
    profile, target = tuple(t.cuda() for t in profile), tuple(t.cuda() for t in target)
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:  # pre-heat as bench.py does: the clock needs tens of ms of load to settle
            for _ in range(8):
                model(profile=profile, targets=[target])
            torch.cuda.synchronize()
        steps = 100
        t0 = time.perf_counter()
        for _ in range(steps):
            model(profile=profile, targets=[target])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # per-stage kernel time of the same forward (HIP events bound to the launches: feature GEMM, scoring kernel, the FIRST
    # SelfAttentionBlock's kernel, joint GEMM -- the hooks bench.py's roofline keys use) and which row kernels the products took
    from carca_replication_amd import ops

    evs = [[ops.HipEvent() for _ in range(8)] for _ in range(12)]
    with torch.no_grad():
        ops.gemm_rows_log(True)
        model(profile=profile, targets=[target])
        rows_log = ops.gemm_rows_log()
        ops.gemm_rows_log(False)
        for e in evs:
            ops.set_fused_events([h.handle for h in e])
            model(profile=profile, targets=[target])
        ops.set_fused_events(None)
        torch.cuda.synchronize()
    med = lambda i: sorted(e[i].elapsed_ms(e[i + 1]) for e in evs)[len(evs) // 2]  # noqa: E731
    B_, L_, N_, d_, g_, H_, nb_ = c["B"], c["L"], c["N"], c["d"], c["g"], c["H"], c["nb"]
    F = c["n_attrs"] + c["n_ctx"]
    rows = B_ * (L_ + N_)
    stages = {  # name: (ms per launch, launches per forward, algorithmic flop per launch)
        "feature_gemm": (med(0), 1, 2.0 * rows * F * g_),
        "cross_score": (med(2), 1, B_ * (4.0 * L_ * d_ * d_ + 2.0 * N_ * d_ * d_ + 4.0 * N_ * L_ * d_ + 2.0 * N_ * d_)),
        "sa_block": (med(4), nb_, B_ * (10.0 * L_ * d_ * d_ + 4.0 * L_ * L_ * d_)),
        "joint_gemm": (med(6), 1, 2.0 * rows * (d_ + g_) * d_),
    }
    dominant = max(stages, key=lambda k: stages[k][0] * stages[k][1])
    feat_kernel = next((t for t in rows_log.split(";") if f"N={g_} " in t), "?")
    joint_kernel = next((t for t in rows_log.split(";") if f"N={d_} K={d_ + g_} " in t or f"N={d_} K={g_} " in t), "?")  # (K = g: the item term from the projected table)
    LEAVES_ROWS_OUT = ("gemm_rows_skc_kernel", "gemm_rows_cuc_kernel", "gemm_rows_n96c_kernel")
    kname = {"feature_gemm": feat_kernel, "joint_gemm": joint_kernel, "cross_score": "cross_stream_kernel / cross_fold_kernel",
             "sa_block": "sa_eval_kernel"}
    # the row products leave rows with id 0 out (exactly: carca.py:94 zeroes them): `frac` counts the flops EXECUTED -- the
    # algorithmic figure x the share of rows multiplied -- and can never exceed 1; `effective_frac` counts every padded row
    kept = float((profile[0] != 0).sum() + (target[0] != 0).sum()) / float(profile[0].numel() + target[0].numel())
    stage_out = {k: dict(ms=v[0], launches=v[1], effective_tflops=v[2] / v[0] / 1e9, effective_frac=v[2] / v[0] / 1e9 / 157.3)
                 for k, v in stages.items()}
    for k, kern in (("feature_gemm", feat_kernel), ("joint_gemm", joint_kernel)):
        share = kept if any(n in kern for n in LEAVES_ROWS_OUT) else 1.0
        stage_out[k].update(rows_multiplied_share=share, tflops=stage_out[k]["effective_tflops"] * share,
                            frac=stage_out[k]["effective_frac"] * share)
    for k in ("cross_score", "sa_block"):  # (these execute FEWER flops than counted -- folded value projection, pad slots --: effective figures only)
        stage_out[k].update(tflops=stage_out[k]["effective_tflops"], frac=stage_out[k]["effective_frac"])
    flop = c["B"] * ((c["L"] + c["N"]) * (2 * F * c["g"] + 2 * (c["d"] + c["g"]) * c["d"]) + c["nb"] * (10 * c["L"] * c["d"] ** 2 + 4 * c["L"] ** 2 * c["d"])
                     + 2 * c["N"] * c["d"] ** 2 + 4 * c["L"] * c["d"] ** 2 + 4 * c["N"] * c["L"] * c["d"] + 2 * c["N"] * c["d"])
    results.append(dict(config=" ".join(name.split()), ms_per_batch=dt * 1e3, users_per_s=c["B"] / dt,
                        algorithmic_tflops=flop / dt / 1e12, stages=stage_out,
                        dominant=dict(stage=dominant, kernel=kname[dominant], share_of_step=stages[dominant][0] * stages[dominant][1] / (dt * 1e3),
                                      roofline=dict(bound="mfma", peak=157.3, unit="TFLOP/s", achieved=stage_out[dominant]["tflops"],
                                                    frac=stage_out[dominant]["frac"],
                                                    effective_frac=stage_out[dominant]["effective_frac"])), **c))
    print(f"{name}: {dt*1e3:7.3f} ms/batch  {c['B']/dt:10.0f} users/s  {flop/dt/1e12:6.1f} TFLOP/s (algorithmic); dominant: "
          f"{dominant} [{kname[dominant]}] {stages[dominant][0]*1e3:.1f} us x{stages[dominant][1]} = "
          f"{100*stage_out[dominant]['frac']:.1f} % of the fp32 MFMA peak on executed flops ({100*stage_out[dominant]['effective_frac']:.1f} % effective); "
          + ", ".join(f"{k} {v[0]*1e3:.1f} us" for k, v in stages.items()), flush=True)

if os.environ.get("JSON_OUT"):  # e.g. JSON_OUT=gpurun_out/r03_configs.json (copied to profiles/ by hand)
    import json

    with open(os.environ["JSON_OUT"], "w") as fh:
        json.dump(dict(what="eval forward, one MI355X, inputs resident in HBM, 0.3 s pre-heat + 100 steps (tools/bench_configs.py)",
                       peak_fp32_mfma_tflops=157.3, results=results), fh, indent=1)
