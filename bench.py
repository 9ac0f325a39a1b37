#!/usr/bin/env python3
"""Headline benchmark: scored users/sec (1 + 100 candidates) of the CARCA eval forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (CARCA.forward, eval mode, reference call shape train.py:44)
over one batch of synthetic Beauty-shaped users already resident in HBM: BASELINE.json configs[1]
= SURVEY.md 8d "C2" (B=128 users, L=50, N=1+100, d=90, g=450, H=3, 2 blocks, n_attrs=4096, n_ctx=6,
n_items=12,102).  Users shard across ranks with no data-path collective (weak scaling: every rank
scores its own 128-user batch); the only collectives are the timing barrier and the max-over-ranks.

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel (the F->g feature GEMM of
AllEmbedding, 97% of the model's flops): the flops it EXECUTES per launch (rows with id != 0; `effective_*` = SURVEY 8d's
algorithmic flops over every padded row; `*_full_profiles` = the same step with nothing to leave out) / its mean duration measured
with events bound to its own dispatch on the launch stream inside the timed region (every fourth step:
a pair costs ~5 us per step); the smaller kernels' entries come from a pass right behind it.  `cpu_baseline` is the CPU oracle
(oracle/carca_oracle.py, a port of the reference's PyTorch-CPU path) on all host cores.  `other_configs` (side pass, one GPU):
BASELINE.json's C3 and C5 configurations by the same protocol -- never `value`.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C2 = dict(B=128, L=50, N=101, d=90, g=450, H=3, n_blocks=2, n_attrs=4096, n_ctx=6, n_items=12102)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
PREHEAT_S = 0.2  # untimed load before the warm-up steps (the clock ramps for tens of ms after an idle period)
PEAK_HBM_GBS = 8000.0


def flops_per_user(c):
    """SURVEY.md 8d algorithmic flops per user (eval), 2 flops per MAC."""
    L, N, d, g, F = c["L"], c["N"], c["d"], c["g"], c["n_attrs"] + c["n_ctx"]
    E = (L + N) * (2 * F * g + 2 * (d + g) * d)
    SA = c["n_blocks"] * (10 * L * d * d + 4 * L * L * d)
    CA = 2 * N * d * d + 4 * L * d * d + 4 * N * L * d + 2 * N * d
    feat = (L + N) * 2 * F * g
    return dict(total=E + SA + CA, embed=E, sa=SA, ca=CA, feat=feat)


def build_inputs(c, seed, device):
    """SURVEY.md 8d synthetic inputs, generated on the host with numpy, moved to HBM once."""
    import torch

    from carca_replication_amd.synth import eval_batch

    profile, target, _ = eval_batch(c["B"], c["L"], c["N"], c["n_items"], c["n_attrs"], c["n_ctx"], seed=seed)
    to = lambda t: tuple(x.to(device) for x in t)  # noqa: E731
    return profile, target, to(profile), to(target)


def build_model(c, device):
    import torch

    from carca_replication_amd import modules as M

    torch.manual_seed(0)  # the reference's factories (training.py:76-100, 165-172), random-init weights
    emb = M.AllEmbedding(c["n_items"], c["d"], c["g"], c["n_ctx"], c["n_attrs"], M.IdentityEncoding())
    blocks = torch.nn.ModuleList([M.SelfAttentionBlock(c["d"], c["H"], 0.0, True) for _ in range(c["n_blocks"])])
    dec = M.CrossAttentionBlock(c["d"], c["H"], 0.0, True)
    return M.CARCA(d=c["d"], p=0.0, emb=emb, enc=blocks, dec=dec).eval().to(device)


def host_cores():
    """CPU threads this process may really use: cgroup quota, else affinity, else cpu_count."""
    env = os.environ.get("CARCA_CPU_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(c, model, profile, target, budget_s=15.0):
    """The CPU oracle (port of the reference's PyTorch-CPU forward) on all host cores, bounded sample."""
    import torch

    from oracle import carca_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.CarcaConfig(d=c["d"], H=c["H"], n_blocks=c["n_blocks"])
    with torch.no_grad():
        for _ in range(2):
            O.carca_forward(P, cfg, profile, [target], training=False)
        times = []
        t_start = time.perf_counter()
        while len(times) < 40 and (time.perf_counter() - t_start < budget_s or len(times) < 3):
            t0 = time.perf_counter()
            O.carca_forward(P, cfg, profile, [target], training=False)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": c["B"] / med, "unit": "users/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} batches of {c['B']} users (same C2 tensors, in RAM), median; "
                      f"oracle/carca_oracle.py on torch-CPU, {cores} threads"}


def timed_loop(fn, steps, warmup, fence, preheat_s=PREHEAT_S, preheat_calls=None):
    """The protocol of the headline for every side measurement: `preheat_s` of the same call untimed (the clock needs tens
    of milliseconds of load after an idle period -- building the next measurement's inputs is one), W warm-up calls,
    then `steps` calls between two fences.  Returns seconds for the `steps` calls.
    preheat_calls: a fixed number of pre-heat calls instead of a duration -- for calls that contain collectives, where
    every rank must make the same number of them."""
    import torch

    for _ in range(preheat_calls or 0):
        fn()
    t_heat = time.perf_counter()
    while preheat_calls is None and time.perf_counter() - t_heat < preheat_s:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    return time.perf_counter() - t0


def train_flops_per_user(c):
    """NEEDED flops of one train step per user (2 per MAC): rows = 3 L (profile + L positives + L negatives).  Forward as
    SURVEY 8d with N = 2 L targets; backward = for every linear map its weight gradient (as many flops as the forward
    product) + its input gradient WHERE one is needed: the two big products of AllEmbedding feed on inputs (attrs, ctx,
    item rows) that take no gradient, so feats_embed costs fwd + dW only -- 2x, not 3x."""
    L, d, g, F = c["L"], c["d"], c["g"], c["n_attrs"] + c["n_ctx"]
    rows = 3 * L
    feat = rows * 2 * F * g                  # forward; dW the same; no input gradient (attrs / ctx are data)
    joint = rows * 2 * (d + g) * d           # forward; dW the same; d[z;q] the same (q feeds feats_embed's dW, z the table)
    sa = c["n_blocks"] * (10 * L * d * d + 4 * L * L * d)
    ca = 2 * (2 * L) * d * d + 2 * (4 * L * d * d) + 4 * (2 * L) * L * d + 2 * (2 * L) * d  # two groups: K/V projected twice
    return dict(feat_fwd=feat, feat_dw=feat, rest=3 * (joint + sa + ca), total=2 * feat + 3 * (joint + sa + ca))


def measure_train(c, model, rank, world, device, steps, fold=False, graphed=False):
    """Secondary metric: train users/sec = fwd + bwd + gradient all-reduce + Adam, L pos + L neg
    targets per user (train.py:84-96 call shape), dropout p = 0, same C2 model and batch size per GPU.
    fold: the opt-in re-associated embedding in both directions (CARCA.fold_embedding(True, training=True)).
    graphed: forward + backward replayed from a hipGraph (engine.GraphedTrainStep, single process only)."""
    import torch
    import torch.distributed as dist

    from carca_replication_amd import engine
    from carca_replication_amd.synth import eval_batch

    L = c["L"]
    profile, pos, _ = eval_batch(c["B"], L, L, c["n_items"], c["n_attrs"], c["n_ctx"], seed=4321 + rank)
    px = profile[0]
    o_x = torch.cat([pos[0] * (px != 0), pos[0].flip(1) * (px != 0)], dim=1)
    o_a = torch.cat([pos[1], pos[1].flip(1)], dim=1)
    o_c = torch.cat([pos[2], pos[2]], dim=1)
    y_true = torch.cat([(px != 0).int(), torch.zeros_like(px)], dim=1)
    batch = tuple(t.to(device) for t in (profile[0], profile[1], profile[2], o_x, o_a, o_c, y_true))
    model.train()
    model.fold_embedding(fold, training=fold)
    from carca_replication_amd.optim import Adam

    # training.py:174's update, one launch for all tensors (lr 1e-5: with the synthetic U[0,1) attributes one 1e-3 step
    # saturates the sigmoid and `last_loss` reads -log(1e-4) from then on; the step's cost does not depend on lr)
    opt = Adam(model.parameters(), lr=1e-5, betas=(0.9, 0.98))
    if graphed:
        captured = engine.GraphedTrainStep(model, opt, batch)
        run = lambda: captured(captured.inputs)  # noqa: E731  (the graph's own input tensors: no per-step 315 MB copy, like the eager loop that reuses `batch`)
    else:
        run = lambda: engine.train_step(model, opt, batch, sharded=world > 1)  # noqa: E731
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    last = []

    def run_keep():
        last[:] = [run()]

    # (allocator pools, lazily loaded code objects and the optimizer state settle in the first steps; then the same
    # pre-heat as the headline: this measurement starts after seconds of host-side set-up with the chip idle)
    dt = timed_loop(run_keep, steps, 6, fence, preheat_calls=96 if world > 1 else None)
    loss = last[0]
    model.eval()
    model.fold_embedding(False)
    what = "fwd+bwd+Adam (+RCCL grad all-reduce when n_gpus>1), p=0, L pos + L neg targets, B=%d per GPU" % c["B"]
    if fold:
        what += ("; opt-in re-association of the linear embedding (one F->d GEMM forward, one F->d weight-gradient "
                 "product backward: ~5x fewer executed flops there, same gradients to ~1e-6)")
    if graphed:
        two = "" if fold else ("; the captured backward runs on two streams: the target rows' embedding backward beside the "
                               "encoder's backward")
        what += ("; forward + backward replayed from ONE hipGraph (engine.GraphedTrainStep" + two + "), the optimizer's launch "
                 "issued behind it: ~40 launches cost 1.3-2.3 ms of host time per step when issued eagerly")
    out = {"users_per_s": world * c["B"] * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "what": what,
           "last_loss": float(loss)}
    if not fold:
        tf = train_flops_per_user(c)
        ach = c["B"] * tf["total"] / (dt / steps) / 1e12
        # rows the two feats_embed products multiply (ids != 0 of profile + both target halves) out of 3 L per user
        kept = float((batch[0] != 0).sum() + (batch[3] != 0).sum()) / float(batch[0].numel() + batch[3].numel())
        executed = c["B"] * (kept * (tf["feat_fwd"] + tf["feat_dw"]) + tf["rest"])
        ach_x = executed / (dt / steps) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": ach_x, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach_x / PEAK_F32_MFMA_TFLOPS, "executed_gflop_per_step": executed / 1e9,
                           "rows_multiplied_share": kept,
                           "effective_tflops": ach, "effective_frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "needed_gflop_per_step": c["B"] * tf["total"] / 1e9,
                           "dominant_kernels": "gemm_wgrad_cu_kernel (dW of feats_embed, %.1f GFLOP) and gemm_rows_skc_kernel "
                                               "(its forward product, the same flops): %.0f %% of the step's needed flops"
                                               % (c["B"] * tf["feat_dw"] / 1e9, 200.0 * tf["feat_fwd"] / tf["total"]),
                           "note": "WHOLE step (fwd + bwd + Adam) against the fp32 MFMA peak on NEEDED flops: forward + "
                                   "weight gradient for the two products whose inputs are data (feats_embed: 2x its forward, "
                                   "not 3x), forward + both gradients for everything else; per-kernel durations of the same "
                                   "step: profiles/*_train_kernel_stats.csv.  Both feats_embed products leave the padding's rows "
                                   "out (ids == 0: ~47 %% of a training batch's rows with BASELINE's profile lengths): "
                                   "achieved / frac count those two products over the rows they multiply (EXECUTED flops, "
                                   "never above the peak); effective_* count every padded row (NEEDED flops: how fast the step "
                                   "delivers the reference's arithmetic)"}
    return out


def measure_scoring_scaling(c, model, device, batches=(1024, 4096), reps=30):
    """The scoring kernel (final norm + CrossAttentionBlock, eval mode) ALONE at batch sizes where several users share a
    CU -- above #CUs users the library runs it as persistent workgroups that pipeline their users (csrc/cross_stream.hip).
    Untimed side pass: `reps` back-to-back launches between two events (launch gaps included), synthetic encoder outputs /
    embedded targets of the C2 shape, profile lengths as BASELINE.md draws them (U{3..L}, left-padded) and, beside it,
    every profile full.  Fraction = SURVEY 8d's algorithmic CA flops per user x users / time / fp32 MFMA peak."""
    import torch

    from carca_replication_amd import ops

    L, N, d, H = c["L"], c["N"], c["d"], c["H"]
    dpi, _, _ = ops.padded_dims(d, H)
    cw = model.decoder.weights_struct(device, model.norm)
    ca = flops_per_user(c)["ca"]
    out = {}
    for B in batches:
        gen = torch.Generator(device=device).manual_seed(B)
        x = torch.zeros(B, L, dpi, device=device)
        x[..., :d] = torch.randn(B, L, d, device=device, generator=gen)
        o = torch.zeros(B, N, dpi, device=device)
        o[..., :d] = torch.randn(B, N, d, device=device, generator=gen)
        o_ids = torch.randint(1, 5, (B, N), device=device, dtype=torch.int32, generator=gen)
        res = {}
        ln_drawn = torch.randint(3, L + 1, (B,), device=device, generator=gen)
        for name in ("baseline_lengths", "full_profiles", "length_sorted"):
            ln = {"baseline_lengths": ln_drawn, "full_profiles": torch.full((B,), L, device=device),
                  "length_sorted": torch.sort(ln_drawn, descending=True)[0]}[name]
            p_ids = ((torch.arange(L, device=device)[None, :] >= (L - ln)[:, None]).int() * 7).contiguous()
            run = lambda: ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, False)  # noqa: E731
            for _ in range(8):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / reps
        tf = B * ca / (res["baseline_lengths"] * 1e-3) / 1e12
        tf_full = B * ca / (res["full_profiles"] * 1e-3) / 1e12
        out[B] = {"kernel": "cross_stream_kernel<96,32,3> (final norm + CrossAttentionBlock, eval mode; persistent 16-wave "
                            "workgroups, W_Q / W_K resident in LDS, one barrier per user)",
                  "users": B, "bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": tf / PEAK_F32_MFMA_TFLOPS, "avg_ms": res["baseline_lengths"],
                  "frac_full_profiles": tf_full / PEAK_F32_MFMA_TFLOPS, "avg_ms_full_profiles": res["full_profiles"],
                  "frac_length_sorted": B * ca / (res["length_sorted"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                  "avg_ms_length_sorted": res["length_sorted"],
                  "algorithmic_gflop_per_launch": B * ca / 1e9, "launches_timed": reps,
                  "timing": "side pass, untimed by the headline: back-to-back launches of the kernel alone between two "
                            "events (launch gaps included); `frac` at BASELINE.md's profile lengths U{3..50}, "
                            "`frac_full_profiles` with every profile at L = 50, `frac_length_sorted` the SAME users "
                            "as `frac` ordered by profile length (what device_data.DeviceLoader(order='length') "
                            "delivers: the persistent workgroups take users w, w + #CUs, ... and a user's cost is its "
                            "key tiles, so sorted neighbours give every workgroup the same mix)"}
    return out


def measure_split_path(c, model, profile, target, profile_cpu, target_cpu, fence, steps, warmup, world, fl):
    """Opt-in split-precision feature GEMM (SURVEY 7 hard part 1; csrc/gemm_split.hip): the SAME eval forward with
    AllEmbedding.feats_embed on the 16-bit MFMA pipe, both operands split into 16-bit parts -- 'bf16x3': three bf16 parts,
    six products; 'fp16x2': two fp16 parts (the residual scaled by 2^11), three products.  Never the headline `value`.
    Per scheme: users/s and ms per step by the headline's protocol, the kernel's own duration (events bound to its dispatch),
    the largest score difference against the default exact-fp32 path on the bench batch, the error of both paths against the
    CPU oracle run in fp64 on the first 8 users, whether every user's rank of the positive is unchanged, and a roofline entry
    against the dense 16-bit MFMA peak divided by the products a scheme spends per fp32 product."""
    import torch

    from carca_replication_amd import ops
    from oracle import carca_oracle as O

    PEAK_16BIT_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense
    with torch.no_grad():
        y32 = model(profile=profile, targets=[target]).clone()
        n8 = min(8, c["B"])
        P64 = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
        cfg = O.CarcaConfig(d=c["d"], H=c["H"], n_blocks=c["n_blocks"])
        cut = lambda t: tuple(x[:n8].double() if x.is_floating_point() else x[:n8] for x in t)  # noqa: E731
        y64 = O.carca_forward(P64, cfg, cut(profile_cpu), [cut(target_cpu)], training=False)
        err32 = float((y32[:n8].cpu().double() - y64).abs().max())
        out = {"fp32_path_max_abs_err_vs_fp64_oracle": err32, "oracle_users": n8,
               "note": "opt-in (ops.set_feature_gemm_precision / carca_set_tuning(16, mode)); the headline `value` and every "
                       "parity figure stay on the exact-fp32 MFMA path"}
        for mode, passes in (("bf16x3", 6), ("fp16x2", 3)):
            ops.set_feature_gemm_precision(mode)
            try:
                n0 = ops.split_launch_count()
                y = model(profile=profile, targets=[target]).clone()
                if ops.split_launch_count() == n0:
                    out[mode] = {"error": "the launcher did not take the split kernel for this shape"}
                    continue
                dt = timed_loop(lambda: model(profile=profile, targets=[target]), steps, warmup, fence)
                evs = [[ops.HipEvent() for _ in range(2)] for _ in range(24)]
                for e in evs:
                    ops.set_fused_events([e[0].handle, e[1].handle] + [None] * 6)
                    model(profile=profile, targets=[target])
                    ops.set_fused_events(None)
                fence()
                k_ms = sorted(e[0].elapsed_ms(e[1]) for e in evs)
                k_avg = sum(k_ms) / len(k_ms)
            finally:
                ops.set_feature_gemm_precision("fp32")
            tf = c["B"] * fl["feat"] / (k_avg * 1e-3) / 1e12
            r32, r = ops.rank_metrics(y32, 10, want_rank=True)[1], ops.rank_metrics(y, 10, want_rank=True)[1]
            out[mode] = {"passes": passes, "users_per_s": world * c["B"] * steps / dt, "ms_per_step": 1e3 * dt / steps,
                         "kernel_avg_ms": k_avg, "kernel_min_ms": k_ms[0],
                         "max_abs_diff_vs_fp32_path": float((y - y32).abs().max()),
                         "max_abs_err_vs_fp64_oracle": float((y[:n8].cpu().double() - y64).abs().max()),
                         "ranks_equal": bool(torch.equal(r, r32)),
                         "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_16BIT_TFLOPS / passes, "unit": "TFLOP/s",
                                      "frac": tf / (PEAK_16BIT_TFLOPS / passes),
                                      "note": "algorithmic fp32 flops of the product / kernel time, against the dense 16-bit "
                                              "MFMA peak (2500 TF) / %d products per fp32 product" % passes}}
    return out


def measure_full_profiles(c, model, device, fence, steps, warmup, fl):
    """The same step with EVERY profile full (L items, no padding: nothing for the compacting feature GEMM to leave out), so
    that the line carries a figure that does not depend on the batch's padding share ("What's weak" 2).  Side
    pass: the forward between two fences as the headline, then 24 launches of the feature GEMM with an event pair bound to
    its dispatch."""
    import torch

    from carca_replication_amd import ops
    from carca_replication_amd.synth import eval_batch

    prof, targ, _ = eval_batch(c["B"], c["L"], c["N"], c["n_items"], c["n_attrs"], c["n_ctx"], seed=777, min_len=c["L"])
    prof, targ = tuple(t.to(device) for t in prof), tuple(t.to(device) for t in targ)
    assert bool((prof[0] != 0).all())
    with torch.no_grad():
        dt = timed_loop(lambda: model(profile=prof, targets=[targ]), steps, warmup, fence)
        evs = [[ops.HipEvent() for _ in range(2)] for _ in range(24)]
        for e in evs:
            ops.set_fused_events([e[0].handle, e[1].handle] + [None] * 6)
            model(profile=prof, targets=[targ])
        ops.set_fused_events(None)
        fence()
    ms = sorted(e[0].elapsed_ms(e[1]) for e in evs)
    avg = sum(ms) / len(ms)
    tf = c["B"] * fl["feat"] / (avg * 1e-3) / 1e12
    return {"users_per_s": c["B"] * steps / dt, "ms_per_step": 1e3 * dt / steps, "feat_gemm_avg_ms": avg,
            "feat_gemm_tflops": tf, "feat_gemm_frac": tf / PEAK_F32_MFMA_TFLOPS}


def measure_other_configs(device, fence, steps=60, warmup=10):
    """BASELINE.json's other single-GPU configurations, eval forward, by the headline's protocol (inputs resident in HBM,
    pre-heated, `steps` timed steps): C3 (Fashion-shape: B = 512, same per-user dimensions) and C5 (Games-shape: n_attrs =
    512, 1 + 1000 candidates).  Side pass, never `value`; tools/bench_configs.py is the same measurement with per-stage
    kernel times (profiles/*_configs.json)."""
    import torch

    from carca_replication_amd import ops

    out = {}
    for name, over in (("C3", dict(B=512)), ("C5", dict(n_attrs=512, N=1001))):
        c = dict(C2, **over)
        model = build_model(c, device)
        _, _, profile, target = build_inputs(c, 4321, device)
        with torch.no_grad():
            ops.gemm_rows_log(True)
            model(profile=profile, targets=[target])
            log = ops.gemm_rows_log()
            ops.gemm_rows_log(False)
            dt = timed_loop(lambda: model(profile=profile, targets=[target]), steps, warmup, fence)
        fl = flops_per_user(c)
        out[name] = {"users_per_s": c["B"] * steps / dt, "ms_per_step": 1e3 * dt / steps,
                     "model_tflops": c["B"] * steps / dt * fl["total"] / 1e12,
                     "workload": f"B={c['B']} L={c['L']} N={c['N']} d={c['d']} g={c['g']} H={c['H']} n_attrs={c['n_attrs']}",
                     "row_kernels": [t.split(" rows=")[0] for t in log.split(";") if t]}
        del model, profile, target
        torch.cuda.empty_cache()
    return out


def measure_epoch_pipeline(c, model, device, world, fence, n_batches=256):
    """END-TO-END evaluation epoch of the assembled fast loop: users/s INCLUDING batch construction
    and the metrics.  A synthetic interaction log lives in HBM (device_data.DeviceInteractions: per user a history whose
    test-split window has BASELINE.md's profile lengths U{3..L}, ids U{1..n_items-1}, a context row per interaction);
    `train.evaluate()` runs over a `DeviceLoader` (leave-one-out windows, left padding, 100 sampled negatives and the
    context assignment of data.py:140-192 built on the device, ids + context only), the model gathers attribute rows from
    its registered table, HR@10 / NDCG@10 / loss accumulate on the device and the host reads them once per epoch.  The
    attribute table must be registered on `model` by the caller."""
    import numpy as np
    import torch

    from carca_replication_amd.device_data import DeviceInteractions, DeviceLoader
    from carca_replication_amd.train import evaluate

    rng = np.random.default_rng(4242)
    U, L, N = n_batches * c["B"], c["L"], c["N"] - 1
    lens = rng.integers(3, L + 1, size=U) + 1  # window of the test split = all but the last interaction
    hist = rng.integers(1, c["n_items"], size=int(lens.sum()), dtype=np.int64).astype(np.int32)
    hctx = rng.random((int(lens.sum()), c["n_ctx"]), dtype=np.float32)
    log = DeviceInteractions.from_arrays(lens, hist, hctx, c["n_items"], device=device)
    loader = DeviceLoader(log, "test", c["B"], L, N, shuffle=False, seed=1, chunk_batches=64)
    evaluate(model, loader, device, 10)  # (first pass: code objects of the batch builder / metric kernels, workspaces)
    fence()
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < PREHEAT_S:
        evaluate(model, loader, device, 10)
    fence()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        hr, ndcg, loss = evaluate(model, loader, device, 10)
    fence()
    dt = (time.perf_counter() - t0) / reps
    return {"users_per_s": world * U / dt, "ms_per_batch": 1e3 * dt / n_batches, "users_per_epoch": U, "epochs_timed": reps,
            "HR@10": hr, "NDCG@10": ndcg,
            "what": "train.evaluate() over device_data.DeviceLoader: batch construction on the device (one launch per "
                    "64 batches), ids-only batches + registered attribute table, HR/NDCG/loss/user sums of a batch in one "
                    "launch on the device, one host read per epoch of %d batches; same C2 model and batch size as the "
                    "headline" % n_batches}


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, script=None, python=None, env=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py argv`)
    and relay rank 0's JSON line and the exit code.  The parent never touches the GPU (no HIP call, no re-exec of a process
    that initialised one); the children read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torchrun.
    Returns (exit code, last stdout line that parses as JSON or None)."""
    import subprocess

    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    child_env = dict(os.environ if env is None else env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC: RCCL needs it)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=child_env, text=True)
    line = None
    for out in proc.stdout:  # (stderr passes through; stdout is relayed as it comes, the JSON line remembered)
        sys.stdout.write(out)
        sys.stdout.flush()
        s = out.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s)
                line = s
            except ValueError:
                pass
    return proc.wait(), line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=C2["B"])
    ap.add_argument("--no-fold", action="store_true", help="skip the folded-embedding (composed weights) measurement")
    ap.add_argument("--no-table", action="store_true", help="skip the attribute-table (ids-only batch) measurement")
    ap.add_argument("--no-split", action="store_true", help="skip the opt-in split-precision feature GEMM measurement")
    ap.add_argument("--train-steps", type=int, default=24, help="extra, untimed-by-the-headline train-step measurement")
    ap.add_argument("--no-full-profiles", action="store_true", help="skip the side pass with every profile at L = 50 (the PMC "
                    "passes of tools/profile_round.sh: their per-launch averages must cover the headline batch only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side pass over BASELINE's C3 / C5 configurations")
    ap.add_argument("--no-scoring-scaling", action="store_true", help="skip the scoring kernel's B = 1024 / 4096 side pass")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the plain invocation: launch the ranks ourselves, BEFORE anything of this process touches the GPU
        rc, line = spawn_ranks(args.gpus, sys.argv[1:])
        if rc == 0 and line is None:
            print("bench.py: the ranks exited 0 but rank 0 printed no JSON line", file=sys.stderr)
            rc = 1
        raise SystemExit(rc)

    # The box exposes every host core but the cgroup grants only a share of them: torch's default intra-op pool
    # (one thread per visible core, spinning after each CPU op) burns the quota and gets the launching thread
    # throttled for tens of ms -- seen as random 2x slow bench runs.  Size the pool to the real share.
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))
    os.environ.setdefault("MKL_NUM_THREADS", str(host_cores()))
    os.environ.setdefault("OPENBLAS_NUM_THREADS", str(host_cores()))
    os.environ.setdefault("KMP_BLOCKTIME", "0")      # worker threads sleep, not spin, after a parallel region
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    import torch
    import torch.distributed as dist

    torch.set_num_threads(host_cores())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test hooks (a 1-GPU box cannot run RCCL across ranks): CARCA_BENCH_DEVICE pins every rank to one device and
    # CARCA_BENCH_BACKEND=gloo swaps the collective backend, so the N > 1 control flow can be rehearsed there
    if os.environ.get("CARCA_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["CARCA_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = {"backend": None, "world_size": 1}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CARCA_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # what the process group itself reports (not the environment): the line's n_gpus must be the ranks that ran
        collective = {"backend": "rccl (torch 'nccl')" if backend == "nccl" else backend, "world_size": dist.get_world_size()}
        assert collective["world_size"] == args.gpus

    from carca_replication_amd import ops

    c = dict(C2, B=args.batch)
    model = build_model(c, device)
    profile_cpu, target_cpu, profile, target = build_inputs(c, 1234 + rank, device)
    fl = flops_per_user(c)


    # hipEvent_t pairs handed to carca_forward are BOUND to the dispatch packets of its kernels (hipExtLaunchKernel: the
    # kernel's own start / end on the launch stream, nothing extra queued -- an event RECORD would be a barrier packet of
    # ~6 us between two kernels).  Even so a pair costs ~5 us per kernel and step (tools/event_cost.py: 0.6185 ms per step
    # without events, 0.6234 with the feature GEMM's pair, 0.6382 with four pairs), so the timed region carries ONLY the
    # dominant kernel's pair, on every EVENT_EVERY-th step; the three smaller kernels (scoring, first SelfAttentionBlock,
    # joint GEMM) are timed in a pass of their own right behind the timed region (SIDE_STEPS steps, untimed).
    EVENT_EVERY, SIDE_STEPS = 4, 48
    pool = [[ops.HipEvent() for _ in range(8)] for _ in range((args.steps + EVENT_EVERY - 1) // EVENT_EVERY + SIDE_STEPS)]
    used, side = [], []

    def step(record, every_kernel=False):
        if record:
            evs = pool[len(used) + len(side)]
            (side if every_kernel else used).append(evs)
            hs = [e.handle for e in evs]
            ops.set_fused_events(hs if every_kernel else hs[:2] + [None] * 6)
        y = model(profile=profile, targets=[target])
        ops.set_fused_events(None)
        return y

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        step(False)  # (first call: lazy code-object loads, workspace allocation)
        fence()
        # The interpreter's cyclic collector scans every container object of the process when its oldest generation
        # fills up: 80-120 ms here (torch + the synthetic inputs), i.e. one such pause inside a 24-step train measurement
        # reads as +4 ms per step.  Everything built so far is long-lived: park it in the permanent generation.
        gc.collect()
        gc.freeze()
        time.sleep(0.3)  # let the CPU pools used while building the inputs go idle (cgroup CPU quota, see above)
        # The chip idles through set-up and that pause, and its clock takes tens of milliseconds of load to come back:
        # the feature GEMM reads 0.62 ms over the first ten launches after an idle period, 0.57 over fifty, 0.55 over two
        # hundred (same binary, same box).  Sustained throughput is the metric, so the W warm-up steps are preceded by
        # an untimed pre-heat of the same step until PREHEAT_S of wall time have passed (reported as `preheat_ms`).
        t_heat, n_heat = time.perf_counter(), 0
        while time.perf_counter() - t_heat < PREHEAT_S:
            for _ in range(16):
                step(False)
            n_heat += 16
            torch.cuda.synchronize()
        for _ in range(args.warmup):
            step(False)
        fence()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        host_done = []
        t0 = time.perf_counter()
        g0.record()
        for i in range(args.steps):
            y = step(i % EVENT_EVERY == 0)
            host_done.append(time.perf_counter())
        g1.record()
        fence()
        elapsed = time.perf_counter() - t0
        gpu_span_ms = g0.elapsed_time(g1)
        host_issue_ms = 1e3 * (host_done[-1] - t0)
        for _ in range(SIDE_STEPS):  # (outside the timed region: every kernel's pair)
            step(True, every_kernel=True)
        fence()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
    per_rank_s = [elapsed]
    if world > 1:
        # every rank's own time for its K steps (for the line's per-rank users/s), then the max over ranks
        mine = torch.zeros(world, dtype=torch.float64, device=device)
        mine[rank] = elapsed
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        per_rank_s = [float(v) for v in mine.tolist()]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    split_info = None
    if not args.no_split and rank == 0:
        split_info = measure_split_path(c, model, profile, target, profile_cpu, target_cpu, fence if world == 1 else
                                        torch.cuda.synchronize, args.steps, args.warmup, 1, fl)

    full_info = None
    if rank == 0 and world == 1 and not args.no_full_profiles:
        full_info = measure_full_profiles(c, model, device, fence, args.steps, args.warmup, fl)

    # extension path: attribute table resident in HBM, ids-only batches, gather fused into the GEMM
    table_info = epoch_info = None
    if not args.no_table:
        import numpy as np

        tab = torch.from_numpy(np.random.default_rng(99).random((c["n_items"], c["n_attrs"]), dtype=np.float32))
        tab[0] = 0
        model.embeds.register_attr_table(tab.to(device))
        pf, tg = (profile[0], None, profile[2]), (target[0], None, target[2])
        with torch.no_grad():
            model(profile=pf, targets=[tg])
            torch.cuda.synchronize()
            gc.collect()
            gc.freeze()
            time.sleep(0.3)  # (the CPU pools that drew and uploaded the table go idle, as before the headline)
            dt = timed_loop(lambda: model(profile=pf, targets=[tg]), args.steps, args.warmup, fence)
            epoch_info = measure_epoch_pipeline(c, model, device, world, fence) if world == 1 else None
        model.embeds.register_attr_table(None)
        table_info = {"users_per_s": world * c["B"] * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                      "what": "same model and ids; attrs gathered by item id from a device-resident [n_items, n_attrs] "
                              "table inside the feature GEMM (register_attr_table), no dense attrs batch tensor"}

    # opt-in inference shortcut: AllEmbedding's two Linear layers composed into one (fewer EXECUTED flops, same
    # algorithmic work; ~1e-6 relative re-association).  Reported beside the headline, never as `value`.
    fold_info = None
    if not args.no_fold:
        model.fold_embedding(True)
        with torch.no_grad():
            dt = timed_loop(lambda: model(profile=profile, targets=[target]), args.steps, args.warmup, fence)
            yf = model(profile=profile, targets=[target])
            model.fold_embedding(False)
            y0 = model(profile=profile, targets=[target])
        fold_info = {"users_per_s": world * c["B"] * args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                     "max_abs_diff_vs_unfolded": float((yf - y0).abs().max()),
                     "what": "CARCA.fold_embedding(True): e = z W_jz^T + [a;c] (W_jq W_f)^T + const, one F->d GEMM instead of "
                             "F->g->d; executed flops per user 5x lower in the embedding, algorithmic flops unchanged"}

    scoring_scaling = measure_scoring_scaling(c, model, device) if (rank == 0 and not args.no_scoring_scaling) else {}

    train_info = train_graph_info = train_fold_info = train_fold_graph_info = None
    if args.train_steps > 0:
        train_info = measure_train(c, model, rank, world, device, args.train_steps)
        if world == 1:  # (the same step replayed from a hipGraph: wins where the host, not the GPU, paces the eager loop)
            train_graph_info = measure_train(c, model, rank, world, device, args.train_steps, graphed=True)
        if not args.no_fold:
            train_fold_info = measure_train(c, model, rank, world, device, args.train_steps, fold=True)
            if world == 1:
                train_fold_graph_info = measure_train(c, model, rank, world, device, args.train_steps, fold=True,
                                                      graphed=True)

    feat_ms = sorted(e[0].elapsed_ms(e[1]) for e in used)
    ca_ms = sorted(e[2].elapsed_ms(e[3]) for e in side)
    sa_ms = sorted(e[4].elapsed_ms(e[5]) for e in side)
    joint_ms = sorted(e[6].elapsed_ms(e[7]) for e in side)
    feat_avg = sum(feat_ms) / len(feat_ms)
    ca_avg = sum(ca_ms) / len(ca_ms)

    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the figure
    # comes from the NEWEST committed PMC passes (tools/profile_round.sh -> profiles/*_feat_gemm_traffic.json) and is
    # only reported while the kernel's source is the one those passes measured (sha256 of csrc/gemm.hip in the file);
    # after any change to the kernel it reads null until the passes are re-run.
    traffic, traffic_note = None, "no committed PMC pass"
    import glob
    import hashlib

    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_feat_gemm_traffic.json")))
    if c["B"] == C2["B"] and tfiles:
        with open(tfiles[-1]) as fh:
            tj = json.load(fh)
        with open(os.path.join(ROOT, "carca_replication_amd", "csrc", "gemm.hip"), "rb") as fh:
            src_hash = hashlib.sha256(fh.read()).hexdigest()
        if tj.get("gemm_hip_sha256") == src_hash:
            traffic = tj["hbm_bytes_per_launch"]
            traffic_note = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                            "separate passes, profiles/%s (same csrc/gemm.hip); algorithmic bytes 359 MB"
                            % os.path.basename(tfiles[-1]))
        else:
            traffic_note = ("profiles/%s was measured on another version of csrc/gemm.hip: stale, not reported"
                            % os.path.basename(tfiles[-1]))

    if rank == 0:
        users = world * c["B"] * args.steps
        value = users / elapsed
        feat_tflops = c["B"] * fl["feat"] / (feat_avg * 1e-3) / 1e12
        n_rows = profile[0].numel() + target[0].numel()
        kept_share = float((profile[0] != 0).sum() + (target[0] != 0).sum()) / n_rows  # rows the feature GEMM multiplies
        ca_tflops = c["B"] * fl["ca"] / (ca_avg * 1e-3) / 1e12
        sa_avg, joint_avg = sum(sa_ms) / len(sa_ms), sum(joint_ms) / len(joint_ms)
        sa_tflops = c["B"] * (fl["sa"] / c["n_blocks"]) / (sa_avg * 1e-3) / 1e12
        joint_flops = c["B"] * (c["L"] + c["N"]) * 2 * (c["d"] + c["g"]) * c["d"]
        out = {
            "metric": "scored users/sec (1+100 candidates), CARCA eval forward",
            "value": value, "unit": "users/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2 synthetic Beauty-shape eval: per GPU B=%d users x (L=50 profile + 1+100 candidates), "
                                   "d=90 g=450 H=3 2 SA blocks + cross-attention decoder, n_attrs=4096 n_ctx=6 "
                                   "n_items=12102, random-init weights" % c["B"],
                       "users_per_gpu_per_step": c["B"], "parallelism": f"users sharded x{world}, no data-path collective",
                       "collective": collective,
                       "per_rank_users_per_s": [c["B"] * args.steps / t for t in per_rank_s]},
            "model_tflops": value * fl["total"] / 1e12,
            "preheat_ms": 1e3 * PREHEAT_S, "preheat_steps": n_heat,
            "timeline": {"gpu_span_ms": gpu_span_ms, "host_issue_ms": host_issue_ms,
                         "note": "GPU time between the first and last launch of the timed region, and host time to "
                                 "issue them; wall >> gpu_span means the host, not the GPU, set the pace"},
            "roofline": {"kernel": "gemm_rows_skc_kernel (feature GEMM launch, AllEmbedding feats_embed, carca.py:86: 384-row x 96-column "
                                   "tiles over the rows whose id is not 0 -- the padding's rows are masked two lines later, "
                                   "carca.py:92-94 --, one workgroup per CU; teams of workgroups share the K steps of all row "
                                   "blocks as stretches and hand partial tiles to their neighbours)",
                         "bound": "mfma", "achieved": feat_tflops * kept_share, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": feat_tflops * kept_share / PEAK_F32_MFMA_TFLOPS,
                         "rows_multiplied_share": kept_share,
                         "effective_tflops": feat_tflops,
                         "effective_frac": feat_tflops / PEAK_F32_MFMA_TFLOPS,
                         "frac_full_profiles": None if full_info is None else full_info["feat_gemm_frac"],
                         "avg_ms_full_profiles": None if full_info is None else full_info["feat_gemm_avg_ms"],
                         "value_full_profiles": None if full_info is None else full_info["users_per_s"],
                         "ms_per_step_full_profiles": None if full_info is None else full_info["ms_per_step"],
                         "flops_note": "achieved / frac: the flops the kernel EXECUTES (SURVEY 8d's 2 F g per row x the rows "
                                       "with id != 0: BASELINE's profile lengths U{3..50} leave %.1f %% of this batch's rows "
                                       "out, exactly -- carca.py:94 zeroes them) / kernel time: the share of the MFMA peak the "
                                       "kernel really reaches, never above 1; effective_*: SURVEY 8d's ALGORITHMIC flops (every "
                                       "slot of the padded id matrices) / the same time; *_full_profiles: the same step with "
                                       "every profile at L = 50 (nothing left out: executed = algorithmic), a side pass"
                                       % (100 * (1 - kept_share)),
                         "traffic": traffic,
                         "traffic_note": traffic_note,
                         "avg_ms": feat_avg, "min_ms": feat_ms[0], "algorithmic_gflop_per_launch": c["B"] * fl["feat"] / 1e9,
                         "launches_timed": len(feat_ms),
                         "timing": "HIP events bound to the kernel's own dispatch, every %d-th step of the timed region "
                                   "(a pair costs ~5 us per step: tools/event_cost.py)" % EVENT_EVERY},
            "roofline_cross_score": {"kernel": "cross_fold_kernel<96,32,3,16,staged> (final norm + CrossAttentionBlock, eval mode)",
                                     "bound": "mfma", "achieved": ca_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                                     "unit": "TFLOP/s", "frac": ca_tflops / PEAK_F32_MFMA_TFLOPS, "avg_ms": ca_avg,
                                     "min_ms": ca_ms[0], "algorithmic_gflop_per_launch": c["B"] * fl["ca"] / 1e9,
                                     "launches_timed": len(ca_ms),
                                     "note": ("the kernel's own dispatch, in a pass of %d steps right behind the timed region (as "
                                              "roofline_sa_block and roofline_joint_gemm: their event pairs would cost the timed "
                                              "region ~5 us per kernel and step).  " % SIDE_STEPS) +
                                             "Algorithmic flops = SURVEY 8d's CA "
                                             "per user (2 N d^2 + 4 L d^2 + 4 N L d + 2 N d) x users; the kernel executes "
                                             "fewer: decoder.ffn is folded into the value projection (no V, no P.V) and "
                                             "leading pad slots of the left-padded profiles are not projected or scored "
                                             "(exact: their weights are 0).  At B = 128 one launch is one latency chain per "
                                             "workgroup (two per user); B-scaling in profiles/"},
            "roofline_sa_block": {"kernel": "sa_eval_kernel<96,32,3> (one SelfAttentionBlock in eval mode, first of %d; leading pad slots re-based away)" % c["n_blocks"],
                                  "bound": "mfma", "achieved": sa_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": sa_tflops / PEAK_F32_MFMA_TFLOPS, "avg_ms": sa_avg, "min_ms": sa_ms[0],
                                  "launches_timed": len(sa_ms), "timing": "pass behind the timed region, as roofline_cross_score",
                                  "algorithmic_gflop_per_launch": c["B"] * fl["sa"] / c["n_blocks"] / 1e9},
            "roofline_joint_gemm": {"kernel": "gemm_rows_n96_kernel, one 80x96 block per CU (AllEmbedding.joint_embed, carca.py:89)", "bound": "mfma",
                                    "achieved": joint_flops / (joint_avg * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                    "unit": "TFLOP/s", "frac": joint_flops / (joint_avg * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                    "avg_ms": joint_avg, "min_ms": joint_ms[0], "launches_timed": len(joint_ms),
                                    "timing": "pass behind the timed region, as roofline_cross_score",
                                    "algorithmic_gflop_per_launch": joint_flops / 1e9},
        }
        for bsz, info in scoring_scaling.items():
            out["roofline_cross_score_b%d" % bsz] = info
        if train_info is not None:
            out["train"] = train_info
        if train_graph_info is not None:
            out["train_graphed"] = train_graph_info
        if train_fold_info is not None:
            out["train_folded_embedding"] = train_fold_info
        if train_fold_graph_info is not None:
            out["train_folded_embedding_graphed"] = train_fold_graph_info
        if table_info is not None:
            out["attr_table_path"] = table_info
        if epoch_info is not None:
            epoch_info["vs_headline"] = epoch_info["users_per_s"] / value
            out["epoch_pipeline"] = epoch_info
        if fold_info is not None:
            out["folded_embedding_path"] = fold_info
        if split_info is not None:
            out["split_bf16_path"] = split_info
        if world == 1 and not args.no_configs:
            out["other_configs"] = measure_other_configs(device, fence)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c, model, profile_cpu, target_cpu)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
