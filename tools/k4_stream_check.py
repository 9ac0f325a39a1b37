"""cross_stream_kernel (persistent, user-pipelined scoring kernel for B > #CUs) against cross_fold_kernel (tuning key 7 = 2)
and the V-materialising kernel (key 6 = 1): agreement over shapes that exercise every path (several rounds of target
tiles, several groups, one tile, pads inside profiles, all-pad users, every (d, H) instantiation), then timing.
  python tools/k4_stream_check.py      (env: BS=1024,4096  TIMING=1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

PEAK = 157.3e12
lib = _lib.load()


def make(d, H, L):
    torch.manual_seed(0)
    model = build_model(dict(d=d, H=H, n_blocks=1), 500, 64, 6, 64, L).eval().cuda()
    for p in model.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    dpi, _, _ = ops.padded_dims(d, H)
    return model, dpi, model.decoder.weights_struct(torch.device("cuda"), model.norm)


def inputs(B, L, Ns, d, dpi, lengths, seed=0):
    gen = torch.Generator(device="cuda").manual_seed(B + seed)
    x = torch.zeros(B, L, dpi, device="cuda")
    x[..., :d] = torch.randn(B, L, d, device="cuda", generator=gen)
    if lengths == "full":
        ln = torch.full((B,), L, device="cuda")
    else:
        ln = torch.randint(min(3, L), L + 1, (B,), device="cuda", generator=gen)
    p_ids = (torch.arange(L, device="cuda")[None, :] >= (L - ln)[:, None]).int() * 7
    if lengths == "holes":
        p_ids = p_ids * (torch.rand(B, L, device="cuda", generator=gen) > 0.2).int()
        p_ids[0] = 0
        p_ids[B // 2] = 0
    groups = []
    for N in Ns:
        o = torch.zeros(B, N, dpi, device="cuda")
        o[..., :d] = torch.randn(B, N, d, device="cuda", generator=gen)
        o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32, generator=gen)
        o_ids[:, -1] = 0
        groups.append((o, o_ids))
    return x, p_ids.int().contiguous(), groups


def run(x, p_ids, groups, cw, d, H, mode):
    lib.carca_set_tuning(7, {"stream": 3, "ticket": 3, "fold": 2, "mat": 2}[mode])
    lib.carca_set_tuning(6, 1 if mode == "mat" else 0)
    lib.carca_set_tuning(14, 1 if mode == "ticket" else 0)
    try:
        ys, _ = ops.cross_score_fwd(x, p_ids, groups, cw, d, H, True, False)
    finally:
        lib.carca_set_tuning(7, 0)
        lib.carca_set_tuning(6, 0)
        lib.carca_set_tuning(14, 0)
    return [y.clone() for y in ys]


def timed_mode(a, cw, d, H, mode, reps=40):
    """back-to-back launches between two events, the variant switches set once around the whole loop"""
    lib.carca_set_tuning(7, {"stream": 3, "ticket": 3, "fold": 2}[mode])
    lib.carca_set_tuning(14, 1 if mode == "ticket" else 0)
    try:
        t = timed(lambda: ops.cross_score_fwd(*a, cw, d, H, True, False), reps)
    finally:
        lib.carca_set_tuning(7, 0)
        lib.carca_set_tuning(14, 0)
    return t


def timed(fn, reps=30):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


worst = 0.0
bad = []
CASES = [
    # d, H, L, Ns, B, lengths
    (90, 3, 50, [101], 300, "uniform"), (90, 3, 50, [101], 300, "full"), (90, 3, 50, [101], 300, "holes"),
    (90, 3, 50, [101], 37, "holes"), (90, 3, 50, [101], 1, "uniform"), (90, 3, 50, [101], 777, "uniform"),
    (90, 3, 50, [1], 300, "holes"), (90, 3, 50, [16], 300, "uniform"), (90, 3, 50, [17], 300, "uniform"),
    (90, 3, 50, [300], 290, "holes"), (90, 3, 50, [1001], 260, "uniform"), (90, 3, 50, [50, 50], 300, "holes"),
    (90, 3, 50, [5, 130, 33], 270, "holes"), (90, 2, 50, [101], 300, "holes"), (90, 1, 50, [101], 300, "holes"),
    (64, 2, 50, [101], 300, "holes"), (64, 4, 50, [101], 300, "holes"), (64, 1, 64, [101], 300, "holes"),
    (48, 4, 17, [40], 300, "holes"), (40, 4, 5, [7], 300, "holes"), (96, 3, 64, [129], 300, "holes"),
    (128, 4, 50, [101], 300, "holes"),  # (d > 96: no stream instantiation -> the call falls through to cross_fold_kernel)
]
models = {}
for d, H, L, Ns, B, lengths in CASES:
    key = (d, H, L)
    if key not in models:
        models[key] = make(d, H, L)
    model, dpi, cw = models[key]
    a = inputs(B, L, Ns, d, dpi, lengths)
    ys, yf, ym, yt = (run(*a, cw, d, H, m) for m in ("stream", "fold", "mat", "ticket"))
    same = all(torch.equal(s, t) for s, t in zip(ys, yt))  # dynamic and static dealing must give the same bits
    e_sf = max(float((s - f).abs().max()) for s, f in zip(ys, yf))
    e_sm = max(float((s - m).abs().max()) for s, m in zip(ys, ym))
    fin = all(bool(torch.isfinite(s).all()) for s in ys)
    worst = max(worst, e_sf, e_sm)
    flag = "" if (e_sf < 2e-6 and e_sm < 2e-6 and fin and same) else "   <-- BAD" + ("" if same else " (static != ticket)")
    if flag:
        bad.append((d, H, L, Ns, B, lengths))
    print(f"d={d:3d} H={H} L={L:2d} N={Ns} B={B:4d} {lengths:8s}: |stream-fold| {e_sf:.2e}  |stream-mat| {e_sm:.2e}{flag}", flush=True)
print("worst:", worst, "BAD cases:", bad)
if os.environ.get("TIMING", "1") != "0":
    d, H, L, N = 90, 3, 50, 101
    model, dpi, cw = models[(d, H, L)]
    CA = 2 * N * d * d + 4 * L * d * d + 4 * N * L * d + 2 * N * d
    for B in [int(b) for b in os.environ.get("BS", "257,512,1024,2048,4096,8192").split(",")]:
        for lengths in ("full", "uniform"):
            a = inputs(B, L, [N], d, dpi, lengths)
            t_f = timed_mode(a, cw, d, H, "fold")
            t_s = timed_mode(a, cw, d, H, "stream")
            t_t = timed_mode(a, cw, d, H, "ticket")
            print(f"B={B:5d} {lengths:8s} fold {t_f * 1e6:8.1f} us {B * CA / t_f / PEAK * 100:5.1f} %   "
                  f"stream {t_s * 1e6:8.1f} us {B * CA / t_s / PEAK * 100:5.1f} %   "
                  f"stream, ticket jobs {t_t * 1e6:8.1f} us {B * CA / t_t / PEAK * 100:5.1f} %", flush=True)
if os.environ.get("STAMPS"):  # per-wave clocks of one step (tuning key 5 = step index) of every workgroup
    d, H, L, N = 90, 3, 50, 101
    model, dpi, cw = models[(d, H, L)]
    step = int(os.environ.get("STEP", "2"))
    for B, lengths in [(int(b), l) for b in os.environ.get("STAMP_BS", "4096").split(",") for l in ("full", "uniform")]:
        a = inputs(B, L, [N], d, dpi, lengths)
        for _ in range(5):
            run(*a, cw, d, H, "stream")
        nwg = min(256, B * (2 if 2 * B <= 256 else 1))
        buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
        lib.carca_set_debug_buffer(buf.data_ptr())
        lib.carca_set_tuning(15, step)
        run(*a, cw, d, H, "stream")
        torch.cuda.synchronize()
        lib.carca_set_tuning(15, 0)
        lib.carca_set_debug_buffer(None)
        st = buf.view(256, 64)[:nwg].cpu().double()
        opened, arrive, mid = st[:, 0:16], st[:, 16:32], st[:, 32:48]
        t0 = opened.min(dim=1, keepdim=True).values
        med = lambda t: [int(x) for x in t.median(dim=0).values.tolist()]  # noqa: E731
        print(f"B={B} {lengths} step {step}: cycles since the step opened (median over workgroups), waves 0..11 = C, 12..15 = B")
        print("   opened        ", med(opened - t0))
        print("   first job/fin ", med(mid - t0))
        print("   at the barrier", med(arrive - t0))
        print("   step length (last arrival - first open): median %.0f  max %.0f" % (
            float((arrive.max(dim=1).values - t0[:, 0]).median()), float((arrive.max(dim=1).values - t0[:, 0]).max())))
        print("   kernel span: median %.0f cycles; step opened %.0f cycles after the kernel's first instruction" % (
            float((st[:, 52] - st[:, 48]).median()), float((t0[:, 0] - st[:, 48]).median())))
