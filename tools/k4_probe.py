"""Cross-attention scoring kernel (K4): folded inference kernel against the V-materialising one -- agreement, timing at
several batch sizes (full-length profiles and BASELINE's lengths U{3..L}, left-padded), phase stamps.
  python tools/k4_probe.py            (env: BS=128,1024,4096  N=101  STAMPS=1)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from carca_replication_amd import _lib, ops  # noqa: E402
from tests.model_util import build_model  # noqa: E402

L, N, d, g, H = 50, int(os.environ.get("N", "101")), int(os.environ.get("D", "90")), 450, int(os.environ.get("H", "3"))
PEAK = 157.3e12
CA = 2 * N * d * d + 4 * L * d * d + 4 * N * L * d + 2 * N * d
torch.manual_seed(0)
model = build_model(dict(d=d, H=H, n_blocks=2), 500, g, 6, 64, L).eval().cuda()
for p in model.parameters():  # biases are zero-initialised: make every term of the arithmetic visible
    if p.dim() == 1:
        p.data.add_(0.1 * torch.randn_like(p))
dpi, _, _ = ops.padded_dims(d, H)
cw = model.decoder.weights_struct(torch.device("cuda"), model.norm)
lib = _lib.load()


def inputs(B, lengths):
    gen = torch.Generator(device="cuda").manual_seed(B)
    x = torch.zeros(B, L, dpi, device="cuda")
    x[..., :d] = torch.randn(B, L, d, device="cuda", generator=gen)
    o = torch.zeros(B, N, dpi, device="cuda")
    o[..., :d] = torch.randn(B, N, d, device="cuda", generator=gen)
    if lengths == "full":
        ln = torch.full((B,), L, device="cuda")
    else:
        ln = torch.randint(3, L + 1, (B,), device="cuda", generator=gen)
    p_ids = (torch.arange(L, device="cuda")[None, :] >= (L - ln)[:, None]).int() * 7
    if lengths == "holes":  # pads inside the profile too, and one all-pad user
        p_ids = p_ids * (torch.rand(B, L, device="cuda", generator=gen) > 0.2).int()
        p_ids[0] = 0
    o_ids = torch.randint(1, 5, (B, N), device="cuda", dtype=torch.int32, generator=gen)
    o_ids[:, -1] = 0
    return x, p_ids.int().contiguous(), o, o_ids


def run(x, p_ids, o, o_ids, training, fold):
    lib.carca_set_tuning(6, 0 if fold else 1)
    ys, _ = ops.cross_score_fwd(x, p_ids, [(o, o_ids)], cw, d, H, True, training)
    lib.carca_set_tuning(6, 0)
    return ys[0]


def timed(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print("agreement (max |folded - materialised|):")
for lengths in ("full", "uniform", "holes"):
    for training in (False, True):
        if training and N != L:
            continue
        a = inputs(37, lengths)
        print(f"  {lengths:8s} training={training}: {float((run(*a, training, True) - run(*a, training, False)).abs().max()):.3e}")
for B in [int(b) for b in os.environ.get("BS", "128,256,512,1024,4096").split(",")]:
    for lengths in ("full", "uniform"):
        a = inputs(B, lengths)
        t_old = timed(lambda: run(*a, False, False))
        t_new = timed(lambda: run(*a, False, True))
        extra = ""
        if B > 256:  # one 16-wave workgroup per user with LDS staging instead of two 8-wave ones per CU (tuning key 1 = 1)
            lib.carca_set_tuning(1, 1)
            t_16 = timed(lambda: run(*a, False, True))
            lib.carca_set_tuning(1, 0)
            extra = f"   one staged 16-wave workgroup per CU {t_16 * 1e6:8.1f} us"
        if B <= 256:  # the 16-wave workgroups without LDS staging (tuning key 7)
            lib.carca_set_tuning(7, 1)
            t_ns = timed(lambda: run(*a, False, True))
            lib.carca_set_tuning(7, 0)
            extra = f"   folded, unstaged {t_ns * 1e6:8.1f} us"
        print(f"B={B:5d} {lengths:8s} materialised {t_old * 1e6:8.1f} us {B * CA / t_old / PEAK * 100:5.1f} %   "
              f"folded {t_new * 1e6:8.1f} us {B * CA / t_new / PEAK * 100:5.1f} %" + extra, flush=True)
NAMES = ["start", "ids in", "A barrier passed", "B done", "end", "wave 0: B work done", "LN done", "job: operands", "job: Q proj", "job: scores",
         "job: softmax", "C barrier passed", "at A barrier"]
if os.environ.get("STAMPS", "1") != "0":
    for dbg in [int(x) for x in os.environ.get("DBG", "0,2,4,16,22").split(",")]:
        lib.carca_set_tuning(5, dbg)
        cases = ((128, "full"), (1024, "full")) if dbg else ((128, "full"), (128, "uniform"), (1024, "full"))
        if os.environ.get("STAMP_BS"):
            cases = tuple((int(b), "full") for b in os.environ["STAMP_BS"].split(","))
        for B, lengths in cases:
            a = inputs(B, lengths)
            for _ in range(5):
                run(*a, False, True)
            t = timed(lambda: run(*a, False, True))
            nwg = 2 * B
            buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
            lib.carca_set_debug_buffer(buf.data_ptr())
            run(*a, False, True)
            torch.cuda.synchronize()
            lib.carca_set_debug_buffer(None)
            st = buf.view(nwg, 16).cpu().double()
            st = st[st[:, 0] > 0]
            rel = st - st[:, :1]
            order = [1, 6, 12, 2, 5, 3, 7, 8, 9, 10, 11, 4]
            print(f"dbg={dbg} B={B} {lengths}: {t * 1e6:.1f} us; cycles since kernel start, median over {len(st)} workgroups (max):")
            print("   " + "  ".join(f"[{NAMES[i]}] {rel[:, i][rel[:, i] > 0].median() if (rel[:, i] > 0).any() else 0:.0f}"
                                    f" ({rel[:, i].max():.0f})" for i in order), flush=True)
    lib.carca_set_tuning(5, 0)
if os.environ.get("WAVES"):  # per-wave clocks: start of every wave (dbg 32) and its arrival at the A barrier (dbg 64)
    a = inputs(128, "full")
    per = {}
    for dbg in (32,):
        lib.carca_set_tuning(5, dbg)
        for _ in range(5):
            run(*a, False, True)
        buf = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
        lib.carca_set_debug_buffer(buf.data_ptr())
        run(*a, False, True)
        torch.cuda.synchronize()
        lib.carca_set_debug_buffer(None)
        per[dbg] = buf.view(256, 16).cpu().double()
    lib.carca_set_tuning(5, 0)
    st = per[32]
    print("wave start - earliest wave start of the workgroup, median (max) over workgroups:")
    rel = st - st.min(dim=1, keepdim=True).values
    print("   " + "  ".join(f"w{w}: {rel[:, w].median():.0f} ({rel[:, w].max():.0f})" for w in range(16)))
    print(f"   workgroup start - earliest workgroup start: median {float((st.min(1).values - st.min()).median()):.0f}, max {float((st.min(1).values - st.min()).max()):.0f}")
    for extra, what in ((0, ""), (8, ", no LayerNorm arithmetic"), (6, ", no W_Q / target-tile DMA issued"), (14, ", neither"), (32, ": END OF PHASE B work instead"), (38, ": END OF PHASE B work, no DMA issued")):
        lib.carca_set_tuning(5, 64 | extra)
        for _ in range(5):
            run(*a, False, True)
        buf = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
        lib.carca_set_debug_buffer(buf.data_ptr())
        run(*a, False, True)
        torch.cuda.synchronize()
        lib.carca_set_debug_buffer(None)
        ar = buf.view(256, 16).cpu().double()
        print(f"(separate launch{what}) arrival at the A barrier - earliest arrival of the workgroup:")
        rel = ar - ar.min(dim=1, keepdim=True).values
        print("   " + "  ".join(f"w{w}: {rel[:, w].median():.0f} ({rel[:, w].max():.0f})" for w in range(16)))
        print(f"   last wave of a workgroup: median {float(rel.max(dim=1).values.median()):.0f}; which wave is last: "
              + str(torch.bincount(rel.argmax(dim=1), minlength=16).tolist()))
    lib.carca_set_tuning(5, 0)
