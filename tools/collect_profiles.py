"""Copy one round's rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked):
  python tools/collect_profiles.py r02_e [--attn r02_d_attn] [--train]
Keeps what profiles/README.md lists: kernel stats, per-grid split, the three PMC passes filtered to the feature-GEMM
kernel, the traffic JSON, both bench lines; with --attn the K2/K4 PMC summary and their kernel stats; with --train the
train-step kernel stats of tools/train_trace.sh."""
import argparse
import csv
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--attn")
ap.add_argument("--train", action="store_true")
a = ap.parse_args()
src, dst = os.path.join(ROOT, "gpurun_out", a.tag), os.path.join(ROOT, "profiles")


def one(base, pat):
    f = glob.glob(os.path.join(base, pat), recursive=True)
    return f[0] if f else None


def put(path, name):
    if path and os.path.exists(path):
        shutil.copyfile(path, os.path.join(dst, f"{a.tag}_{name}"))
        print("profiles/%s_%s" % (a.tag, name))


def filtered(path, name, key="gemm_rows_sk"):
    if not path:
        return
    rows = list(csv.reader(open(path)))
    with open(os.path.join(dst, f"{a.tag}_{name}"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(rows[0])
        w.writerows(r for r in rows[1:] if any(key in c for c in r))
    print("profiles/%s_%s" % (a.tag, name))


if os.path.isdir(src):
    put(one(src, "stats/**/*kernel_stats.csv"), "kernel_stats.csv")
    put(os.path.join(src, "kernels_by_grid.csv"), "kernels_by_grid.csv")
    put(os.path.join(src, "feat_gemm_traffic.json"), "feat_gemm_traffic.json")
    put(os.path.join(src, "bench.json"), "bench.json")
    put(os.path.join(src, "bench_under_rocprof.json"), "bench_under_rocprof.json")
    filtered(one(src, "fetch/**/*counter_collection.csv"), "pmc_fetch_feat_gemm.csv")
    filtered(one(src, "write/**/*counter_collection.csv"), "pmc_write_feat_gemm.csv")
    filtered(one(src, "sq/**/*counter_collection.csv"), "pmc_sq_feat_gemm.csv")
if a.attn:
    at = os.path.join(ROOT, "gpurun_out", a.attn)
    put(os.path.join(at, "attn_pmc_summary.csv"), "attn_pmc_summary.csv")
    put(one(at, "stats/**/*kernel_stats.csv"), "attn_kernel_stats.csv")
if a.train:
    put(one(os.path.join(ROOT, "gpurun_out", "train_trace"), "stats/**/*kernel_stats.csv"), "train_kernel_stats.csv")
