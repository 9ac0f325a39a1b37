#!/bin/bash
# HBM-traffic PMC passes for the kernels profiles/ had no counters for: tools/pmc_extra.sh <tag>  (on the GPU box via gpurun)
#   table: feature GEMM with the attribute rows gathered by id inside it;  train: the weight-gradient kernel of feats_embed;
#   knn: the KNN baseline's scoring kernel.  One kernel-trace pass (durations) + one pass per counter, as the guide asks.
set -u
TAG=${1:-r03_x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${TAG}_pmc_extra
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for MODE in table train knn; do
  W="python3 $ROOT/tools/pmc_extra_workload.py $MODE"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$MODE/stats" -o s -- $W > /dev/null 2> "$OUT/$MODE.stats.log" || exit 1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/$MODE/fetch" -o f -- $W > /dev/null 2> "$OUT/$MODE.fetch.log" || exit 1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/$MODE/write" -o w -- $W > /dev/null 2> "$OUT/$MODE.write.log" || exit 1
  echo "$MODE passes done"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
def one(pat):
    f = glob.glob(os.path.join(out, pat), recursive=True)
    return f[0] if f else None
WANT = {"table": ["gemm_rows_sk"], "train": ["gemm_wgrad_cu_kernel", "gemm_rows_sk"], "knn": ["knn_score_kernel"]}
ALG = {  # algorithmic HBM bytes per launch
    ("table", "gemm_rows_sk"): "attribute rows gathered from the table: 19328 rows x 4096 x 4 B = 316.7 MB (+ ctx 0.5 MB, W_f 7.4 MB, q out 34.8 MB) = 359 MB",
    ("train", "gemm_wgrad_cu_kernel"): "dy = dq [19200 x 450] 34.6 MB + x = [attrs | ctx] 315 MB + dW out 7.4 MB = 357 MB",
    ("train", "gemm_rows_sk"): "forward feature GEMM of the train step: 19200 rows, 357 MB",
    ("knn", "knn_score_kernel"): "B T F 4 + B F 4 + B T 4 = 214 MB",
}
res = []
for mode, names in WANT.items():
    dur = collections.defaultdict(list)
    tr = one(f"{mode}/stats/**/*kernel_trace.csv")
    for r in csv.DictReader(open(tr)):
        for n in names:
            if n in r["Kernel_Name"]:
                dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    cnt = {}
    for what, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = one(f"{mode}/{what}/**/*counter_collection.csv")
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            for n in names:
                if n in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    acc[n].append(float(r["Counter_Value"]))
        cnt[ctr] = acc
    for n in names:
        d = dur[n][len(dur[n]) // 4:]  # (drop the first quarter: cold launches, clock ramp)
        if not d or not cnt["FETCH_SIZE"][n]:
            continue
        fe = cnt["FETCH_SIZE"][n][len(cnt["FETCH_SIZE"][n]) // 2:]
        wr = cnt["WRITE_SIZE"][n][len(cnt["WRITE_SIZE"][n]) // 2:]
        rb, wb = sum(fe) / len(fe) * 2 * 1024, sum(wr) / len(wr) * 1024
        us = sum(d) / len(d) / 1e3
        res.append(dict(workload=mode, kernel=n, launches_timed=len(d), avg_us=us, read_bytes=rb, write_bytes=wb,
                        hbm_bytes_per_launch=rb + wb, hbm_GBps=(rb + wb) / us / 1e3, peak_GBps=8000.0,
                        frac_of_hbm_peak=(rb + wb) / us / 1e3 / 8000.0, algorithmic_bytes=ALG[(mode, n)],
                        correction="FETCH_SIZE x2 (gfx950), WRITE_SIZE exact, KiB x1024 (MI355X_MICROARCH.md)"))
json.dump(res, open(os.path.join(out, "pmc_extra_summary.json"), "w"), indent=1)
for r in res:
    print("%-6s %-24s %8.1f us  %7.1f MB read  %6.1f MB written  %6.0f GB/s (%.0f %% of 8 TB/s)" % (
        r["workload"], r["kernel"], r["avg_us"], r["read_bytes"] / 1e6, r["write_bytes"] / 1e6, r["hbm_GBps"], 100 * r["frac_of_hbm_peak"]))
PY
