#!/bin/bash
# Collect the rocprofv3 artefacts of profiles/README.md for the current build (run on the GPU box via gpurun):
#   tools/profile_round.sh r01_d
# Each counter set gets its own pass (kernel-trace only), every pass is bounded by `timeout`.
set -u
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the stats pass keeps the split-precision and B = 1024 / 4096 scoring side passes -- their kernels have rows of their own in the
# summary; the side pass with full profiles launches the SAME feature-GEMM kernel over more rows and is left out of every pass, so
# that the kernel's per-launch averages are the headline batch's)
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-table --no-fold --no-full-profiles --no-configs --train-steps 0"
BENCH_PMC="$BENCH --no-split --no-scoring-scaling"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log" || exit 1
echo "stats pass done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- $BENCH_PMC > /dev/null 2> "$OUT/fetch.log" || exit 1
echo "FETCH_SIZE pass done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- $BENCH_PMC > /dev/null 2> "$OUT/write.log" || exit 1
echo "WRITE_SIZE pass done"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -o q -- $BENCH_PMC > /dev/null 2> "$OUT/sq.log" || exit 1
echo "SQ pass done"
cd "$ROOT" && timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log" || exit 1
echo "default bench done"
python3 - "$OUT" "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def one(pat):
    f = glob.glob(os.path.join(out, pat), recursive=True)
    return f[0] if f else None
# kernel stats + per-grid split
res = {}
st = one("stats/**/*kernel_stats.csv")
tr = one("stats/**/*kernel_trace.csv")
rows = list(csv.DictReader(open(tr)))
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"], r.get("Grid_Size_X") or r.get("Grid_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(out, "kernels_by_grid.csv"), "w") as fh:
    fh.write("kernel,grid,calls,avg_us,median_us,min_us\n")
    for (k, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        fh.write('"%s",%s,%d,%.2f,%.2f,%.2f\n' % (k[:120], g, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3))
def pmc(pat, name):
    f = one(pat)
    vals = []
    for r in csv.DictReader(open(f)):
        if ("gemm_rows_skc_kernel" in r["Kernel_Name"] or "gemm_rows_sk_kernel" in r["Kernel_Name"] or "gemm_rows_cu_kernel<0, 3>" in r["Kernel_Name"]) and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
    return vals
fe, wr = pmc("fetch/**/*counter_collection.csv", "FETCH_SIZE"), pmc("write/**/*counter_collection.csv", "WRITE_SIZE")
B, L, N, K, g = 128, 50, 101, 4102, 450
alg = (B * (L + N) * K + g * K + B * (L + N) * g) * 4
t = {"kernel": "gemm_rows_skc_kernel (feature GEMM launch, grid 256 x 768: 255 workgroups on the tiles of the rows with id != 0 + the item-row gather's; partial tiles of the hand-over and the cleared rows included)",
     "source": "separate rocprofv3 --pmc passes of bench.py --steps 20 (tools/profile_round.sh)",
     "FETCH_SIZE_KiB_avg": sum(fe) / len(fe), "WRITE_SIZE_KiB_avg": sum(wr) / len(wr),
     "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM), WRITE_SIZE exact, KiB x1024",
     "read_bytes": sum(fe) / len(fe) * 2 * 1024, "write_bytes": sum(wr) / len(wr) * 1024, "algorithmic_bytes": alg}
t["hbm_bytes_per_launch"] = t["read_bytes"] + t["write_bytes"]
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(out)))
t["gemm_hip_sha256"] = hashlib.sha256(open(os.path.join(root, "carca_replication_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()
json.dump(t, open(os.path.join(out, "feat_gemm_traffic.json"), "w"), indent=1)
print(json.dumps(t))
PY
echo "summary done"
