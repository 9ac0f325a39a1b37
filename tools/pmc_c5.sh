#!/bin/bash
# rocprofv3 passes over the C5 eval forward (tools/bench_configs.py, ONLY=C5): kernel stats + one SQ counter pass.
#   tools/pmc_c5.sh <tag>     -> gpurun_out/<tag>/{c5_kernel_stats.csv, c5_sq.csv}
set -u
TAG=${1:-c5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-C5}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5stats" -o s -- python3 $ROOT/tools/bench_configs.py > "$OUT/c5_stats.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/c5sq" -o q -- python3 $ROOT/tools/bench_configs.py > "$OUT/c5_sq.log" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
st = glob.glob(os.path.join(out, "c5stats/**/*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(st)))[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
f = glob.glob(os.path.join(out, "c5sq/**/*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "gemm_rows" not in k and "cross" not in k: continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    print(k, "n=%d" % len(d["GRBM_GUI_ACTIVE"]), "mfma_busy=%.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, 1024 * gui / 8)),
          "wait_any=%.3f" % (m.get("SQ_WAIT_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))),
          "wait_inst=%.3f" % (m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))),
          "active=%.3f" % (m.get("SQ_ACTIVE_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))),
          "lds_conflict=%.0f" % m.get("SQ_LDS_BANK_CONFLICT", 0), "gui=%.0f" % gui)
PY
