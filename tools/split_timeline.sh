#!/bin/bash
# Kernel timeline of ONE graphed train step (start / end of every dispatch relative to the step's first kernel) under a
# setting of tools/ab_train_graph.py: shows what the two streams of the backward's split overlap.
# usage: gpurun -- bash tools/split_timeline.sh "split=1"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/split_timeline
rm -rf "$OUT"; mkdir -p "$OUT"
export ROUNDS=2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o s -- python3 $ROOT/tools/ab_train_graph.py "$1" > "$OUT/run.log" 2> "$OUT/prof.log" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
f = glob.glob(os.path.join(out, "tr/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "gemm_rows_sk" in r["Kernel_Name"]]
i0, i1 = starts[-3], starts[-2]
# a step begins a few launches before its feature GEMM: take everything from the previous step's adam_kernel end
t0 = int(rows[i0]["Start_Timestamp"])
print("step = %.1f us between two feature GEMMs" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
for r in rows[i0:i1]:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} -> {e:9.1f}  ({e - s:7.1f})  q{r.get('Queue_Id', '?'):>3s}  grid {r.get('Grid_Size', '?'):>8s}  {n[:70]}")
PY
