#!/bin/bash
# Kernel trace of the train step (rocprofv3 --kernel-trace --stats over tools/train_host_time.py); prints per-kernel
# calls / avg / share.  Run on the GPU box: gpurun -- bash tools/train_trace.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/train_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 $ROOT/tools/train_host_time.py > "$OUT/run.log" 2> "$OUT/stats.log" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
st = glob.glob(os.path.join(out, "stats/**/*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(st)))
steps = 5 + 90 + 10
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU kernel time per train step: {tot / steps / 1e3:.1f} us over {sum(int(r['Calls']) for r in rows) / steps:.1f} launches")
for r in rows[:40]:
    print(f"{int(r['Calls']) / steps:6.2f}/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  {float(r['Percentage']):5.1f} %  {r['Name'][:90]}")
PY
cat "$OUT/run.log"
