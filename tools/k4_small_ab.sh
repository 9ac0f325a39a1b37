#!/bin/bash
# A/B of the scoring kernel's two eval-mode variants at small batches under rocprofv3 --kernel-trace (kernel durations,
# not host-paced loops): tools/k4_small_ab.sh <tag>    (run on the GPU box via gpurun)
set -u
TAG=${1:-k4_small_ab}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BS=${BS:-64,128,192,256}
for T in 2 3; do
  TUNE7=$T timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t$T" -o s -- python3 $ROOT/tools/attn_pmc_workload.py > /dev/null 2> "$OUT/t$T.log" || exit 1
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for t in ("t2", "t3"):
    f = glob.glob(os.path.join(out, t, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    runs, last = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if "cross_" not in n:
            last = None
            continue
        key = (n.split("(")[0][:60], r.get("Grid_Size") or r["Grid_Size_X"])
        if key != last:
            runs.append([key, []])
            last = key
        runs[-1][1].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for key, d in runs:
        d = d[10:] if len(d) > 10 else d
        print(t, key, "n=%d avg %.2f us min %.2f us" % (len(d), sum(d) / len(d) / 1e3, min(d) / 1e3))
PY
