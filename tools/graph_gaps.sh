#!/bin/bash
# Inter-kernel gaps of the eager train loop against its hipGraph replay (kernel trace of tools/graph_train_probe.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/graph_gaps
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -o t -- python3 $ROOT/tools/graph_train_probe.py > "$OUT/run.log" 2> "$OUT/err.log" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, statistics as st
out = sys.argv[1]
tr = glob.glob(os.path.join(out, "tr/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))))
n = len(rows)
for name, lo, hi in (("eager (second quarter of the trace)", n // 4, n // 2), ("graph replays (last quarter)", 3 * n // 4, n)):
    seg = rows[lo:hi]
    gaps = [b[0] - a[1] for a, b in zip(seg, seg[1:])]
    small = [g for g in gaps if g < 50_000]
    busy = sum(e - s for s, e, _ in seg)
    print(f"{name}: {len(seg)} kernels, kernel time {busy / 1e3:.0f} us, span {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us, "
          f"median gap {st.median(gaps) / 1e3:.2f} us, mean gap (< 50 us) {st.mean(small) / 1e3:.2f} us, gaps > 50 us: {len(gaps) - len(small)}")
PY
cat "$OUT/run.log" | tail -3
