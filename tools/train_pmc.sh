#!/bin/bash
# rocprofv3 --pmc passes over the kernels of the (eager) train step: tools/train_pmc.sh <kernel-name-substring> [tag]
# Three passes of <= 8 SQ counters (kernel-trace only beside them); prints per-kernel averages of every counter.
set -u
PAT=${1:-row_chain_bwd_kernel}
TAG=${2:-train_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="python3 $ROOT/tools/ab_train.py"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p1" -o p -- $W "" > /dev/null 2> "$OUT/p1.log" || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/p2" -o p -- $W "" > /dev/null 2> "$OUT/p2.log" || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d "$OUT/p3" -o p -- $W "" > /dev/null 2> "$OUT/p3.log" || exit 1
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, os, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):14.0f}   ({len(v)} dispatches)")
PY
