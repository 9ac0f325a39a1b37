#!/bin/bash
# rocprofv3 --pmc passes over the split-precision feature GEMM: tools/split_pmc.sh [tag]   (MODE=bf16x3|fp16x2, default both)
# Three passes of <= 8 SQ counters + one TCC pass each for FETCH_SIZE / WRITE_SIZE (kernel-trace only beside them).
set -u
TAG=${1:-split_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="python3 $ROOT/tools/split_probe.py"
export DIAGS=0
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES"
i=0
for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- $W > "$OUT/p$i.out" 2> "$OUT/p$i.log" || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_rows_split_kernel" in r["Kernel_Name"] or "gemm_rows_sk" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)")
PY
