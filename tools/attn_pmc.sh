#!/bin/bash
# rocprofv3 passes over the attention kernels (K2, K4) alone: tools/attn_pmc.sh <tag>   (run on the GPU box via gpurun)
# One kernel-trace pass and three --pmc passes (<= 8 SQ counters each, kernel-trace only beside them); the summary
# (per kernel and grid: duration, MFMA-pipe busy share, wave-cycle split, instruction mix) goes to gpurun_out/<tag>/.
set -u
TAG=${1:-r02_attn}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="python3 $ROOT/tools/attn_pmc_workload.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- $W > /dev/null 2> "$OUT/stats.log" || exit 1
echo "trace pass done"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p1" -o p -- $W > /dev/null 2> "$OUT/p1.log" || exit 1
echo "pmc pass 1 done"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/p2" -o p -- $W > /dev/null 2> "$OUT/p2.log" || exit 1
echo "pmc pass 2 done"
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d "$OUT/p3" -o p -- $W > /dev/null 2> "$OUT/p3.log" || exit 1
echo "pmc pass 3 done"
python3 "$ROOT/tools/attn_pmc_summary.py" "$OUT"
