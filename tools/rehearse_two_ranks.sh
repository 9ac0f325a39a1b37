#!/bin/bash
# The N > 1 path on a ONE-GPU box: two ranks share device 0 over gloo (RCCL refuses two ranks per device).
#   1. tools/two_rank_check.py ASSERTS that two ranks x B/2 users on the HIP path reproduce the single-process B-user
#      step (gradients, losses, parameters after 2 Adam steps, identical replicas; deterministic mode) -- PASS / FAIL;
#   2. bench.py --gpus 2: the sharded eval (timing barrier, max over ranks) and the sharded train step end to end; the
#      numbers say nothing about xGMI.
#   gpurun -- bash tools/rehearse_two_ranks.sh  [log]       (the log goes to gpurun_out/two_rank_rehearsal.log by default)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LOG=${1:-gpurun_out/two_rank_rehearsal.log}
mkdir -p "$(dirname "$LOG")"
export CARCA_BENCH_DEVICE=0 CARCA_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 \
  tools/two_rank_check.py 2>&1 | grep -v "amdgpu.ids" | tee "$LOG"
rc=${PIPESTATUS[0]}
[ "$rc" -eq 0 ] || { echo "two-rank check FAILED (rc $rc)" | tee -a "$LOG"; exit "$rc"; }
# (the PLAIN invocation: no WORLD_SIZE in the environment -- bench.py starts torch.distributed.run itself, as a child process)
echo "== python bench.py --gpus 2 (launches its own ranks) ==" | tee -a "$LOG"
timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 3 --no-fold --no-table --no-split --train-steps 6 2>&1 | grep -v "amdgpu.ids" | tee -a "$LOG"
exit "${PIPESTATUS[0]}"
