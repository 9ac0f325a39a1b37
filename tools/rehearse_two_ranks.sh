#!/bin/bash
# The N > 1 control flow of bench.py on a ONE-GPU box: two ranks share device 0 over gloo (RCCL refuses two ranks per
# device).  Exercises the sharded eval (users sharded, timing barrier, max over ranks) and the sharded TRAIN step (global
# mask count, flat gradient buffer reduced in place with the early range started under the backward's last kernel,
# one-launch Adam) end to end; the numbers say nothing about xGMI.
#   gpurun -- bash tools/rehearse_two_ranks.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export CARCA_BENCH_DEVICE=0 CARCA_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --steps 10 --warmup 3 --no-fold --no-table --train-steps 6
