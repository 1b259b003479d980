#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N gloo ranks share the device (RCCL refuses two ranks on one GPU).
# Usage: tools/rehearse_multirank_one_gpu.sh N [extra bench.py args];  DMT_HIP_LIB selects an alternative build.
N=${1:-4}; shift
export DMT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
exec timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus "$N" --steps 2 --warmup 1 --no-cpu-baseline "$@"
