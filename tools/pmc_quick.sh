#!/bin/bash
# quick SQ counter pass for the bench workload (GPU box). Usage: tools/pmc_quick.sh <outdir> [bench args]
set -e
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$out/sq1" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/sq1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --output-format csv -d "$out/sq2" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/sq2.log" 2>&1
python3 tools/pmc_summarize.py "$out"
