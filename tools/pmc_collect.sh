#!/bin/bash
# Collect PMC counters for the bench workload (run ON the GPU box via gpurun).  Separate rocprofv3
# runs per counter group: --pmc only, no tracing flags (gpurun refuses pmc + trace combinations).
# Usage: tools/pmc_collect.sh <outdir> [bench args...]
set -e
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
run() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; }
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
find "$out" -name "*counter_collection.csv" | head -20
