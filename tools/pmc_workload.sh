#!/bin/bash
# All PMC passes for one bench workload (run ON the GPU box via gpurun), then merge into profiles/pmc_summary.json.
# Separate rocprofv3 runs per counter group: --pmc only, no tracing flags (gpurun refuses pmc + trace combinations);
# FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots), as MI355X_MICROARCH.md prescribes.
# Usage: tools/pmc_workload.sh <outdir> <tag> [bench args...]     e.g. gpurun_out/pmc_c2 r02 --workload cornell_1024x1024_1024spp_8bounces
set -e
export TMPDIR=/tmp
out=$1; tag=$2; shift; shift
mkdir -p "$out"
run() { # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; exit 1; }
  echo "pass $name done"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summarize.py "$out" "$tag"
