#!/bin/bash
# Vector-memory path counters of one bench workload (GPU box): is the CU's address path (TA) the co-bound of the BVH step?
# Separate --pmc passes, no tracing flags.  Usage: tools/pmc_vmem.sh <outdir> [bench args...]
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
pass() { # name counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary "${ARGS[@]}" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return; }
  python3 - "$out/$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "megakernel" in r.get("Kernel_Name", "") and "stats" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()): print(f"{k:40s} {v:.6g}")
PY
  find "$out/$name" -name "*.csv" -delete
}
ARGS=("$@")
pass ta1 TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass sq4 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
pass tcp TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum
