#!/bin/bash
# Extra PMC passes (vector-memory path: TA / TCP / TD + SQ wait breakdown) for one bench workload.  Every pass is its
# own rocprofv3 run under `timeout`; progress goes to <outdir>/progress.log.  Usage: tools/pmc_extra.sh <outdir> [bench args]
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
run() { name=$1; shift
  timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1 || echo "pass $name failed" >> "$out/progress.log"
  echo "pass $name done $(date +%T)" >> "$out/progress.log"
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq4 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
run ta1 TA_BUSY_avr TA_TA_BUSY_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run tcp4 TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run td1 TD_TD_BUSY_sum TD_TC_STALL_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
python3 - "$out" <<'PY'
import csv, sys, json
from collections import defaultdict
from pathlib import Path
out = Path(sys.argv[1])
vals = defaultdict(list)
for f in out.rglob("*counter_collection.csv"):
    for row in csv.DictReader(f.open()):
        n = row.get("Kernel_Name", "")
        if ("k_megakernel" not in n and "k_wf_" not in n) or "stats" in n: continue
        vals[(n.split("::")[-1].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
d = {f"{k[0]}:{k[1]}": sum(v) / len(v) for k, v in sorted(vals.items())}
for k, v in d.items(): print(f"{k:60s} {v:.6g}")
(out / "extra_summary.json").write_text(json.dumps(d, indent=1))
PY
