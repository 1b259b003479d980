#!/bin/bash
# HBM traffic of the bench workload's kernel from PMC counters (GPU box).  Separate rocprofv3 passes for
# FETCH_SIZE and WRITE_SIZE (TCC slots), no tracing flags.  Usage: tools/pmc_traffic.sh <outdir> [bench args]
set -e
export TMPDIR=/tmp
out=$1; shift
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$out/$c.log" 2>&1
done
python3 tools/pmc_summarize.py "$out"
