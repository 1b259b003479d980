#!/bin/bash
# Round profile collection (GPU box): rocprofv3 --kernel-trace --stats of the bench command + PMC passes, per workload.
# Usage: tools/collect_profiles.sh <tag> <workload> [more workloads...]   -> gpurun_out/profiles_<tag>/ (copy into profiles/<tag>/)
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/profiles_$tag
mkdir -p "$out"
for w in "$@"; do
  echo "== $w $(date +%T)" >> "$out/progress.log"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -- python3 bench.py --steps 3 --warmup 1 --workload "$w" > "$out/bench_under_rocprof_$w.log" 2>&1 || echo "trace $w failed" >> "$out/progress.log"
  f=$(find "$out/trace_$w" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_$w.csv"
  grep '^{' "$out/bench_under_rocprof_$w.log" > "$out/bench_under_rocprof_$w.json"
  rm -rf "$out/trace_$w"
  timeout -k 10 900 tools/pmc_workload.sh "$out/pmc_$w" "$tag" --workload "$w" >> "$out/progress.log" 2>&1 || echo "pmc $w failed" >> "$out/progress.log"
  find "$out/pmc_$w" -name "*.csv" -delete
done
cp -r profiles/$tag "$out/profiles_dir" 2>/dev/null
cp profiles/pmc_summary.json "$out/pmc_summary.json"
echo "done $(date +%T)" >> "$out/progress.log"
