#!/bin/bash
# Fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes) of k_megakernel as a function of the samples-per-item
# chunk (GPU box): tools/pmc_chunk_sweep.sh 2 4 8 16 32 64 -> gpurun_out/chunk_sweep/summary.txt
export TMPDIR=/tmp
out=gpurun_out/chunk_sweep
rm -rf "$out"; mkdir -p "$out"
for c in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 120 rocprofv3 --pmc $ctr --output-format csv -d "$out/c${c}_$ctr" -- python3 tools/diag_chunk_render.py $c > "$out/c${c}_$ctr.log" 2>&1 || echo "pass $c $ctr failed" >> "$out/progress.log"
    echo "chunk $c $ctr done $(date +%T)" >> "$out/progress.log"
  done
done
python3 - "$out" "$@" > "$out/summary.txt" <<'PY'
import csv, re, sys
from pathlib import Path
out = Path(sys.argv[1])
print("chunk  kernel ms   Msamples/s   FETCH GB   WRITE GB   total GB   B/sample   (1024^2 x 256 spp per launch; counters of the second launch)")
for c in sys.argv[2:]:
    vals, ms = {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in (out / f"c{c}_{ctr}").rglob("*counter_collection.csv"):
            for row in csv.DictReader(f.open()):
                if "k_megakernel" in row.get("Kernel_Name", "") and row["Counter_Name"] == ctr:
                    rows.append(float(row["Counter_Value"]))
        vals[ctr] = rows[-1] if rows else float("nan")
        m = re.search(r"chunk \d+: ([\d.]+) ms", (out / f"c{c}_{ctr}.log").read_text())
        ms = float(m.group(1)) if m else float("nan")
    fetch, write = vals["FETCH_SIZE"] * 1024 * 2 / 1e9, vals["WRITE_SIZE"] * 1024 / 1e9
    n = 1024 * 1024 * 256
    print(f"{int(c):5d}  {ms:9.3f}  {n / ms / 1e3:11.1f}  {fetch:9.2f}  {write:9.2f}  {fetch + write:9.2f}  {(fetch + write) * 1e9 / n:9.1f}")
PY
find "$out" -name "*.csv" -delete
cat "$out/summary.txt"
