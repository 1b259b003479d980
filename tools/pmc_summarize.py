#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs of tools/pmc_workload.sh for the timed megakernel dispatches and merge the
record into profiles/pmc_summary.json (read by bench.py for roofline.traffic / hbm_view / issue_view).
Usage: tools/pmc_summarize.py <outdir> [tag]"""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
out = Path(sys.argv[1])
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"

# the bench JSON line of each pass (workload, kernel, kspp, live kernel time of THAT profiled run)
bench = {}
for log in sorted(out.glob("*.log")):
    for line in log.read_text().splitlines():
        if line.startswith("{") and '"metric"' in line:
            bench[log.stem] = json.loads(line)
if not bench:
    sys.exit("no bench JSON line found in the pass logs")
first = next(iter(bench.values()))
workload, kernel = first["config"]["workload"], first["roofline"]["kernel"]
kspp = first["config"]["kspp"]
samples_per_launch = first["config"]["width"] * first["config"]["height"] * kspp

vals = defaultdict(list)
for f in out.rglob("*counter_collection.csv"):
    with f.open() as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if not re.search(rf"\b{kernel}\(", name) and name.split("(")[0].split("::")[-1].strip() != kernel:
                continue            # skips k_megakernel_bvh_stats and torch's fill kernels
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
for k, v in d.items():
    print(f"{k:28s} {v:.6g}")

import hashlib
_so = ROOT / "cuda-optix-pathtracing_amd" / "csrc" / "libdmt_hip.so"
lib_sha16 = (first.get("roofline", {}).get("pmc_record") or {}).get("loaded_lib_sha16") or \
    (hashlib.sha256(_so.read_bytes()).hexdigest()[:16] if _so.exists() else None)
rec = {"kspp": kspp, "kernel": kernel, "tag": tag, "samples_per_launch": samples_per_launch, "lib_sha16": lib_sha16}
rec.update(d)
ms = bench.get("sq1", first)["roofline"]["avg_launch_ms"]
rec["kernel_ms"] = ms
if "GRBM_GUI_ACTIVE" in d and "grbm" in bench:
    rec["clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / 8.0 / (bench["grbm"]["roofline"]["avg_launch_ms"] * 1e-3) / 1e9, 4)
if "SQ_WAVE_CYCLES" in d:
    rec["valu_active_share_per_wave"] = round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], 4)
    rec["lane_utilisation"] = round(d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]), 4)
    rec["wait_any_share"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    rec["wait_inst_any_share"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
    rec["valu_insts_per_sample"] = round(d["SQ_INSTS_VALU"] / samples_per_launch, 2)
    clk = rec.get("clock_ghz", 2.4)
    cycles = d["SQ_BUSY_CYCLES"] / 32.0 if d.get("SQ_BUSY_CYCLES") else ms * 1e-3 * clk * 1e9   # same pass (bench.py build_roofline)
    rec["valu_busy"] = round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024 * cycles), 4)
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    # gfx950: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reads HALF the bytes (MI355X_MICROARCH.md, HBM)
    rec["FETCH_SIZE_KiB"], rec["WRITE_SIZE_KiB"] = d["FETCH_SIZE"], d["WRITE_SIZE"]
    rec["hbm_bytes_per_launch"] = int(d["FETCH_SIZE"] * 1024 * 2 + d["WRITE_SIZE"] * 1024)
    rec["hbm_GBs"] = round(rec["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
for k in ("valu_busy", "lane_utilisation", "wait_any_share", "valu_insts_per_sample", "clock_ghz", "hbm_GBs"):
    if k in rec:
        print(f"{k:28s} {rec[k]}")
(out / "summary.json").write_text(json.dumps(rec, indent=1))

prof = ROOT / "profiles" / tag
prof.mkdir(parents=True, exist_ok=True)
(prof / f"pmc_{workload}.json").write_text(json.dumps(rec, indent=1))
summ = ROOT / "profiles" / "pmc_summary.json"
try:
    S = json.loads(summ.read_text())
except Exception:
    S = {"workloads": {}}
S["note"] = ("rocprofv3 --pmc passes of tools/pmc_workload.sh (SQ x2, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE; separate runs, no tracing "
             "flags), averaged over the timed dispatches of the workload's kernel; hbm_bytes_per_launch = FETCH_SIZE[KiB]*1024*2 (gfx950 "
             "FETCH_SIZE reads half, MI355X_MICROARCH.md HBM; calibrated for wide streams, uncalibrated for per-lane gathers) + WRITE_SIZE[KiB]*1024; "
             "kernel_ms = live HIP-event time of the SQ pass; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel time of the GRBM pass")
S.setdefault("workloads", {})[workload] = rec
summ.write_text(json.dumps(S, indent=1))
print("merged into", summ)
