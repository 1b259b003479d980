#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/pmc_collect.sh output) for the megakernel dispatches."""
import csv, json, sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
vals = defaultdict(list)
for f in out.rglob("*counter_collection.csv"):
    with f.open() as fh:
        for row in csv.DictReader(fh):
            if "k_megakernel" not in row.get("Kernel_Name", ""):
                continue
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
for k, v in summary.items():
    print(f"{k:28s} {v:.6g}")
d = summary
if "SQ_WAVE_CYCLES" in d and "SQ_ACTIVE_INST_VALU" in d:
    print("VALU active / wave cycles        ", d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"])
if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d:
    print("VALU lane utilisation            ", d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]))
if "SQ_WAIT_ANY" in d:
    print("wait_any / wave cycles           ", d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"])
    print("wait_inst_any / wave cycles      ", d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"])
if "FETCH_SIZE" in d:
    # gfx950: FETCH_SIZE is in KiB and reads HALF the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM)
    print("HBM read bytes (FETCH_SIZE KiB*1024, x2 gfx950 correction):", d["FETCH_SIZE"] * 1024, d["FETCH_SIZE"] * 2048)
if "WRITE_SIZE" in d:
    print("HBM write bytes (WRITE_SIZE KiB*1024):", d["WRITE_SIZE"] * 1024)
(out / "summary.json").write_text(json.dumps(summary, indent=1))
