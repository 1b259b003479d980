#!/usr/bin/env python3
"""Loop profile of the BVH megakernel on BASELINE config 4 (counting build): wave-level iteration counts and lane
utilisation per step kind, plus the timed kernel's ms for the same spp.  Usage: tools/diag_bvh_profile.py [spp] [ntri]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
scene = pkg.host_scene.random_triangle_scene(ntri, width=1024, height=1024)
with pkg.Renderer(0) as r:
    r.upload_scene(scene); r.set_limits(8); r.set_accel(1)
    r.render(spp); r.sync(); r.kernel_time(reset=True)
    r.film_clear(); r.render(spp); r.sync()
    ms, n = r.kernel_time(reset=True)
    p = r.render_profile(spp)
    S = p["samples"]
    print(f"timed kernel: {ms / n:.2f} ms for {spp} spp -> {1024 * 1024 * spp / (ms / n) / 1e3:.1f} Msamples/s")
    print({k: round(v / S, 3) for k, v in p.items()})
    waves = {k: p[k] / 64 for k in ("it_node", "it_leaf", "it_shade", "it_outer", "it_prep")}
    print("wave iterations per sample:", {k: round(v / S, 4) for k, v in waves.items()})
    print("lane utilisation: node %.3f leaf %.3f shade %.3f prep %.3f" % (
        p["node_visits"] / max(1, p["it_node"]), p["lanes_leaf"] / max(1, p["it_leaf"]),
        p["lanes_shade"] / max(1, p["it_shade"]), p["lanes_prep"] / max(1, p["it_prep"])))
    info = r.kernel_info()
    waves_res = info["cu_count"] * info["blocks_per_cu"] * 4
    t_wave_sample = (ms / n) * 1e-3 * waves_res / (1024 * 1024 * spp)
    print(f"resident waves {waves_res}; wave-time per sample {t_wave_sample * 1e6:.2f} us; per wave-iteration (node+leaf+shade) "
          f"{t_wave_sample * 1e9 / ((waves['it_node'] + waves['it_leaf'] + waves['it_shade']) / S):.0f} ns")
