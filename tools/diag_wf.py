#!/usr/bin/env python3
"""Diagnostic (GPU box): BVH wavefront vs megakernel on the random-triangle scene.  Usage: diag_wf.py [spp] [ntri] [strategy]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
strategy = int(sys.argv[3]) if len(sys.argv) > 3 else 2
scene = pkg.host_scene.random_triangle_scene(ntri, width=1024, height=1024)
with pkg.Renderer(0) as r:
    r.upload_scene(scene); r.set_limits(8); r.set_accel(1); r.set_bvh_strategy(strategy)
    r.render(spp); r.sync(); r.kernel_time(reset=True)
    r.film_clear(); r.render(spp); r.render(spp, sample_offset=spp); r.sync()
    ms, n = r.kernel_time(reset=True)
    print(f"strategy {strategy}: {ms / n:9.3f} ms per launch of {spp} spp  {1024 * 1024 * spp / (ms / n) / 1e3:9.1f} Msamples/s")
    if len(sys.argv) > 4:
        p = r.render_profile(4, sample_offset=2 * spp)
        S = p["samples"]
        print({k: round(v / S, 3) for k, v in p.items()})
        print("wave iterations per sample:", {k: round(p[k] / 64 / S, 4) for k in ("it_node", "it_leaf")},
              "lane utilisation: node %.3f leaf %.3f" % (p["node_visits"] / max(1, p["it_node"]), p["lanes_leaf"] / max(1, p["it_leaf"])))
