#!/usr/bin/env python3
"""Diagnostic (GPU box): throughput of the BVH megakernel on the 1M-triangle scene."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ntri = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
scene = pkg.host_scene.random_triangle_scene(ntri, width=res, height=res)
with pkg.Renderer(0) as r:
    r.upload_scene(scene); r.set_limits(8); r.set_accel(1)
    print(r.kernel_info())
    r.render(spp); r.sync(); r.kernel_time(reset=True)
    r.film_clear(); r.render(spp); r.render(spp, sample_offset=spp)
    ms, n = r.kernel_time(reset=True)
    st = r.render_stats(1, sample_offset=2 * spp)
    print(f"{ms / n:9.3f} ms per launch  {res * res * spp / (ms / n) / 1e3:9.1f} Msamples/s", {k: round(v / st['samples'], 2) for k, v in st.items()})
    pr = r.render_profile(32, sample_offset=2 * spp + 1, region=(256, 256, 768, 768))
    n = pr["samples"]
    print({k: round(v / n, 3) for k, v in pr.items()})
    print("wave iterations per sample (x64 lanes):", {k: round(pr[k] / n, 2) for k in ("it_node", "it_leaf", "it_shade", "it_outer", "it_prep")})
    print("lane utilisation: node %.3f leaf %.3f shade %.3f prep %.3f" % (pr["node_visits"] / max(pr["it_node"], 1), pr["lanes_leaf"] / max(pr["it_leaf"], 1), pr["lanes_shade"] / max(pr["it_shade"], 1), pr["lanes_prep"] / max(pr["it_prep"], 1)))
