#!/usr/bin/env python3
"""Diagnostic (GPU box): throughput of the megakernel for several depth caps / sizes (HIP-event time)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
depths = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4, 8, 32]
scene = pkg.host_scene.cornell_box(res, res)
with pkg.Renderer(0) as r:
    r.upload_scene(scene)
    print(r.kernel_info())
    for d in depths:
        r.set_limits(d)
        r.film_clear(); r.render(spp); r.sync(); r.kernel_time(reset=True)
        r.film_clear(); r.render(spp); r.render(spp, sample_offset=spp)
        ms, n = r.kernel_time(reset=True)
        print(f"depth {d:2d}: {ms / n:9.3f} ms per launch  {res * res * spp / (ms / n) / 1e3:9.1f} Msamples/s   ns/sample-lane {ms / n * 1e6 / (res * res * spp) * 256 * 4 * 64:8.1f}")
