#!/usr/bin/env python3
"""Diagnostic (GPU box): render time of rank 0's share for world = 1,2,4,8 and several chunk sizes.
time(world) * world / time(1) = what tile scaling would cost before the film combine."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
chunks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [128, 64, 32, 16]
scene = pkg.host_scene.cornell_box(res, res)
with pkg.Renderer(0) as r:
    r.upload_scene(scene)
    r.set_limits(8)
    base = {}
    for world in (1, 2, 4, 8):
        r.set_partition(0, world)
        for c in chunks:
            r.set_chunk(c)
            r.film_clear(); r.render(spp); r.sync(); r.kernel_time(reset=True)
            r.film_clear(); r.render(spp); r.sync()
            ms, n = r.kernel_time(reset=True)
            t = ms / n
            base.setdefault(c, t)
            print(f"world {world} chunk {c:4d}: {t:9.3f} ms   x world / t(1) = {t * world / base[c]:.3f}   vs best t(1): {t * world / min(base.values()):.3f}", flush=True)
