"""Subprocess body of test_bvh_overflow_stack_variant: run with DMT_HIP_LIB pointing at a variant build of the HIP
library.  BVH traversal must equal brute force bit for bit (closest hits and whole films), and the run must have
pushed traversal-stack entries into the global overflow area."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as graft  # noqa: E402
from test_parity_gpu import _random_soup, _rays  # noqa: E402

pkg = graft.load_package()
out = {"lib": str(pkg.library_path())}
with pkg.Renderer(0) as r:
    n = 20000
    xs, ys, zs = _random_soup(n, 7)
    r.upload_triangles(xs, ys, zs, np.zeros(n, np.uint32))
    o, d = _rays(16384, 11)
    r.set_accel(0)
    bi, bt = r.test_closest_hit(o, d)
    r.set_accel(1)
    ai, at = r.test_closest_hit(o, d)
    out["closest_equal"] = bool(np.array_equal(ai, bi) and np.array_equal(at.view(np.uint32), bt.view(np.uint32)))
    out["hit_share"] = float((ai >= 0).mean())
    scene = pkg.host_scene.random_triangle_scene(30000, width=64, height=64)
    r.upload_scene(scene)
    r.set_limits(8)
    films = []
    for mode in (0, 1):
        r.set_accel(mode)
        r.film_clear()
        r.render(8)
        r.sync()
        films.append(r.download_film())
    out["film_equal"] = bool(np.array_equal(films[0][0], films[1][0]) and np.array_equal(films[0][1], films[1][1]))
    out["film_max"] = float(films[1][0][..., :3].max())
    prof = r.render_profile(1, sample_offset=8)
    out["overflow_pushes"] = int(prof["overflow_pushes"])
    out["node_visits"] = int(prof["node_visits"])
print(json.dumps(out))
