#!/usr/bin/env python3
"""Diagnostic (GPU box): build + validate the BVH of a random soup of n triangles and compare closest hits with brute force."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
from test_parity_gpu import _random_soup, _rays
n = int(sys.argv[1])
xs, ys, zs = _random_soup(n, n)
print(pkg.bvh_validate(xs, ys, zs))
with pkg.Renderer(0) as r:
    r.upload_triangles(xs, ys, zs, np.zeros(n, np.uint32))
    o, d = _rays(8192, n + 1)
    bi, bt = r.test_closest_hit(o, d)
    r.set_accel(1)
    ai, at = r.test_closest_hit(o, d)
    bad = np.nonzero(ai != bi)[0]
    print("hits brute", (bi >= 0).sum(), "hits bvh", (ai >= 0).sum(), "mismatch", len(bad))
    for k in bad[:10]:
        print(k, "brute", bi[k], bt[k], "bvh", ai[k], at[k], "o", o[k], "d", d[k])
