#!/usr/bin/env python3
"""Diagnostic (GPU box): BVH vs brute force closest hit for exactly axis-parallel rays over a flat floor."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
def tri(p0, p1, p2):
    return [p0[0], p1[0], p2[0], 0.0], [p0[1], p1[1], p2[1], 0.0], [p0[2], p1[2], p2[2], 0.0]
with pkg.Renderer(0) as r:
    for nfloor in (1, 2, 3):
        t = [tri((-1, 0, -1), (1, 0, -1), (1, 0, 1)), tri((-1, 0, -1), (1, 0, 1), (-1, 0, 1)), tri((2, 0, 2), (3, 0, 2), (3, 0, 3))][:nfloor]
        xs = np.array([a[0] for a in t], np.float32).reshape(-1); ys = np.array([a[1] for a in t], np.float32).reshape(-1)
        zs = np.array([a[2] for a in t], np.float32).reshape(-1)
        r.upload_triangles(xs, ys, zs, np.zeros(nfloor, np.uint32))
        gg = np.linspace(-1.5, 3.5, 41, dtype=np.float32)
        gx, gz = np.meshgrid(gg, gg)
        n = gx.size
        o = np.stack([gx.ravel(), np.full(n, 5.0, np.float32), gz.ravel()], axis=1).astype(np.float32)
        d = np.tile(np.array([0.0, -1.0, 0.0], np.float32), (n, 1))
        o = np.concatenate([o, o * np.array([1, -1, 1], np.float32), np.stack([np.full(n, -9.0, np.float32), gz.ravel() * 0, gx.ravel()], axis=1)])
        d = np.concatenate([d, -d, np.tile(np.array([1.0, 0.0, 0.0], np.float32), (n, 1))])
        r.set_accel(0); bi, bt = r.test_closest_hit(o, d)
        r.set_accel(1); ai, at = r.test_closest_hit(o, d)
        r.set_accel(0)
        bad = np.nonzero((ai != bi) | (at.view(np.uint32) != bt.view(np.uint32)))[0]
        print(f"nfloor {nfloor}: {len(bad)} mismatches of {len(o)}; hits brute {int((bi >= 0).sum())} bvh {int((ai >= 0).sum())}")
        for k in bad[:12]:
            print("   ray", k, "o", o[k], "d", d[k], "brute", bi[k], bt[k], "bvh", ai[k], at[k])
