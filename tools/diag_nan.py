#!/usr/bin/env python3
"""Diagnostic (GPU box): non-finite / wrong-count pixels of a render, and what the oracle gives there."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
res = int(sys.argv[1]); spp = int(sys.argv[2]); depth = int(sys.argv[3])
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "oracle hw threads", O.lib().oracle_hardware_threads())
scene = pkg.host_scene.cornell_box(res, res)
with pkg.Renderer(0) as r:
    r.upload_scene(scene); r.set_limits(depth); r.render(spp)
    mean, m2 = r.download_film()
    bad = ~np.isfinite(mean).all(axis=2) | ~np.isfinite(m2).all(axis=2)
    print("non-finite pixels", bad.sum(), "count!=spp", (m2[..., 3] != spp).sum())
    ys, xs = np.nonzero(bad)
    osc = O.cornell_box(res, res)
    for y, x in list(zip(ys, xs))[:6]:
        # find the offending samples
        ss = np.arange(spp, dtype=np.int32)
        L = r.test_trace_samples(np.full(spp, x, np.int32), np.full(spp, y, np.int32), ss)
        Lo = O.trace_samples(osc, np.full(spp, x, np.int32), np.full(spp, y, np.int32), ss, max_depth=depth)
        nb = np.nonzero(~np.isfinite(L).all(axis=1))[0]
        no = np.nonzero(~np.isfinite(Lo).all(axis=1))[0]
        print(f"pixel ({x},{y}): gpu non-finite samples {nb[:8]} oracle non-finite samples {no[:8]}")
        for s in nb[:2]:
            a, La = r.test_trace_log(x, y, int(s)); b, Lb = O.trace_log(osc, x, y, int(s), max_depth=depth)
            np.set_printoptions(precision=6, suppress=True, linewidth=220)
            print("  gpu L", La, "cpu L", Lb)
            for k in range(max(len(a), len(b))):
                print("   gpu", a[k] if k < len(a) else None); print("   cpu", b[k] if k < len(b) else None)
