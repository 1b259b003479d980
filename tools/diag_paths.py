#!/usr/bin/env python3
"""Diagnostic (GPU box): per-sample radiance, HIP path vs CPU oracle, to localise divergence."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
O = g.load_oracle()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 32
scene = O.cornell_box(res, res)
ys, xs, ss = np.meshgrid(np.arange(res), np.arange(res), np.arange(spp), indexing="ij")
pxs, pys, ss = xs.ravel().astype(np.int32), ys.ravel().astype(np.int32), ss.ravel().astype(np.int32)
with pkg.Renderer(0) as r:
    r.upload_scene(scene)
    r.set_limits(depth)
    L = r.test_trace_samples(pxs, pys, ss)
ref = O.trace_samples(scene, pxs, pys, ss, max_depth=depth)
err = np.abs(L - ref).max(axis=1)
rel = err / np.maximum(np.abs(ref).max(axis=1), 1e-6)
print("samples", len(err), "max abs", err.max(), "mean abs", err.mean())
for thr in (1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
    print(f"  frac abs err > {thr:g}: {(err > thr).mean():.5f}   rel > {thr:g}: {(rel > thr).mean():.5f}")
bad = np.argsort(-err)[:12]
for i in bad:
    print(f"px {pxs[i]:3d} py {pys[i]:3d} s {ss[i]:3d}  gpu {L[i]}  ref {ref[i]}  err {err[i]:.3e}")
np.savez("gpurun_out/diag_paths.npz", px=pxs, py=pys, s=ss, L=L, ref=ref)

# per-bounce logs of the worst few
np.set_printoptions(precision=6, suppress=True, linewidth=200)
with pkg.Renderer(0) as r:
    r.upload_scene(scene)
    r.set_limits(depth)
    for i in bad[:4]:
        a, La = r.test_trace_log(pxs[i], pys[i], ss[i])
        b, Lb = O.trace_log(scene, pxs[i], pys[i], ss[i], max_depth=depth)
        print(f"=== px {pxs[i]} py {pys[i]} s {ss[i]}: gpu records {len(a)} L {La} | cpu records {len(b)} L {Lb}")
        for k in range(max(len(a), len(b))):
            ra = a[k] if k < len(a) else None
            rb = b[k] if k < len(b) else None
            print("  gpu", ra)
            print("  cpu", rb)
