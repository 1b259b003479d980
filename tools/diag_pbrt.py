#!/usr/bin/env python3
"""Diagnostic (GPU box): render tests/golden/pbrt/cornell_box.pbrt and compare with pbrt's own output PNG."""
import sys
from pathlib import Path
import numpy as np
from PIL import Image
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = pkg.host_scene.load_pbrt(ROOT / "tests/golden/pbrt/cornell_box.pbrt")
with pkg.Renderer(0) as r:
    r.upload_scene(sc); r.set_limits(sc.max_depth); r.render(spp); r.sync()
    mean, m2 = r.download_film()
lin = np.clip(mean[..., :3], 0, 1)
srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
img = np.clip(srgb * 255 + 0.5, 0, 255).astype(np.uint8)
ref = np.asarray(Image.open(ROOT / "tests/golden/pbrt/pbrt_render_256.png"))[..., :3]
for name, cand in (("as is", img), ("mirrored", img[:, ::-1])):
    d = cand.astype(np.float64) - ref
    print(name, "mean abs diff /255:", np.abs(d).mean(), "mean signed:", d.mean(axis=(0, 1)), "p95:", np.percentile(np.abs(d), 95))
Image.fromarray(img).save(ROOT / "gpurun_out" / "pbrt_hip.png")
print("means", img.mean(axis=(0, 1)), ref.mean(axis=(0, 1)))
