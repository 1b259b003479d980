#!/usr/bin/env python3
"""Diagnostic (GPU box): textured-metallic blend on scene_test.json, GPU vs oracle per pixel at 1 spp."""
import json, shutil, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
tmp = Path(tempfile.mkdtemp())
shutil.copytree(ROOT / "tests" / "golden" / "scene_test", tmp / "s")
j = json.loads((tmp / "s" / "scene_test.json").read_text())
j["textures"].append({"name": "m", "type": "metallic", "path": "./res/textures/chippedPaint/Paint_Chipped_1K_roughness.png"})
j["materials"][0]["metallic"] = "m"
(tmp / "s" / "m.json").write_text(json.dumps(j))
hs = pkg.host_scene.load_json(tmp / "s" / "m.json")
osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera); osc.set_envmap(hs.env_rgb)
osc.set_textures(hs.tex_rgba, hs.tex_desc, hs.mat_tex, hs.tri_uv)
w, h = hs.width, hs.height
y0, y1 = h // 2 - 8, h // 2 + 8
with pkg.Renderer(0) as r:
    r.upload_scene(hs); r.set_accel(1)
    for depth in (1, 2, hs.max_depth):
        r.set_limits(depth)
        for s in range(2):
            r.film_clear(); r.render(1, sample_offset=s); mean, _ = r.download_film()
            om = O.render(osc, 1, sample_offset=s, max_depth=depth, region=(0, y0, w, y1), threads=16)[0]
            a, b = mean[y0:y1, :, :3], om[y0:y1, :, :3]
            diff = np.abs(a - b).max(axis=2)
            big = diff > 1e-3 * np.maximum(1, np.abs(b).max(axis=2))
            print(f"depth {depth} sample {s}: pixels {diff.size}, >1e-3 rel: {int(big.sum())}, max diff {diff.max():.4g}, oracle range [{b.min():.3g}, {b.max():.3g}]")
            ys, xs = np.nonzero(big)
            for y, x in list(zip(ys, xs))[:5]:
                print("    px", x, y + y0, "gpu", a[y, x], "oracle", b[y, x])
