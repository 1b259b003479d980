#!/usr/bin/env python3
"""Diagnostic (GPU box): fractional-metallic blend, GPU vs oracle per pixel at 1 spp (= per-sample radiance)."""
import json, shutil, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
src = ROOT / "tests" / "golden" / "json_scene"
tmp = Path(tempfile.mkdtemp())
d = json.loads((src / "three_boxes.json").read_text())
gold = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
glass = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
d["materials"][1]["metallic"] = gold; d["materials"][0]["metallic"] = glass
shutil.copy(src / "sky_32x16.png", tmp / "sky_32x16.png")
(tmp / "b.json").write_text(json.dumps(d))
hs = pkg.host_scene.load_json(tmp / "b.json")
osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera); osc.set_envmap(hs.env_rgb)
with pkg.Renderer(0) as r:
    r.upload_scene(hs); r.set_limits(hs.max_depth)
    for depth in (1, 2, hs.max_depth):
        r.set_limits(depth)
        for s in range(3):
            r.film_clear(); r.render(1, sample_offset=s); mean, _ = r.download_film()
            om = O.render(osc, 1, sample_offset=s, max_depth=depth, threads=8)[0]
            diff = np.abs(mean[..., :3] - om[..., :3]).max(axis=2)
            big = diff > 1e-3 * np.maximum(1, np.abs(om[..., :3]).max(axis=2))
            print(f"depth {depth} sample {s}: pixels {diff.size}, >1e-3 rel: {int(big.sum())}, max diff {diff.max():.4g}, median {np.median(diff):.3g}, oracle range [{om[...,:3].min():.3g}, {om[...,:3].max():.3g}]")
            ys, xs = np.nonzero(big)
            for y, x in list(zip(ys, xs))[:4]:
                print("    px", x, y, "gpu", mean[y, x, :3], "oracle", om[y, x, :3])
    r.set_limits(hs.max_depth)
    r.film_clear(); r.render(32); mean, m2 = r.download_film()
    om, om2 = O.render(osc, 32, max_depth=hs.max_depth, threads=8)[:2]
    rms = np.sqrt(om[..., :3].astype(np.float64) ** 2 + om2[..., :3] / 32.0)
    rel = np.abs(mean[..., :3] - om[..., :3]) / np.maximum(1.0, rms)
    print("32 spp: quantiles of |diff| / max(1, sample rms):", [float(np.quantile(rel, q)) for q in (0.5, 0.9, 0.99, 0.999)], "max", float(rel.max()),
          "share > 1e-3:", float((rel > 1e-3).mean()), "share > 1e-2:", float((rel > 1e-2).mean()))
    ys, xs, cs = np.nonzero(rel > 1e-3)
    for y, x, c in list(zip(ys, xs, cs))[:6]:
        print("   px", x, y, c, "gpu", mean[y, x, c], "oracle", om[y, x, c], "rms", rms[y, x, c])
