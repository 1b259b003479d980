#!/usr/bin/env python3
"""Diagnostic (GPU box): SHA-256 of the film of a few fixed renders, one line per scene.

For kernel work that must not change a single bit: run before and after (or with DMT_HIP_LIB pointing at a variant build) and
diff the output.  Scenes: Cornell (brute force and BVH), the env-map sphere, the PBRT Cornell box with emissive triangles.
"""
import hashlib
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
hs = pkg.host_scene


def film_hash(scene, spp, depth, accel):
    with pkg.Renderer(0) as r:
        r.upload_scene(scene)
        r.set_limits(depth)
        r.set_accel(1 if accel == "bvh" else 0)
        r.film_clear()
        r.render(spp)
        mean, m2 = r.download_film()
    h = hashlib.sha256()
    h.update(mean.tobytes()), h.update(m2.tobytes())
    return h.hexdigest()[:24], float(mean[..., :3].mean())


cases = [("cornell 256x256 64spp depth 8 brute", hs.cornell_box(256, 256), 64, 8, "brute"),
         ("cornell 200x136 33spp depth 32 brute", hs.cornell_box(200, 136), 33, 32, "brute"),
         ("cornell 256x256 16spp depth 8 bvh", hs.cornell_box(256, 256), 16, 8, "bvh"),
         ("env sphere 128x128 16spp depth 6 brute", hs.sphere_envmap_scene(128, 128, lat=8, lon=16, env_height=64), 16, 6, "brute"),
         ("env sphere 128x128 16spp depth 6 bvh", hs.sphere_envmap_scene(128, 128, lat=8, lon=16, env_height=64), 16, 6, "bvh")]
pbrt = ROOT / "tests" / "golden" / "pbrt" / "cornell_box.pbrt"
if pbrt.exists():
    cases.append(("pbrt cornell (emissive) 16spp depth 6 brute", hs.load_pbrt(str(pbrt)).set_resolution(128, 128), 16, 6, "brute"))
for name, scene, spp, depth, accel in cases:
    digest, mean = film_hash(scene, spp, depth, accel)
    print(f"{digest}  mean {mean:.9f}  {name}")
