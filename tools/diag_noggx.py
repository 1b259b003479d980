#!/usr/bin/env python3
"""Diagnostic (GPU box): cost of the GGX branch = speed with the glass sphere swapped to Oren-Nayar."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
for swap in (False, True):
    scene = pkg.host_scene.cornell_box(1024, 1024)
    if swap:
        scene.bsdfs[1] = scene.bsdfs[0]
    with pkg.Renderer(0) as r:
        r.upload_scene(scene); r.set_limits(8)
        r.render(256); r.sync(); r.kernel_time(reset=True)
        r.film_clear(); r.render(256); r.render(256, sample_offset=256)
        ms, n = r.kernel_time(reset=True)
        print("glass->diffuse" if swap else "reference scene", f"{ms / n:.2f} ms  {1024 * 1024 * 256 / (ms / n) / 1e3:.0f} Msamples/s")
