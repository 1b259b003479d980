#!/usr/bin/env python3
"""Diagnostic (GPU box): per-launch overhead of small dmt_render calls (the reference CLI's default is kspp = 4)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.host_scene.cornell_box(256, 256)
with pkg.Renderer(0) as r:
    r.upload_scene(scene); r.set_limits(32)
    for kspp in (4, 16, 64):
        n = 256 // kspp * 4
        r.film_clear(); r.render(kspp); r.sync(); r.kernel_time(reset=True)
        t0 = time.perf_counter()
        for i in range(n):
            r.render(kspp, sample_offset=i * kspp); r.sync()
        wall = (time.perf_counter() - t0) * 1e3
        ms, cnt = r.kernel_time(reset=True)
        t0 = time.perf_counter()
        for i in range(n):
            r.render(kspp, sample_offset=(n + i) * kspp)
        r.sync()
        wall2 = (time.perf_counter() - t0) * 1e3
        r.kernel_time(reset=True)
        print(f"kspp {kspp:3d}: {n} launches  wall+sync each {wall / n * 1e3:7.1f} us/launch   kernel {ms / cnt * 1e3:7.1f} us/launch   back-to-back {wall2 / n * 1e3:7.1f} us/launch")
