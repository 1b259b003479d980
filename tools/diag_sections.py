#!/usr/bin/env python3
"""Diagnostic (GPU box): where the brute-force megakernel's cycles go.

Needs the timing build: make -C cuda-optix-pathtracing_amd/csrc variant NAME=sect DEFS=-DDMT_SECTION_TIMING=1, then
DMT_HIP_LIB=.../variants/libdmt_hip_sect.so python tools/diag_sections.py [res] [spp] [depth].  Prints the share of wave
cycles per section of the loop (the cycles of a section include the time the wave waited for its SIMD's other waves).
"""
import ctypes
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
NAMES = ["draw/prepare/begin", "retire/fold/check", "triangle loop", "shadow resolve", "miss / hit record", "bsdf prepare",
         "light sample", "NEE bsdf eval + weight", "bsdf sample", "bounce tail (offset, beta, RR)", "park / stage", "", "", "", "",
         "start-up"]
scene = pkg.host_scene.cornell_box(res, res)
if len(sys.argv) > 4 and sys.argv[4] == "all-diffuse":  # every material becomes the first Oren-Nayar record: no material divergence
    scene.bsdfs[:] = scene.bsdfs[0]
lib = pkg.binding.load_library()
out = (ctypes.c_ulonglong * 16)()
with pkg.Renderer(0) as r:
    r.upload_scene(scene)
    r.set_limits(depth)
    r.film_clear(); r.render(spp); r.sync()
    assert lib.dmt_diag_section_cycles(out, 1) == 0
    r.film_clear(); r.render(spp); r.sync()
    ms, n = r.kernel_time(reset=True)
    assert lib.dmt_diag_section_cycles(out, 1) == 0
    tot = float(sum(out))
    print(f"{res}x{res} x {spp} spp, depth {depth}: {ms / n:.2f} ms (timing build)")
    counts = [int(out[k]) for k in (11, 12, 13, 14)]
    for k in (11, 12, 13, 14):
        out[k] = 0
    tot = float(sum(out))
    if counts[2]:
        print(f"  shading passes {counts[2]}, with a GGX lane {counts[0]} ({100.0 * counts[0] / counts[2]:.1f} %), lanes per pass "
              f"{counts[3] / counts[2]:.1f}, GGX lanes per pass {counts[1] / counts[2]:.2f}")
    for k, name in enumerate(NAMES):
        if out[k]:
            print(f"  {name:34s} {100.0 * out[k] / tot:6.2f} %   {out[k] / 1e9:9.3f} G wave-cycles")
