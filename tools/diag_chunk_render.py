#!/usr/bin/env python3
"""Diagnostic (GPU box): one Cornell 1024^2 x 256 spp render (after one untimed one) with a given samples-per-item chunk;
prints the kernel time.  Run under rocprofv3 --pmc by tools/pmc_chunk_sweep.sh to get the fabric traffic per chunk size."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
pkg = g.load_package()
chunk = int(sys.argv[1])
res, spp = 1024, 256
with pkg.Renderer(0) as r:
    r.upload_scene(pkg.host_scene.cornell_box(res, res))
    r.set_limits(8)
    r.set_chunk(chunk)
    r.film_clear(); r.render(spp); r.sync(); r.kernel_time(reset=True)
    r.film_clear(); r.render(spp); r.sync()
    ms, n = r.kernel_time(reset=True)
    print(f"chunk {chunk}: {ms / n:.3f} ms per launch, {res * res * spp / (ms / n) / 1e3:.1f} Msamples/s")
