"""MI355X-native path-tracing hot path (drop-in for the reference's megakernel launch boundary).

The product is the C-ABI shared library ``csrc/libdmt_hip.so`` (see ``include/dmt_hip.h``) plus the
C++ host side in ``host/``.  This Python package is only the ctypes plumbing that tests and
``bench.py`` use to call through that C ABI; it contains no rendering logic and NO fallback:
if the HIP library is missing or no GPU is present, calls raise.

The directory name contains a hyphen, so import it through ``__graft_entry__.load_package()`` (or
``tests/conftest.py``), which registers it as ``cuda_optix_pathtracing_amd``.
"""
from .binding import (  # noqa: F401
    DmtError,
    Renderer,
    bvh_validate,
    light_tree_pmfs,
    light_tree_ref_select,
    envmap_tables,
    build_library,
    library_path,
    load_library,
)
from . import host_scene  # noqa: F401,E402
from . import multigpu  # noqa: F401,E402
