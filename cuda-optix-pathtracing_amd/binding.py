"""ctypes binding of include/dmt_hip.h.  Plumbing only -- every call goes to libdmt_hip.so."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_CSRC = _HERE / "csrc"
_LIB = None

DMT_ACCEL_BRUTE_FORCE = 0
DMT_ACCEL_BVH = 1


class DmtError(RuntimeError):
    pass


def library_path():
    # DMT_HIP_LIB: developer knob to A/B an experimental build of the same C ABI
    import os
    alt = os.environ.get("DMT_HIP_LIB")
    return Path(alt) if alt else _CSRC / "libdmt_hip.so"


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of the HIP library, in-tree."""
    args = ["make", "-C", str(_CSRC)]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return library_path()


def load_library():
    """Load libdmt_hip.so.  Raises if it has not been built -- there is no CPU fallback."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not so.exists():
            raise DmtError(f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP path has no fallback)")
        lib = C.CDLL(str(so))
        lib.dmt_last_error.restype = C.c_char_p
        lib.dmt_last_error.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


# every symbol include/dmt_hip.h declares (checked by tests/test_abi.py against the header text)
EXPORTED_SYMBOLS = [
    "dmt_ctx_create", "dmt_ctx_destroy", "dmt_last_error", "dmt_upload_triangles", "dmt_upload_bsdfs",
    "dmt_upload_lights", "dmt_set_camera", "dmt_set_limits", "dmt_set_accel", "dmt_set_light_sampling", "dmt_light_tree_pmfs", "dmt_light_tree_ref_select", "dmt_set_bvh_strategy", "dmt_set_partition", "dmt_set_chunk", "dmt_render_profile",
    "dmt_upload_area_lights", "dmt_upload_textures", "dmt_upload_envmap", "dmt_clear_envmap", "dmt_envmap_tables", "dmt_test_envmap",
    "dmt_set_stream", "dmt_film_clear", "dmt_film_bind", "dmt_film_device_ptrs", "dmt_download_film",
    "dmt_render", "dmt_render_stats", "dmt_sync", "dmt_sched_diag", "dmt_kernel_time", "dmt_kernel_info", "dmt_bvh_validate", "dmt_test_triangle_intersect",
    "dmt_test_sampler", "dmt_test_camera_rays", "dmt_test_bsdf", "dmt_test_light", "dmt_test_half",
    "dmt_test_trace_samples", "dmt_test_trace_log", "dmt_test_closest_hit",
]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    return a.reshape(shape) if shape is not None else a


def _i32(a):
    return np.ascontiguousarray(a, np.int32).reshape(-1)


def light_tree_pmfs(lights32, p, n):
    """Host-only: selection probability of every packed light at point p with normal n; returns (pmfs, node_count, depth)."""
    lib = load_library()
    L = np.ascontiguousarray(lights32, np.uint8).reshape(-1, 32)
    out = np.zeros(L.shape[0], np.float32)
    nc, d = C.c_int(), C.c_int()
    rc = lib.dmt_light_tree_pmfs(_p(L), C.c_uint32(L.shape[0]), _p(_f32(p, (3,))), _p(_f32(n, (3,))), _p(out), C.byref(nc), C.byref(d))
    if rc != 0:
        raise DmtError(f"dmt_light_tree_pmfs failed ({rc})")
    return out, nc.value, d.value


def light_tree_ref_select(lights32, p, n, u, start_pmf=1.0):
    """Host-only: DMT_LIGHTS_TREE_REFERENCE's cut + selection (csrc/light_tree_ref.hpp ltr_select) at shading points p [k,3]
    with normals n [k,3] and one random number each -> (indices [k,4] (-1 = none), pmfs [k,4], counts [k], node_count, depth)."""
    lib = load_library()
    L = np.ascontiguousarray(lights32, np.uint8).reshape(-1, 32)
    p, n = _f32(p).reshape(-1, 3), _f32(n).reshape(-1, 3)
    u = _f32(u).reshape(-1)
    k = p.shape[0]
    idx, pmf, cnt = np.zeros((k, 4), np.int32), np.zeros((k, 4), np.float32), np.zeros(k, np.int32)
    nc, d = C.c_int(), C.c_int()
    rc = lib.dmt_light_tree_ref_select(_p(L), C.c_uint32(L.shape[0]), C.c_int(k), _p(p), _p(n), _p(u), C.c_float(start_pmf), _p(idx), _p(pmf),
                                       _p(cnt), C.byref(nc), C.byref(d))
    if rc != 0:
        raise DmtError(f"dmt_light_tree_ref_select failed ({rc})")
    return idx, pmf, cnt, nc.value, d.value


def bvh_validate(xs, ys, zs):
    """Host-only BVH build + invariant check; returns dict(node_count, depth, max_leaf, ok)."""
    lib = load_library()
    xs, ys, zs = _f32(xs), _f32(ys), _f32(zs)
    n = xs.size // 4
    nc, d, ml = C.c_int(), C.c_int(), C.c_int()
    rc = lib.dmt_bvh_validate(_p(xs), _p(ys), _p(zs), C.c_size_t(n), C.byref(nc), C.byref(d), C.byref(ml))
    return {"ok": rc == 0, "node_count": nc.value, "depth": d.value, "max_leaf": ml.value}


def envmap_tables(rgb):
    """Host-only: the PiecewiseConstant2D tables dmt_upload_envmap builds for an env map [h, w, 3]."""
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    func, cdf = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    row_int, m_func, m_cdf = np.zeros(h, np.float32), np.zeros(h, np.float32), np.zeros(h, np.float32)
    m_int = C.c_float()
    rc = lib.dmt_envmap_tables(_p(rgb), int(w), int(h), _p(func), _p(cdf), _p(row_int), _p(m_func), _p(m_cdf),
                               C.byref(m_int))
    if rc != 0:
        raise DmtError(f"dmt_envmap_tables failed ({rc})")
    return dict(func=func, cdf=cdf, row_int=row_int, m_func=m_func, m_cdf=m_cdf, m_int=np.float32(m_int.value))


class Renderer:
    """One dmt_ctx: one device, one stream.  Mirrors the reference's launch boundary
    (upload helpers + pathTraceMegakernel launches + film download)."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.dmt_ctx_create(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.dmt_last_error(None)
            raise DmtError(f"dmt_ctx_create failed ({rc}): {msg.decode() if msg else ''}")
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dmt_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.dmt_last_error(self._ctx)
            raise DmtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # ---- uploads ---------------------------------------------------------------------------
    def upload_triangles(self, xs, ys, zs, mat_id):
        xs, ys, zs = _f32(xs), _f32(ys), _f32(zs)
        mat_id = np.ascontiguousarray(mat_id, np.uint32)
        n = mat_id.shape[0]
        assert xs.size == 4 * n and ys.size == 4 * n and zs.size == 4 * n
        self._check(self._lib.dmt_upload_triangles(self._ctx, _p(xs), _p(ys), _p(zs), _p(mat_id), C.c_size_t(n)),
                    "dmt_upload_triangles")

    def upload_bsdfs(self, bsdfs):
        b = np.ascontiguousarray(bsdfs, np.uint8).reshape(-1, 32)
        self._check(self._lib.dmt_upload_bsdfs(self._ctx, _p(b), C.c_uint32(b.shape[0])), "dmt_upload_bsdfs")

    def upload_lights(self, lights, inf_lights):
        l = np.ascontiguousarray(lights, np.uint8).reshape(-1, 32)
        i = np.ascontiguousarray(inf_lights, np.uint8).reshape(-1, 32)
        self._check(self._lib.dmt_upload_lights(self._ctx, _p(l) if l.shape[0] else None, C.c_uint32(l.shape[0]),
                                                _p(i) if i.shape[0] else None, C.c_uint32(i.shape[0])),
                    "dmt_upload_lights")

    def set_camera(self, camera44):
        cam = np.ascontiguousarray(camera44, np.uint8).reshape(44)
        self._check(self._lib.dmt_set_camera(self._ctx, _p(cam)), "dmt_set_camera")
        self.width = int(cam[24:28].view(np.int32)[0])
        self.height = int(cam[28:32].view(np.int32)[0])

    def upload_scene(self, scene):
        """`scene`: any object with xs, ys, zs, mat_id, bsdfs, lights, inf_lights, camera arrays."""
        self.upload_triangles(scene.xs, scene.ys, scene.zs, scene.mat_id)
        self.upload_bsdfs(scene.bsdfs)
        self.upload_lights(scene.lights, scene.inf_lights)
        self.set_camera(scene.camera)
        if getattr(scene, "area_tri", None) is not None and len(scene.area_tri):
            self.upload_area_lights(scene.area_tri, scene.area_le)
        if getattr(scene, "env_rgb", None) is not None:
            self.upload_envmap(scene.env_rgb, scene.env_quat, scene.env_scale)
        else:
            self.clear_envmap()
        if getattr(scene, "tex_desc", None) is not None and len(scene.tex_desc):
            self.upload_textures(scene.tex_rgba, scene.tex_desc, scene.mat_tex, scene.tri_uv)
        else:
            self.upload_textures(None, None, None, None)

    def upload_textures(self, tex_rgba, tex_desc, mat_tex, tri_uv):
        """SURVEY 8f-1 image textures (layout: include/dmt_hip.h dmt_upload_textures); all None clears."""
        if tex_desc is None or len(tex_desc) == 0:
            self._check(self._lib.dmt_upload_textures(self._ctx, None, C.c_uint64(0), None, C.c_uint32(0), None, C.c_uint32(0), None,
                                                      C.c_uint64(0)), "dmt_upload_textures")
            return
        rgba = np.ascontiguousarray(tex_rgba, np.uint8).reshape(-1, 4)
        desc = np.ascontiguousarray(tex_desc, np.int32).reshape(-1, 3)
        mt = np.ascontiguousarray(mat_tex, np.uint32).reshape(-1, 4)
        uv = np.ascontiguousarray(tri_uv, np.float32).reshape(-1, 6)
        self._check(self._lib.dmt_upload_textures(self._ctx, _p(rgba), C.c_uint64(rgba.shape[0]), _p(desc), C.c_uint32(desc.shape[0]), _p(mt),
                                                  C.c_uint32(mt.shape[0]), _p(uv), C.c_uint64(uv.shape[0])), "dmt_upload_textures")

    def upload_area_lights(self, tri, le):
        tri = np.ascontiguousarray(tri, np.uint32).reshape(-1)
        le = np.ascontiguousarray(le, np.float32).reshape(-1, 3)
        assert tri.shape[0] == le.shape[0]
        self._check(self._lib.dmt_upload_area_lights(self._ctx, _p(tri), _p(le), C.c_uint32(tri.shape[0])),
                    "dmt_upload_area_lights")

    def upload_envmap(self, rgb, quat=(0, 0, 0, 1), scale=1.0):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w = rgb.shape[:2]
        q = np.ascontiguousarray(quat, np.float32)
        self._check(self._lib.dmt_upload_envmap(self._ctx, _p(rgb), int(w), int(h), _p(q), C.c_float(scale)),
                    "dmt_upload_envmap")

    def clear_envmap(self):
        self._check(self._lib.dmt_clear_envmap(self._ctx), "dmt_clear_envmap")

    def test_envmap(self, u2, wi):
        u2, wi = _f32(u2).reshape(-1, 2), _f32(wi).reshape(-1, 3)
        n = u2.shape[0]
        assert wi.shape[0] == n
        out = dict(wi=np.zeros((n, 3), np.float32), pdf=np.zeros(n, np.float32), uv=np.zeros((n, 2), np.float32),
                   Le=np.zeros((n, 3), np.float32), ok=np.zeros(n, np.int32), Le_dir=np.zeros((n, 3), np.float32),
                   pdf_dir=np.zeros(n, np.float32))
        self._check(self._lib.dmt_test_envmap(self._ctx, n, _p(u2), _p(wi), _p(out["wi"]), _p(out["pdf"]), _p(out["uv"]),
                                              _p(out["Le"]), _p(out["ok"]), _p(out["Le_dir"]), _p(out["pdf_dir"])),
                    "dmt_test_envmap")
        return out

    def set_limits(self, max_depth):
        self._check(self._lib.dmt_set_limits(self._ctx, int(max_depth)), "dmt_set_limits")

    def set_accel(self, mode):
        self._check(self._lib.dmt_set_accel(self._ctx, int(mode)), "dmt_set_accel")

    def set_light_sampling(self, mode):
        """0 = uniform pick (reference, parity mode), 1 = light tree (csrc/light_tree.hpp)."""
        self._check(self._lib.dmt_set_light_sampling(self._ctx, int(mode)), "dmt_set_light_sampling")

    def set_bvh_strategy(self, strategy, paths_per_pass=0):
        """0 = automatic, 1 = megakernel, 2 = device-side wavefront (films are bit-identical)."""
        self._check(self._lib.dmt_set_bvh_strategy(self._ctx, int(strategy), C.c_uint64(int(paths_per_pass))), "dmt_set_bvh_strategy")

    def set_partition(self, rank, world):
        self._check(self._lib.dmt_set_partition(self._ctx, int(rank), int(world)), "dmt_set_partition")

    def set_chunk(self, samples_per_item):
        self._check(self._lib.dmt_set_chunk(self._ctx, C.c_uint32(samples_per_item)), "dmt_set_chunk")

    def set_stream(self, stream_ptr):
        self._check(self._lib.dmt_set_stream(self._ctx, C.c_void_p(stream_ptr)), "dmt_set_stream")

    # ---- film ------------------------------------------------------------------------------
    def film_clear(self):
        self._check(self._lib.dmt_film_clear(self._ctx), "dmt_film_clear")

    def film_bind(self, mean_ptr, m2_ptr):
        self._check(self._lib.dmt_film_bind(self._ctx, C.c_void_p(mean_ptr), C.c_void_p(m2_ptr)), "dmt_film_bind")

    def film_device_ptrs(self):
        a, b = C.c_void_p(), C.c_void_p()
        self._check(self._lib.dmt_film_device_ptrs(self._ctx, C.byref(a), C.byref(b)), "dmt_film_device_ptrs")
        return a.value, b.value

    def download_film(self):
        mean = np.zeros((self.height, self.width, 4), np.float32)
        m2 = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self._lib.dmt_download_film(self._ctx, _p(mean), _p(m2)), "dmt_download_film")
        return mean, m2

    # ---- render ----------------------------------------------------------------------------
    def render(self, spp, sample_offset=0, region=None):
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        self._check(self._lib.dmt_render(self._ctx, C.c_uint32(sample_offset), C.c_uint32(spp), int(x0), int(y0),
                                         int(x1), int(y1)), "dmt_render")

    def render_stats(self, spp, sample_offset=0, region=None):
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        out = np.zeros(6, np.uint64)
        self._check(self._lib.dmt_render_stats(self._ctx, C.c_uint32(sample_offset), C.c_uint32(spp), int(x0), int(y0),
                                               int(x1), int(y1), _p(out)), "dmt_render_stats")
        keys = ["samples", "closest_rays", "shadow_rays", "node_visits", "tri_tests", "bounces"]
        return dict(zip(keys, (int(v) for v in out)))

    def render_profile(self, spp, sample_offset=0, region=None):
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        out = np.zeros(16, np.uint64)
        self._check(self._lib.dmt_render_profile(self._ctx, C.c_uint32(sample_offset), C.c_uint32(spp), int(x0), int(y0),
                                                 int(x1), int(y1), _p(out)), "dmt_render_profile")
        keys = ["samples", "closest_rays", "shadow_rays", "node_visits", "tri_tests", "bounces", "it_node", "it_leaf",
                "it_shade", "it_outer", "it_prep", "lanes_leaf", "lanes_shade", "lanes_prep", "dead_nodes", "overflow_pushes"]
        return dict(zip(keys, (int(v) for v in out)))

    def sync(self):
        self._check(self._lib.dmt_sync(self._ctx), "dmt_sync")

    def sched_diag(self, reset=False):
        """Counters of the in-launch fold hand-over (dmt_sched_diag); synchronises the stream."""
        out = (C.c_uint64 * 8)()
        if not hasattr(self._lib, "dmt_sched_diag"):   # an older build loaded through DMT_HIP_LIB for an A/B run
            return dict.fromkeys(["folds", "handed_over", "folded_for_others", "slab_stalls", "early_exits",
                                  "max_stall_ticks_10ns", "launched", "slabs_per_wave"], 0)
        self._check(self._lib.dmt_sched_diag(self._ctx, out, int(bool(reset))), "dmt_sched_diag")
        keys = ["folds", "handed_over", "folded_for_others", "slab_stalls", "early_exits", "max_stall_ticks_10ns", "launched", "slabs_per_wave"]
        return dict(zip(keys, [int(x) for x in out]))

    def kernel_time(self, reset=True):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.dmt_kernel_time(self._ctx, C.byref(ms), C.byref(n), int(bool(reset))), "dmt_kernel_time")
        return ms.value, n.value

    def kernel_info(self):
        v = [C.c_int() for _ in range(5)]
        self._check(self._lib.dmt_kernel_info(self._ctx, *[C.byref(x) for x in v]), "dmt_kernel_info")
        keys = ["vgprs", "sgprs", "lds_bytes", "blocks_per_cu", "cu_count"]
        return dict(zip(keys, (x.value for x in v)))

    # ---- device unit-test entry points -----------------------------------------------------
    def test_triangle_intersect(self, xs, ys, zs, o, d):
        xs, ys, zs = _f32(xs), _f32(ys), _f32(zs)
        n = xs.size // 4
        o, d = _f32(o), _f32(d)
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        pos, nrm, err = (np.zeros((n, 3), np.float32) for _ in range(3))
        self._check(self._lib.dmt_test_triangle_intersect(self._ctx, _p(xs), _p(ys), _p(zs), C.c_size_t(n), _p(o),
                                                          _p(d), _p(hit), _p(t), _p(pos), _p(nrm), _p(err)),
                    "dmt_test_triangle_intersect")
        return hit, t, pos, nrm, err

    def test_sampler(self, w, h, pxs, pys, ss, ndims):
        pxs, pys, ss = _i32(pxs), _i32(pys), _i32(ss)
        n = pxs.shape[0]
        hi = np.zeros(n, np.int32)
        p2 = np.zeros((n, 2), np.float32)
        d = np.zeros((n, ndims), np.float32)
        self._check(self._lib.dmt_test_sampler(self._ctx, int(w), int(h), n, _p(pxs), _p(pys), _p(ss), int(ndims),
                                               _p(hi), _p(p2), _p(d)), "dmt_test_sampler")
        return hi, p2, d

    def test_camera_rays(self, pxs, pys, ss):
        pxs, pys, ss = _i32(pxs), _i32(pys), _i32(ss)
        n = pxs.shape[0]
        o = np.zeros((n, 3), np.float32)
        d = np.zeros((n, 3), np.float32)
        self._check(self._lib.dmt_test_camera_rays(self._ctx, n, _p(pxs), _p(pys), _p(ss), _p(o), _p(d)),
                    "dmt_test_camera_rays")
        return o, d

    def test_bsdf(self, bsdf32, ns, wo, u2, uc, wi_eval):
        b = np.ascontiguousarray(bsdf32, np.uint8).reshape(32)
        ns, wo, wi_eval = _f32(ns, (-1, 3)), _f32(wo, (-1, 3)), _f32(wi_eval, (-1, 3))
        u2, uc = _f32(u2, (-1, 2)), _f32(uc, (-1,))
        n = ns.shape[0]
        prep = np.zeros((n, 12), np.float32)
        samp = np.zeros((n, 10), np.float32)
        ev = np.zeros((n, 4), np.float32)
        self._check(self._lib.dmt_test_bsdf(self._ctx, _p(b), n, _p(ns), _p(wo), _p(u2), _p(uc), _p(wi_eval),
                                            _p(prep), _p(samp), _p(ev)), "dmt_test_bsdf")
        return prep, samp, ev

    def test_light(self, light32, pos, nrm, u2, had_t):
        l = np.ascontiguousarray(light32, np.uint8).reshape(32)
        pos, nrm, u2 = _f32(pos, (-1, 3)), _f32(nrm, (-1, 3)), _f32(u2, (-1, 2))
        ht = _i32(had_t)
        n = pos.shape[0]
        out = np.zeros((n, 14), np.float32)
        self._check(self._lib.dmt_test_light(self._ctx, _p(l), n, _p(pos), _p(nrm), _p(u2), _p(ht), _p(out)),
                    "dmt_test_light")
        return out

    def test_half(self, floats=None, halves=None):
        h_out = f_out = None
        n = 0
        if floats is not None:
            floats = _f32(floats, (-1,))
            n = floats.shape[0]
            h_out = np.zeros(n, np.uint16)
        if halves is not None:
            halves = np.ascontiguousarray(halves, np.uint16).reshape(-1)
            n = halves.shape[0]
            f_out = np.zeros(n, np.float32)
        if floats is not None and halves is not None:
            assert floats.shape[0] == halves.shape[0]
        self._check(self._lib.dmt_test_half(self._ctx, n, _p(floats), _p(h_out), _p(halves), _p(f_out)),
                    "dmt_test_half")
        return h_out, f_out

    def test_trace_samples(self, pxs, pys, ss):
        pxs, pys, ss = _i32(pxs), _i32(pys), _i32(ss)
        n = pxs.shape[0]
        out = np.zeros((n, 3), np.float32)
        self._check(self._lib.dmt_test_trace_samples(self._ctx, n, _p(pxs), _p(pys), _p(ss), _p(out)),
                    "dmt_test_trace_samples")
        return out

    def test_trace_log(self, px, py, s, cap=64):
        rec = np.zeros((cap, 12), np.float32)
        n = C.c_int()
        L = np.zeros(3, np.float32)
        self._check(self._lib.dmt_test_trace_log(self._ctx, int(px), int(py), int(s), _p(rec), int(cap), C.byref(n),
                                                 _p(L)), "dmt_test_trace_log")
        return rec[:n.value], L

    def test_closest_hit(self, o, d):
        o, d = _f32(o, (-1, 3)), _f32(d, (-1, 3))
        n = o.shape[0]
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        self._check(self._lib.dmt_test_closest_hit(self._ctx, n, _p(o), _p(d), _p(idx), _p(t)),
                    "dmt_test_closest_hit")
        return idx, t
