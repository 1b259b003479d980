"""ctypes view of oracle/libdmt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (cuda-optix-pathtracing_amd/) never does.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


class OracleScene(C.Structure):
    _fields_ = [
        ("xs", C.c_void_p), ("ys", C.c_void_p), ("zs", C.c_void_p), ("matId", C.c_void_p),
        ("triCount", C.c_uint64),
        ("lights", C.c_void_p), ("lightCount", C.c_uint32),
        ("infLights", C.c_void_p), ("infLightCount", C.c_uint32),
        ("bsdfs", C.c_void_p), ("bsdfCount", C.c_uint32),
        ("envRgb", C.c_void_p), ("envW", C.c_int32), ("envH", C.c_int32), ("envQuat", C.c_float * 4),
        ("envScale", C.c_float),
        ("areaTri", C.c_void_p), ("areaLe", C.c_void_p), ("areaCount", C.c_uint32),
        ("texRgba", C.c_void_p), ("texDesc", C.c_void_p), ("texCount", C.c_uint32), ("matTex", C.c_void_p), ("triUv", C.c_void_p),
        ("lightSampling", C.c_int32),
    ]


def build(force=False):
    so = _HERE / "libdmt_oracle.so"
    src = _HERE / "dmt_oracle.cpp"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libdmt_oracle.so"
        if not so.exists():
            build()
        _LIB = C.CDLL(str(so))
        _LIB.oracle_sample_dim.restype = C.c_float
        _LIB.oracle_half_to_float.restype = C.c_float
        _LIB.oracle_half_to_float.argtypes = [C.c_uint16]
        _LIB.oracle_float_to_half.restype = C.c_uint16
        _LIB.oracle_float_to_half.argtypes = [C.c_float]
        _LIB.oracle_octa_from_dir.restype = C.c_uint32
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """Host arrays of one scene in the reference's upload layout (float4-per-axis SoA +
    32-byte BSDF / Light records + 44-byte DeviceCamera)."""

    def __init__(self, xs, ys, zs, mat_id, bsdfs, lights, inf_lights, camera):
        self.xs = np.ascontiguousarray(xs, np.float32).reshape(-1, 4)
        self.ys = np.ascontiguousarray(ys, np.float32).reshape(-1, 4)
        self.zs = np.ascontiguousarray(zs, np.float32).reshape(-1, 4)
        self.mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self.bsdfs = np.ascontiguousarray(bsdfs, np.uint8).reshape(-1, 32)
        self.lights = np.ascontiguousarray(lights, np.uint8).reshape(-1, 32)
        self.inf_lights = np.ascontiguousarray(inf_lights, np.uint8).reshape(-1, 32)
        self.camera = np.ascontiguousarray(camera, np.uint8).reshape(44).copy()
        self.env_rgb = None            # A18: optional env map, float32 [h, w, 3], w == 2 h, powers of two
        self.env_quat = np.array([0, 0, 0, 1], np.float32)  # lightFromRender, x y z w
        self.env_scale = 1.0
        self.area_tri, self.area_le = None, None

    def set_area_lights(self, tri, le):
        """emissive triangles (indices into the triangle arrays) and their rgb radiance."""
        self.area_tri = np.ascontiguousarray(tri, np.uint32).reshape(-1)
        self.area_le = np.ascontiguousarray(le, np.float32).reshape(-1, 3)
        assert self.area_tri.shape[0] == self.area_le.shape[0]
        return self

    def set_envmap(self, rgb, quat=(0, 0, 0, 1), scale=1.0):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w = rgb.shape[:2]
        assert rgb.shape == (h, w, 3) and w == 2 * h and h >= 8 and (h & (h - 1)) == 0
        self.env_rgb, self.env_quat, self.env_scale = rgb, np.asarray(quat, np.float32), float(scale)
        return self

    @property
    def tri_count(self):
        return int(self.mat_id.shape[0])

    # DeviceCamera fields (types.cuh:101-109): dir3 pos3 width height spp focal sensor
    def set_resolution(self, w, h):
        self.camera[24:32] = np.array([w, h], np.int32).view(np.uint8)
        return self

    @property
    def width(self):
        return int(self.camera[24:28].view(np.int32)[0])

    @property
    def height(self):
        return int(self.camera[28:32].view(np.int32)[0])

    def c_struct(self):
        s = OracleScene()
        s.xs, s.ys, s.zs, s.matId = _p(self.xs), _p(self.ys), _p(self.zs), _p(self.mat_id)
        s.triCount = self.tri_count
        s.lights, s.lightCount = _p(self.lights), self.lights.shape[0]
        s.infLights, s.infLightCount = _p(self.inf_lights), self.inf_lights.shape[0]
        s.bsdfs, s.bsdfCount = _p(self.bsdfs), self.bsdfs.shape[0]
        if self.env_rgb is not None:
            s.envRgb, s.envH, s.envW = _p(self.env_rgb), self.env_rgb.shape[0], self.env_rgb.shape[1]
            s.envQuat = (C.c_float * 4)(*[float(v) for v in self.env_quat])
            s.envScale = self.env_scale
        if self.area_tri is not None and self.area_tri.shape[0] > 0:
            s.areaTri, s.areaLe, s.areaCount = _p(self.area_tri), _p(self.area_le), self.area_tri.shape[0]
        s.lightSampling = int(getattr(self, "light_sampling", 0))
        if getattr(self, "tex_desc", None) is not None and len(self.tex_desc) > 0:
            s.texRgba, s.texDesc, s.texCount = _p(self.tex_rgba), _p(self.tex_desc), self.tex_desc.shape[0]
            s.matTex, s.triUv = _p(self.mat_tex), _p(self.tri_uv)
        return s

    def set_textures(self, tex_rgba, tex_desc, mat_tex, tri_uv):
        """RGBA8 texel atlas [T, 4], descriptors int32 [n, 3] = (first texel, width, height), per-BSDF
        uint32 [nbsdf, 4] = (diffuse, roughness, normal texture or 0xFFFFFFFF, anisotropy float bits), per-triangle
        float32 [ntri, 6] = (u0, v0, u1, v1, u2, v2)."""
        self.tex_rgba = np.ascontiguousarray(tex_rgba, np.uint8).reshape(-1, 4)
        self.tex_desc = np.ascontiguousarray(tex_desc, np.int32).reshape(-1, 3)
        self.mat_tex = np.ascontiguousarray(mat_tex, np.uint32).reshape(-1, 4)
        self.tri_uv = np.ascontiguousarray(tri_uv, np.float32).reshape(-1, 6)
        assert self.mat_tex.shape[0] == self.bsdfs.shape[0] and self.tri_uv.shape[0] == self.tri_count
        return self


def cornell_box(width=None, height=None):
    L = lib()
    nt, nb, nl, ni = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    L.oracle_cornell_box(None, None, None, None, C.byref(nt), None, C.byref(nb), None, C.byref(nl),
                         None, C.byref(ni), None)
    xs = np.zeros((nt.value, 4), np.float32)
    ys, zs = np.zeros_like(xs), np.zeros_like(xs)
    mat = np.zeros(nt.value, np.uint32)
    b = np.zeros((nb.value, 32), np.uint8)
    l = np.zeros((nl.value, 32), np.uint8)
    i = np.zeros((ni.value, 32), np.uint8)
    cam = np.zeros(44, np.uint8)
    L.oracle_cornell_box(_p(xs), _p(ys), _p(zs), _p(mat), C.byref(nt), _p(b), C.byref(nb), _p(l),
                         C.byref(nl), _p(i), C.byref(ni), _p(cam))
    sc = Scene(xs, ys, zs, mat, b, l, i, cam)
    if width is not None:
        sc.set_resolution(width, height if height is not None else width)
    return sc


def render(scene, spp, sample_offset=0, max_depth=32, region=None, threads=0, rtl_args=False,
           film=None, want_stats=False):
    """Returns (mean[h,w,4], m2[h,w,4]) (+ stats dict).  `film` continues a running film."""
    L = lib()
    w, h = scene.width, scene.height
    if film is None:
        mean = np.zeros((h, w, 4), np.float32)
        m2 = np.zeros((h, w, 4), np.float32)
    else:
        mean, m2 = film
    x0, y0, x1, y1 = region if region is not None else (0, 0, w, h)
    if threads <= 0:
        threads = os.cpu_count() or 1
    stats = np.zeros(6, np.uint64)
    cs = scene.c_struct()
    L.oracle_render(C.byref(cs), _p(scene.camera), int(max_depth), int(sample_offset), int(spp),
                    int(x0), int(y0), int(x1), int(y1), int(threads), int(bool(rtl_args)), _p(mean),
                    _p(m2), _p(stats) if want_stats else None)
    if want_stats:
        keys = ["samples", "closest_rays", "shadow_rays", "tri_tests", "bounces", "hits"]
        return mean, m2, dict(zip(keys, (int(v) for v in stats)))
    return mean, m2


def trace_samples(scene, pxs, pys, ss, max_depth=32, rtl_args=False):
    L = lib()
    pxs = np.ascontiguousarray(pxs, np.int32)
    pys = np.ascontiguousarray(pys, np.int32)
    ss = np.ascontiguousarray(ss, np.int32)
    out = np.zeros((pxs.shape[0], 3), np.float32)
    cs = scene.c_struct()
    L.oracle_trace_samples(C.byref(cs), _p(scene.camera), int(max_depth), int(pxs.shape[0]), _p(pxs),
                           _p(pys), _p(ss), int(bool(rtl_args)), _p(out))
    return out


def halton_params(w, h):
    out = np.zeros(6, np.int32)
    lib().oracle_halton_params(int(w), int(h), _p(out))
    return out


def sampler_stream(w, h, pxs, pys, ss, ndims):
    pxs = np.ascontiguousarray(pxs, np.int32)
    pys = np.ascontiguousarray(pys, np.int32)
    ss = np.ascontiguousarray(ss, np.int32)
    n = pxs.shape[0]
    hi = np.zeros(n, np.int32)
    p2 = np.zeros((n, 2), np.float32)
    d = np.zeros((n, ndims), np.float32)
    lib().oracle_sampler_stream(int(w), int(h), n, _p(pxs), _p(pys), _p(ss), int(ndims), _p(hi),
                                _p(p2), _p(d))
    return hi, p2, d


def camera_rays(scene, pxs, pys, ss):
    pxs = np.ascontiguousarray(pxs, np.int32)
    pys = np.ascontiguousarray(pys, np.int32)
    ss = np.ascontiguousarray(ss, np.int32)
    n = pxs.shape[0]
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    lib().oracle_camera_rays(_p(scene.camera), n, _p(pxs), _p(pys), _p(ss), _p(o), _p(d))
    return o, d


def triangle_intersect(xs, ys, zs, o, d):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    zs = np.ascontiguousarray(zs, np.float32)
    n = xs.size // 4
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    hit = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    pos = np.zeros((n, 3), np.float32)
    nrm = np.zeros((n, 3), np.float32)
    err = np.zeros((n, 3), np.float32)
    lib().oracle_triangle_intersect(_p(xs), _p(ys), _p(zs), C.c_uint64(n), _p(o), _p(d), _p(hit), _p(t),
                                    _p(pos), _p(nrm), _p(err))
    return hit, t, pos, nrm, err


def host_intersect_mt(xs, ys, zs, o, d):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    zs = np.ascontiguousarray(zs, np.float32)
    n = xs.size // 4
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    hit = np.zeros(n, np.int32)
    lib().oracle_host_intersect_mt(_p(xs), _p(ys), _p(zs), C.c_uint64(n), _p(o), _p(d), _p(hit))
    return hit


def closest_hit(xs, ys, zs, o, d):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    zs = np.ascontiguousarray(zs, np.float32)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    n = o.shape[0]
    idx = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    lib().oracle_closest_hit(_p(xs), _p(ys), _p(zs), C.c_uint64(xs.size // 4), n, _p(o), _p(d), _p(idx),
                             _p(t))
    return idx, t


def bsdf_cases(bsdf32, ns, wo, u2, uc, wi_eval):
    bsdf32 = np.ascontiguousarray(bsdf32, np.uint8).reshape(32)
    ns = np.ascontiguousarray(ns, np.float32).reshape(-1, 3)
    wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
    u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
    uc = np.ascontiguousarray(uc, np.float32).reshape(-1)
    wi_eval = np.ascontiguousarray(wi_eval, np.float32).reshape(-1, 3)
    n = ns.shape[0]
    prepared = np.zeros((n, 32), np.uint8)
    sample = np.zeros((n, 10), np.float32)
    ev = np.zeros((n, 4), np.float32)
    lib().oracle_bsdf_cases(_p(bsdf32), n, _p(ns), _p(wo), _p(u2), _p(uc), _p(wi_eval), _p(prepared),
                            _p(sample), _p(ev))
    return prepared, sample, ev


def light_cases(light32, pos, nrm, u2, had_transmission):
    light32 = np.ascontiguousarray(light32, np.uint8).reshape(32)
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
    u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
    ht = np.ascontiguousarray(had_transmission, np.int32).reshape(-1)
    n = pos.shape[0]
    out = np.zeros((n, 14), np.float32)
    lib().oracle_light_cases(_p(light32), n, _p(pos), _p(nrm), _p(u2), _p(ht), _p(out))
    return out


def envmap_tables(rgb):
    """A18: PiecewiseConstant2D tables of an env map -> dict(func, cdf, row_int, m_func, m_cdf, m_int)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    func, cdf = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    row_int, m_func, m_cdf = np.zeros(h, np.float32), np.zeros(h, np.float32), np.zeros(h, np.float32)
    m_int = C.c_float()
    lib().oracle_envmap_tables(_p(rgb), w, h, _p(func), _p(cdf), _p(row_int), _p(m_func), _p(m_cdf), C.byref(m_int))
    return dict(func=func, cdf=cdf, row_int=row_int, m_func=m_func, m_cdf=m_cdf, m_int=np.float32(m_int.value))


def envmap_sample(rgb, quat, u2):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
    n = u2.shape[0]
    q = np.ascontiguousarray(quat, np.float32)
    wi, pdf, uv, Le, ok = (np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32),
                           np.zeros((n, 3), np.float32), np.zeros(n, np.int32))
    lib().oracle_envmap_sample(_p(rgb), w, h, _p(q), n, _p(u2), _p(wi), _p(pdf), _p(uv), _p(Le), _p(ok))
    return dict(wi=wi, pdf=pdf, uv=uv, Le=Le, ok=ok)


def envmap_eval(rgb, quat, wi):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
    n = wi.shape[0]
    q = np.ascontiguousarray(quat, np.float32)
    Le, pdf = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib().oracle_envmap_eval(_p(rgb), w, h, _p(q), n, _p(wi), _p(Le), _p(pdf))
    return dict(Le=Le, pdf=pdf)


def pixels_from_film(mean, m2):
    mean = np.ascontiguousarray(mean, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    npix = mean.size // 4
    a = np.zeros((npix, 3), np.uint8)
    b = np.zeros((npix, 3), np.uint8)
    lib().oracle_pixels_from_film(_p(mean), _p(m2), C.c_uint64(npix), _p(a), _p(b))
    shp = mean.shape[:-1] + (3,)
    return a.reshape(shp), b.reshape(shp)


def _rec(fn, *args):
    out = np.zeros(32, np.uint8)
    fn(*args, _p(out))
    return out


def _f3(v):
    return _p(np.ascontiguousarray(v, np.float32))


def make_oren_nayar(color, roughness):
    return _rec(lib().oracle_make_oren_nayar, _f3(color), C.c_float(roughness))


def make_ggx_dielectric(rt, tt, phi0, eta, ax, ay):
    return _rec(lib().oracle_make_ggx_dielectric, _f3(rt), _f3(tt), C.c_float(phi0), C.c_float(eta),
                C.c_float(ax), C.c_float(ay))


def make_ggx_blend_dielectric(rt, tt, phi0, eta, ax, ay, metallic):
    """The dielectric half of a fractional-metallic material (this build's BS_GGX_BLEND tag = 4, the fraction as fp16 in the
    first half of the weight field); its conductor record follows it in the BSDF array."""
    rec = make_ggx_dielectric(rt, tt, phi0, eta, ax, ay).copy()
    rec[0:2] = np.array([lib().oracle_float_to_half(C.c_float(metallic))], np.uint16).view(np.uint8)
    rec[6:8] = np.array([4], np.uint16).view(np.uint8)
    return rec


def make_ggx_conductor(eta, kappa, phi0, ax, ay):
    return _rec(lib().oracle_make_ggx_conductor, _f3(eta), _f3(kappa), C.c_float(phi0), C.c_float(ax),
                C.c_float(ay))


def make_lambert():
    return _rec(lib().oracle_make_lambert)


def make_point_light(color, pos, radius):
    return _rec(lib().oracle_make_point_light, _f3(color), _f3(pos), C.c_float(radius))


def make_spot_light(color, pos, direction, cos0, cos_e, radius):
    return _rec(lib().oracle_make_spot_light, _f3(color), _f3(pos), _f3(direction), C.c_float(cos0),
                C.c_float(cos_e), C.c_float(radius))


def make_directional_light(color, direction, one_minus_cos):
    return _rec(lib().oracle_make_directional_light, _f3(color), _f3(direction), C.c_float(one_minus_cos))


def make_env_light(color):
    return _rec(lib().oracle_make_env_light, _f3(color))


def trace_log(scene, px, py, s, max_depth=32, rtl_args=False, cap=64):
    rec = np.zeros((cap, 12), np.float32)
    L = np.zeros(3, np.float32)
    cs = scene.c_struct()
    n = lib().oracle_trace_log(C.byref(cs), _p(scene.camera), int(max_depth), int(px), int(py), int(s),
                               int(bool(rtl_args)), _p(rec), int(cap), _p(L))
    return rec[:n], L


def light_tree_pmfs(lights32, p, n):
    """selection probability of every packed light at point p with normal n under the oracle's own tree."""
    L = np.ascontiguousarray(lights32, np.uint8).reshape(-1, 32)
    out = np.zeros(L.shape[0], np.float32)
    nc = C.c_int()
    lib().oracle_light_tree_pmfs(_p(L), C.c_uint32(L.shape[0]), _p(np.ascontiguousarray(p, np.float32)),
                                 _p(np.ascontiguousarray(n, np.float32)), _p(out), C.byref(nc))
    return out, nc.value



def light_tree_ref_select(lights32, p, n, u, start_pmf=1.0):
    """the reference-semantics tree (cones, SAOH, cuts of up to four nodes) at shading points p [k,3] with
    normals n [k,3] and one random number each: (indices [k,4] (-1 = none), pmfs [k,4], counts [k], node count)."""
    L = np.ascontiguousarray(lights32, np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
    n = np.ascontiguousarray(n, np.float32).reshape(-1, 3)
    u = np.ascontiguousarray(u, np.float32).reshape(-1)
    k = p.shape[0]
    idx = np.zeros((k, 4), np.int32)
    pmf = np.zeros((k, 4), np.float32)
    cnt = np.zeros(k, np.int32)
    nc = C.c_int()
    rc = lib().oracle_light_tree_ref_select(_p(L), C.c_uint32(L.shape[0]), C.c_int(k), _p(p), _p(n), _p(u), C.c_float(start_pmf),
                                            _p(idx), _p(pmf), _p(cnt), C.byref(nc))
    if rc != 0:
        raise ValueError("oracle_light_tree_ref_select: no point / spot lights")
    return idx, pmf, cnt, nc.value
