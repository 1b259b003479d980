"""ctypes view of host/libdmt_host.so: the C++ host side above the C ABI (scene builders, packers,
writers).  Plumbing only."""
import ctypes as C
from pathlib import Path

import numpy as np

_HOST = Path(__file__).resolve().parent / "host"
_LIB = None


def load_host_library():
    global _LIB
    if _LIB is None:
        so = _HOST / "libdmt_host.so"
        if not so.exists():
            raise RuntimeError(f"{so} is missing: run __graft_entry__.build()")
        lib = C.CDLL(str(so))
        for name in ("dmt_host_scene_cornell_box", "dmt_host_scene_random_triangles", "dmt_host_scene_random_triangles_ex", "dmt_host_scene_xs",
                     "dmt_host_scene_ys", "dmt_host_scene_zs", "dmt_host_scene_mat_ids", "dmt_host_scene_bsdfs",
                     "dmt_host_scene_lights", "dmt_host_scene_infinite_lights", "dmt_host_scene_camera",
                     "dmt_host_scene_load_json", "dmt_host_scene_env_rgb", "dmt_host_scene_load_pbrt",
                     "dmt_host_scene_area_tri", "dmt_host_scene_area_le"):
            getattr(lib, name).restype = C.c_void_p
        lib.dmt_host_scene_random_triangles.argtypes = [C.c_uint64, C.c_uint64]
        lib.dmt_host_scene_random_triangles_ex.argtypes = [C.c_uint64, C.c_uint64, C.c_float]
        lib.dmt_host_scene_triangle_count.restype = C.c_uint64
        for name in ("dmt_host_scene_bsdf_count", "dmt_host_scene_light_count", "dmt_host_scene_infinite_light_count"):
            getattr(lib, name).restype = C.c_uint32
        lib.dmt_host_half_bits_to_float.restype = C.c_float
        lib.dmt_host_half_bits_to_float.argtypes = [C.c_uint16]
        lib.dmt_host_float_to_half_bits.restype = C.c_uint16
        lib.dmt_host_float_to_half_bits.argtypes = [C.c_float]
        _LIB = lib
    return _LIB


def _copy(ptr, dtype, count):
    if count == 0:
        return np.zeros(0, dtype)
    n = count * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class HostScene:
    """Arrays of one scene in the upload layout (same attribute names the Renderer.upload_scene uses)."""

    def __init__(self, handle):
        L = load_host_library()
        h = C.c_void_p(handle)
        n = L.dmt_host_scene_triangle_count(h)
        self.xs = _copy(L.dmt_host_scene_xs(h), np.float32, 4 * n).reshape(-1, 4)
        self.ys = _copy(L.dmt_host_scene_ys(h), np.float32, 4 * n).reshape(-1, 4)
        self.zs = _copy(L.dmt_host_scene_zs(h), np.float32, 4 * n).reshape(-1, 4)
        self.mat_id = _copy(L.dmt_host_scene_mat_ids(h), np.uint32, n)
        self.bsdfs = _copy(L.dmt_host_scene_bsdfs(h), np.uint8, 32 * L.dmt_host_scene_bsdf_count(h)).reshape(-1, 32)
        self.lights = _copy(L.dmt_host_scene_lights(h), np.uint8, 32 * L.dmt_host_scene_light_count(h)).reshape(-1, 32)
        self.inf_lights = _copy(L.dmt_host_scene_infinite_lights(h), np.uint8,
                                32 * L.dmt_host_scene_infinite_light_count(h)).reshape(-1, 32)
        self.camera = _copy(L.dmt_host_scene_camera(h), np.uint8, 44)
        ew, eh = C.c_int(), C.c_int()
        env = L.dmt_host_scene_env_rgb(h, C.byref(ew), C.byref(eh))
        self.env_rgb = _copy(env, np.float32, 3 * ew.value * eh.value).reshape(eh.value, ew.value, 3) if env else None
        self.env_quat, self.env_scale = np.array([0, 0, 0, 1], np.float32), 1.0
        self.max_depth, self.spp = None, None
        L.dmt_host_scene_area_light_count.restype = C.c_uint32
        na = L.dmt_host_scene_area_light_count(h)
        self.area_tri = _copy(L.dmt_host_scene_area_tri(h), np.uint32, na)
        self.area_le = _copy(L.dmt_host_scene_area_le(h), np.float32, 3 * na).reshape(-1, 3)
        L.dmt_host_scene_texture_count.restype = C.c_uint32
        L.dmt_host_scene_texel_count.restype = C.c_uint64
        nt = L.dmt_host_scene_texture_count(h)
        self.tex_desc = self.tex_rgba = self.mat_tex = self.tri_uv = None
        if nt:
            for name in ("dmt_host_scene_tex_rgba", "dmt_host_scene_tex_desc", "dmt_host_scene_mat_tex", "dmt_host_scene_tri_uv"):
                getattr(L, name).restype = C.c_void_p
            self.tex_rgba = _copy(L.dmt_host_scene_tex_rgba(h), np.uint8, 4 * L.dmt_host_scene_texel_count(h)).reshape(-1, 4)
            self.tex_desc = _copy(L.dmt_host_scene_tex_desc(h), np.int32, 3 * nt).reshape(-1, 3)
            self.mat_tex = _copy(L.dmt_host_scene_mat_tex(h), np.uint32, 4 * self.bsdfs.shape[0]).reshape(-1, 4)
            self.tri_uv = _copy(L.dmt_host_scene_tri_uv(h), np.float32, 6 * n).reshape(-1, 6)
        L.dmt_host_scene_destroy(h)

    @property
    def tri_count(self):
        return int(self.mat_id.shape[0])

    @property
    def width(self):
        return int(self.camera[24:28].view(np.int32)[0])

    @property
    def height(self):
        return int(self.camera[28:32].view(np.int32)[0])

    def set_resolution(self, w, h):
        self.camera[24:32] = np.array([w, h], np.int32).view(np.uint8)
        return self


def read_fbx(path):
    """First mesh of a binary FBX file -> float32 [n, 3, 3] triangle list (host/dmt_fbx.cpp)."""
    L = load_host_library()
    L.dmt_host_read_fbx.restype = C.c_int64
    err = C.create_string_buffer(1024)
    n = L.dmt_host_read_fbx(str(path).encode(), None, C.c_uint64(0), err, C.c_uint64(len(err)))
    if n < 0:
        raise ValueError(err.value.decode())
    out = np.zeros((n, 3, 3), np.float32)
    L.dmt_host_read_fbx(str(path).encode(), out.ctypes.data_as(C.c_void_p), C.c_uint64(n), err, C.c_uint64(len(err)))
    return out


def load_json(path):
    """The reference's JSON scene description (core-parser.cpp) -> HostScene (+ .max_depth, .spp, .env_rgb).
    Raises ValueError with the loader's message when the file is rejected."""
    L = load_host_library()
    md, spp = C.c_int(), C.c_int()
    err = C.create_string_buffer(1024)
    h = L.dmt_host_scene_load_json(str(path).encode(), C.byref(md), C.byref(spp), err, C.c_uint64(len(err)))
    if not h:
        raise ValueError(err.value.decode() or "dmt_host_scene_load_json failed")
    s = HostScene(h)
    s.max_depth, s.spp = md.value, spp.value
    return s


def load_pbrt(path):
    """PBRT-v4 subset (scenes/cornell-box.pbrt's directives) -> HostScene (+ .max_depth, .spp, .area_tri, .area_le)."""
    L = load_host_library()
    md, spp = C.c_int(), C.c_int()
    err = C.create_string_buffer(1024)
    h = L.dmt_host_scene_load_pbrt(str(path).encode(), C.byref(md), C.byref(spp), err, C.c_uint64(len(err)))
    if not h:
        raise ValueError(err.value.decode() or "dmt_host_scene_load_pbrt failed")
    s = HostScene(h)
    s.max_depth, s.spp = md.value, spp.value
    return s


def cornell_box(width=None, height=None):
    s = HostScene(load_host_library().dmt_host_scene_cornell_box())
    if width is not None:
        s.set_resolution(width, height if height is not None else width)
    return s


def random_triangle_scene(count, seed=0x5EED1234, width=None, height=None, extent=1.0):
    """SURVEY 8(d) generator; extent scales the cube of centroids (density = count / extent^3 of the recipe's)."""
    L = load_host_library()
    s = HostScene(L.dmt_host_scene_random_triangles(int(count), int(seed)) if extent == 1.0 else
                  L.dmt_host_scene_random_triangles_ex(int(count), int(seed), float(extent)))
    if width is not None:
        s.set_resolution(width, height if height is not None else width)
    return s


def film_to_rgb8(mean, m2):
    mean = np.ascontiguousarray(mean, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    n = mean.size // 4
    a = np.zeros((n, 3), np.uint8)
    b = np.zeros((n, 3), np.uint8)
    load_host_library().dmt_host_film_to_rgb8(mean.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p),
                                              C.c_uint64(n), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    shp = mean.shape[:-1] + (3,)
    return a.reshape(shp), b.reshape(shp)


def write_mean_and_mse(mean, m2, base_name):
    mean = np.ascontiguousarray(mean, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    h, w = mean.shape[:2]
    rc = load_host_library().dmt_host_write_mean_and_mse(mean.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p),
                                                         C.c_uint32(w), C.c_uint32(h), str(base_name).encode())
    if rc != 0:
        raise RuntimeError(f"writing {base_name}.png failed")


class ArrayScene:
    """A scene assembled from numpy arrays (same attributes as HostScene)."""

    def __init__(self, xs, ys, zs, mat_id, bsdfs, lights, inf_lights, camera, env_rgb=None):
        self.xs, self.ys, self.zs = (np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (xs, ys, zs))
        self.mat_id = np.ascontiguousarray(mat_id, np.uint32)
        self.bsdfs = np.ascontiguousarray(bsdfs, np.uint8).reshape(-1, 32)
        self.lights = np.ascontiguousarray(lights, np.uint8).reshape(-1, 32)
        self.inf_lights = np.ascontiguousarray(inf_lights, np.uint8).reshape(-1, 32)
        self.camera = np.ascontiguousarray(camera, np.uint8).reshape(44).copy()
        self.env_rgb = None if env_rgb is None else np.ascontiguousarray(env_rgb, np.float32)
        self.env_quat, self.env_scale = np.array([0, 0, 0, 1], np.float32), 1.0

    tri_count = HostScene.tri_count
    width = HostScene.width
    height = HostScene.height
    set_resolution = HostScene.set_resolution


def _record(fn, *args):
    out = np.zeros(32, np.uint8)
    fn(*args, out.ctypes.data_as(C.c_void_p))
    return out


def _f3(v):
    return np.ascontiguousarray(v, np.float32).ctypes.data_as(C.c_void_p)


def synthetic_sky(height=512):
    """Deterministic HDR equirectangular map (height x 2*height x 3): horizon gradient + a small, very bright sun."""
    h, w = int(height), 2 * int(height)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    v = y / h
    sky = np.stack([0.35 + 0.4 * v, 0.45 + 0.35 * v, 0.7 + 0.2 * v], -1).astype(np.float32)
    sun = np.exp(-(((x - 0.3 * w) / (0.01 * w)) ** 2 + ((y - 0.62 * h) / (0.02 * h)) ** 2)).astype(np.float32)
    return (sky + sun[..., None] * np.array([900.0, 800.0, 600.0], np.float32)).astype(np.float32)


def sphere_envmap_scene(width, height, lat=64, lon=128, env_height=512):
    """BASELINE config 3 with synthetic assets (the reference's sphere.fbx / veranda map do not travel): a UV sphere
    (conductor) on a large Oren-Nayar ground plane under an importance-sampled HDR sky, one spot light."""
    L = load_host_library()
    th = np.pi * np.arange(lat + 1, dtype=np.float64) / lat
    ph = 2 * np.pi * np.arange(lon + 1, dtype=np.float64) / lon
    P = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)), np.outer(np.cos(th), np.ones_like(ph))], -1)
    P = (P * 1.0 + np.array([0.0, 4.0, 0.0])).astype(np.float32)
    tris = []
    for i in range(lat):
        for j in range(lon):
            a, b, c, d = P[i, j], P[i + 1, j], P[i + 1, j + 1], P[i, j + 1]
            if i > 0:
                tris.append((a, b, d))
            if i < lat - 1:
                tris.append((b, c, d))
    g = 40.0
    q = np.array([[-g, 4 - g, -1.0], [g, 4 - g, -1.0], [-g, 4 + g, -1.0], [g, 4 + g, -1.0]], np.float32)
    tris += [(q[0], q[3], q[2]), (q[0], q[1], q[3])]
    T = np.array(tris, np.float32)                      # [n, 3 vertices, xyz]
    n = T.shape[0]
    xs, ys, zs = (np.concatenate([T[:, :, k], np.zeros((n, 1), np.float32)], 1) for k in range(3))
    mat = np.zeros(n, np.uint32)
    mat[-2:] = 1
    bsdfs = np.stack([_record(L.dmt_host_make_ggx_conductor, _f3([0.18299, 0.42108, 1.37340]), _f3([3.42420, 2.34590, 1.77040]),
                              C.c_float(0.0), C.c_float(0.2), C.c_float(0.15)),
                      _record(L.dmt_host_make_oren_nayar, _f3([0.55, 0.5, 0.45]), C.c_float(0.8))])
    lights = np.stack([_record(L.dmt_host_make_spot_light, _f3([30, 30, 28]), _f3([-2.0, 1.5, 3.0]), _f3([0.5, 0.65, -0.6]),
                               C.c_float(np.cos(np.radians(25))), C.c_float(np.cos(np.radians(35))), C.c_float(1e-3))])
    cam = np.zeros(44, np.uint8)
    cam[:24] = np.array([0, 1, -0.05, 0, 0, 0], np.float32).view(np.uint8)
    cam[24:36] = np.array([width, height, 1], np.int32).view(np.uint8)
    cam[36:44] = np.array([28.0, 36.0], np.float32).view(np.uint8)
    return ArrayScene(xs, ys, zs, mat, bsdfs, lights, np.zeros((0, 32), np.uint8), cam, synthetic_sky(env_height))
