"""Test oracle for the JSON scene front-end: an independent restatement, in numpy float32, of how the
reference's parser turns a scene description into geometry, lights and camera (src/core/private/core-parser.cpp:256-1455,
core-trianglemesh.cpp:101-188, core-light.cpp:14-80, cudautils-transform.cu:28-85 = glm translate / rotate / scale),
followed by the flattening this build documents in DESIGN.md (one packed BSDF record per material, vertices transformed
on the host).  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.  Records are packed with the oracle's
own packers (oracle_py.make_*), so the product's C++ packers are checked too.  No reference-side vectors exist for
this path: parity unpinned beyond this restatement."""
import json
import math
from pathlib import Path

import numpy as np

F = np.float32


def _mat_identity():
    return np.eye(4, dtype=F)        # indexed [row, col]; glm's m[col][row]


def _mul(a, b):
    r = np.zeros((4, 4), F)
    for c in range(4):
        for row in range(4):
            r[row, c] = F(F(F(a[row, 0] * b[0, c]) + F(a[row, 1] * b[1, c])) + F(a[row, 2] * b[2, c])) + F(a[row, 3] * b[3, c])
    return r


def _normalize(v):
    v = np.asarray(v, F)
    inv = F(1.0) / np.sqrt(F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2])), dtype=F)
    return np.array([v[0] * inv, v[1] * inv, v[2] * inv], F)


def _translate(t):
    m = _mat_identity()
    m[0, 3], m[1, 3], m[2, 3] = F(t[0]), F(t[1]), F(t[2])
    return m


def _scale(s):
    m = _mat_identity()
    m[0, 0], m[1, 1], m[2, 2] = F(s[0]), F(s[1]), F(s[2])
    return m


def _rotate(deg, axis):
    a = F(F(deg) * F(0.01745329251994329576923690768489))
    c, s = F(math.cos(float(a))), F(math.sin(float(a)))     # cosf/sinf up to the last ulp (tolerance in the test)
    ax = _normalize(axis)
    t = np.array([F(F(1) - c) * ax[0], F(F(1) - c) * ax[1], F(F(1) - c) * ax[2]], F)
    m = _mat_identity()
    m[0, 0], m[1, 0], m[2, 0] = c + t[0] * ax[0], t[0] * ax[1] + s * ax[2], t[0] * ax[2] - s * ax[1]
    m[0, 1], m[1, 1], m[2, 1] = t[1] * ax[0] - s * ax[2], c + t[1] * ax[1], t[1] * ax[2] + s * ax[0]
    m[0, 2], m[1, 2], m[2, 2] = t[2] * ax[0] + s * ax[1], t[2] * ax[1] - s * ax[0], c + t[2] * ax[2]
    return m


def _point(M, p):
    return np.array([F(F(F(M[r, 0] * p[0]) + F(M[r, 1] * p[1])) + F(M[r, 2] * p[2])) + M[r, 3] for r in range(3)], F)


def _vector(M, v):
    return np.array([F(F(M[r, 0] * v[0]) + F(M[r, 1] * v[1])) + F(M[r, 2] * v[2]) for r in range(3)], F)


_H = F(0.5)
_CUBE_P = np.array([[-_H, -_H, _H], [-_H, -_H, -_H], [_H, -_H, -_H], [_H, -_H, _H], [-_H, _H, _H], [-_H, _H, -_H],
                    [_H, _H, -_H], [_H, _H, _H]], F)
_CUBE_T = [(1, 4, 5), (1, 0, 4), (2, 0, 1), (2, 3, 0), (6, 3, 2), (6, 7, 3), (0, 3, 4), (4, 3, 7), (7, 5, 4), (7, 6, 5),
           (1, 5, 6), (2, 1, 6)]                      # core-trianglemesh.cpp:155-172, position indices
_PLANE_P = np.array([[-_H, -_H, 0], [_H, -_H, 0], [-_H, _H, 0], [_H, _H, 0]], F)
_PLANE_T = [(0, 3, 2), (0, 1, 3)]                      # :186-187


def _clamp(x, lo, hi):
    return min(max(x, lo), hi)


def load(path, O):
    """-> dict(xs, ys, zs [n,4] float32, mat_id, bsdfs [m,32] u8, lights [k,32] u8, camera dict, max_depth, spp)."""
    path = Path(path)
    data = json.loads(path.read_text())
    cam = data["camera"]
    out = dict(max_depth=int(cam.get("max-depth", 5)), spp=int(data["film"].get("samples", 1)))
    out["camera"] = dict(dir=np.array(cam["direction"], F), pos=np.array(cam.get("position", [0, 0, 0]), F),
                         width=int(data["film"]["resolutionX"]), height=int(data["film"]["resolutionY"]),
                         focal=F(cam["focalLength"]), sensor=F(cam["sensorSize"]))
    # materials -> one packed record each
    mats, bsdfs = {}, []
    for m in data["materials"]:
        mats[m["name"]] = len(bsdfs)
        rough = F(_clamp(float(F(m["roughness"])), 0.0, 1.0))
        metal = F(_clamp(float(F(m["metallic"])), 0.0, 1.0))
        if "oren-nayar-dielectric" in m:
            d = [float(F(int(F(_clamp(float(F(v)), 0.0, 1.0)) * F(255))) / F(255)) for v in m["diffuse"]]   # byte3FromRGB
            bsdfs.append(O.make_oren_nayar(d, float(rough)))
            continue
        aniso = F(1)
        if "ggx-anisotropy" in m:
            t = F(_clamp(float(F(m["ggx-anisotropy"])), 0.0, 1.0))
            aniso = F(1) if t == 0 else F(F(F(8) - F(1)) * t) + F(1)
        ax, ay = F(aniso * rough), rough
        eta = [max(float(F(v)), 0.0) for v in m.get("eta", [0.18299, 0.42108, 1.37340])]
        etak = [max(float(F(v)), 0.0) for v in m.get("etak", [3.42420, 2.34590, 1.77040])]
        g = m.get("ggx-dielectric") or {}
        rt = [_clamp(float(F(v)), 0.0, 1.0) for v in g.get("reflectance-tint", [1, 1, 1])]
        tt = [_clamp(float(F(v)), 0.0, 1.0) for v in g.get("transmittance-tint", [1, 1, 1])]
        ior = max(float(F(m.get("ior", 1.4))), 1.0)
        # core-material.cpp:272-286: <= 0 the dielectric, >= 1 the conductor, in between BOTH records (blended per hit)
        if metal >= F(1):
            bsdfs.append(O.make_ggx_conductor(eta, etak, 0.0, float(ax), float(ay)))
        elif metal > F(0):
            bsdfs.append(O.make_ggx_blend_dielectric(rt, tt, 0.0, ior, float(ax), float(ay), float(metal)))
            bsdfs.append(O.make_ggx_conductor(eta, etak, 0.0, float(ax), float(ay)))
        else:
            bsdfs.append(O.make_ggx_dielectric(rt, tt, 0.0, ior, float(ax), float(ay)))
    objects = {}
    for o in data["objects"]:
        P, T = (_CUBE_P, _CUBE_T) if o["shape"] == "cube" else (_PLANE_P, _PLANE_T)
        objects[o["name"]] = (P, T, mats[o["material"]])
    lights = {}
    for l in data["lights"]:
        if l["type"] == "point":
            lights[l["name"]] = ("point", [float(F(v)) for v in l.get("radiant-intensity", [1, 1, 1])], None, None)
        else:
            cone = F(_clamp(float(F(l.get("cone-angle", 60.0))), 10.0, 120.0))
            fall = F(_clamp(float(F(l.get("falloff-percentage", 10.0))), 1.0, 80.0))
            pi = F(3.14159265358979323846)
            a0 = F(F(F(cone * F(F(1) - F(fall / F(100)))) * pi) / F(180))
            ae = F(F(cone * pi) / F(180))
            c0, ce = F(math.cos(float(a0))), F(math.cos(float(ae)))
            c0 = max(c0, ce)
            ce = min(c0, ce)
            lights[l["name"]] = ("spot", [max(float(F(v)), 0.0) for v in l.get("radiant-intensity", [1, 1, 1])], float(c0), float(ce))
    transforms = {}
    for t in data["transforms"]:
        srt = t.get("srt", {})
        axis = _normalize(srt["rotate-axis"]) if "rotate-axis" in srt else np.array([0, 0, 1], F)
        s = srt.get("scale", [1, 1, 1])
        s = [s, s, s] if not isinstance(s, list) else s
        transforms[t["name"]] = _mul(_mul(_translate(srt.get("translation-vector", [0, 0, 0])),
                                          _rotate(srt.get("rotate-degrees", 0.0), axis)), _scale(s))
    tris, mat_id, recs = [], [], []

    def walk(node, stack):
        for key in sorted(node):                      # nlohmann objects iterate in key order
            value = node[key]
            if key in transforms:
                walk(value, stack + [transforms[key]])
            cur = stack[-1]
            for t in reversed(stack[:-1]):
                cur = _mul(t, cur)
            if key == "instances":
                for name in value:
                    P, T, mat = objects[name]
                    for tri in T:
                        tris.append([_point(cur, P[i]) for i in tri])
                        mat_id.append(mat)
            elif key == "lights":
                for name in value:
                    kind, col, c0, ce = lights[name]
                    pos = [float(cur[0, 3]), float(cur[1, 3]), float(cur[2, 3])]
                    if kind == "spot":
                        d = _normalize(_vector(cur, np.array([0, 1, 0], F)))
                        recs.append(O.make_spot_light(col, pos, [float(v) for v in d], c0, ce, 1e-3))
                    else:
                        recs.append(O.make_point_light(col, pos, 1e-3))

    for key in sorted(data["world"]):
        if key in transforms:
            walk(data["world"][key], [transforms[key]])
    n = len(tris)
    xs, ys, zs = np.zeros((n, 4), F), np.zeros((n, 4), F), np.zeros((n, 4), F)
    for i, t in enumerate(tris):
        for v in range(3):
            xs[i, v], ys[i, v], zs[i, v] = t[v][0], t[v][1], t[v][2]
    out.update(xs=xs, ys=ys, zs=zs, mat_id=np.array(mat_id, np.uint32),
               bsdfs=np.array(bsdfs, np.uint8).reshape(-1, 32), lights=np.array(recs, np.uint8).reshape(-1, 32))
    return out
