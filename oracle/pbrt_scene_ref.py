"""Test oracle for the PBRT-v4 subset front-end: an independent numpy restatement of the directives
host/dmt_pbrt_scene.cpp accepts, with pbrt-v4's semantics (CTM post-multiplication, Rotate by Rodrigues, trianglemesh
indices, diffuse area lights inheriting the current material, fov on the shorter axis), followed by the documented
mapping to the megakernel's arrays (mirror in the camera's right axis + rewind, Lambert as Oren-Nayar roughness 0).
TEST INFRASTRUCTURE ONLY.  The reference has no loader for this format: parity unpinned beyond this restatement and the
comparison with pbrt's own output image."""
import math
import re

import numpy as np


def _tokens(text):
    text = re.sub(r"#[^\n]*", "", text)
    return re.findall(r'"[^"]*"|\[|\]|[^\s\[\]"]+', text)


def _rot(deg, axis):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    t = math.radians(deg)
    s, c = math.sin(t), math.cos(t)
    x, y, z = a
    m = np.eye(4)
    m[:3, :3] = [[x * x + (1 - x * x) * c, x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                 [x * y * (1 - c) + z * s, y * y + (1 - y * y) * c, y * z * (1 - c) - x * s],
                 [x * z * (1 - c) - y * s, y * z * (1 - c) + x * s, z * z + (1 - z * z) * c]]
    return m


def load(path, O):
    tk = _tokens(open(path).read())
    i = 0

    def params():
        nonlocal i
        out = {}
        while i < len(tk) and tk[i].startswith('"'):
            typ, name = tk[i][1:-1].split()
            i += 1
            vals = []
            if tk[i] == "[":
                i += 1
                while tk[i] != "]":
                    vals.append(tk[i])
                    i += 1
                i += 1
            else:
                vals.append(tk[i])
                i += 1
            out[name] = (typ, [v[1:-1] if v.startswith('"') else float(v) for v in vals])
        return out

    ctm, stack = np.eye(4), []
    mat, emissive, L = -1, False, None
    names, mats, default = {}, [], None
    film, spp, maxdepth, fov = {}, 16, 5, 90.0
    eye = look = up = None
    tris = []
    while i < len(tk):
        d = tk[i]
        i += 1
        if d in ("Film", "Sampler", "Integrator", "ColorSpace", "PixelFilter", "Accelerator", "Camera"):
            i += 1
            p = params()
            if d == "Film":
                film = p
            elif d == "Sampler" and "pixelsamples" in p:
                spp = int(p["pixelsamples"][1][0])
            elif d == "Integrator" and "maxdepth" in p:
                maxdepth = int(p["maxdepth"][1][0])
            elif d == "Camera":
                fov = p.get("fov", ("float", [90.0]))[1][0]
                ctm = np.eye(4)
        elif d == "Option":
            params()
        elif d == "LookAt":
            v = [float(x) for x in tk[i:i + 9]]
            i += 9
            eye, look, up = np.array(v[0:3]), np.array(v[3:6]), np.array(v[6:9])
        elif d == "WorldBegin":
            ctm, mat, emissive, L = np.eye(4), -1, False, None
        elif d == "AttributeBegin":
            stack.append((ctm.copy(), mat, emissive, L))
        elif d == "AttributeEnd":
            ctm, mat, emissive, L = stack.pop()
        elif d in ("Translate", "Scale"):
            v = [float(x) for x in tk[i:i + 3]]
            i += 3
            m = np.eye(4)
            if d == "Translate":
                m[:3, 3] = v
            else:
                m[0, 0], m[1, 1], m[2, 2] = v
            ctm = ctm @ m
        elif d == "Rotate":
            v = [float(x) for x in tk[i:i + 4]]
            i += 4
            ctm = ctm @ _rot(v[0], v[1:])
        elif d == "MakeNamedMaterial":
            name = tk[i][1:-1]
            i += 1
            p = params()
            names[name] = len(mats)
            mats.append(p.get("reflectance", ("rgb", [0.5, 0.5, 0.5]))[1])
        elif d == "NamedMaterial":
            mat = names[tk[i][1:-1]]
            i += 1
        elif d == "AreaLightSource":
            i += 1
            p = params()
            emissive, L = True, [v * p.get("scale", ("float", [1.0]))[1][0] for v in p["L"][1]]
        elif d == "Shape":
            i += 1
            p = params()
            P = np.array(p["P"][1], np.float64).reshape(-1, 3)
            idx = [int(v) for v in p["indices"][1]] if "indices" in p else [0, 1, 2]
            if mat < 0 and default is None:
                default = len(mats)
                mats.append([0.5, 0.5, 0.5])
            W = (ctm[:3, :3] @ P.T).T + ctm[:3, 3]
            for k in range(0, len(idx), 3):
                tris.append((W[idx[k]], W[idx[k + 1]], W[idx[k + 2]], default if mat < 0 else mat, emissive, L))
        else:
            raise ValueError(d)
    dirv = look - eye
    right = np.cross(dirv, up)
    right /= np.linalg.norm(right)

    def mirror(p):
        p = p.astype(np.float32)                        # the loader mirrors the float32 world-space vertex
        dist = np.float32(np.dot((p - eye.astype(np.float32)), right.astype(np.float32)))
        return p - np.float32(2) * dist * right.astype(np.float32)

    n = len(tris)
    xs, ys, zs = (np.zeros((n, 4), np.float32) for _ in range(3))
    mat_id, area_tri, area_le = np.zeros(n, np.uint32), [], []
    for t, (a, b, c, m, e, Lr) in enumerate(tris):
        for v, p in enumerate((mirror(a), mirror(c), mirror(b))):   # rewound
            xs[t, v], ys[t, v], zs[t, v] = p
        mat_id[t] = m
        if e:
            area_tri.append(t)
            area_le.append(Lr)
    bsdfs = np.array([O.make_oren_nayar(r, 0.0) for r in mats], np.uint8).reshape(-1, 32)
    return dict(xs=xs, ys=ys, zs=zs, mat_id=mat_id, bsdfs=bsdfs, area_tri=np.array(area_tri, np.uint32),
                area_le=np.array(area_le, np.float32).reshape(-1, 3), width=int(film["xresolution"][1][0]),
                height=int(film["yresolution"][1][0]), spp=spp, max_depth=maxdepth,
                focal=np.float32(18.0 / math.tan(math.radians(fov) / 2)), dir=dirv.astype(np.float32), pos=eye.astype(np.float32))
