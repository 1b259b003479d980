#!/usr/bin/env python3
"""Writes the binary-FBX fixtures of tests/golden/fbx/ (this repo's own files, Kaydara binary 7400 layout):
a UV sphere with quads + triangle fans at the poles, deflated arrays, a Model with Lcl TRS and a unit scale.
Also returns the triangles a correct reader must produce (numpy float64 -> float32)."""
import struct
import sys
import zlib
from pathlib import Path

import numpy as np


def _prop(t, v):
    if t == 'S':
        b = v if isinstance(v, bytes) else v.encode()
        return b'S' + struct.pack('<I', len(b)) + b
    if t in 'IDLFYC':
        fmt = {'I': '<i', 'D': '<d', 'L': '<q', 'F': '<f', 'Y': '<h', 'C': '<B'}[t]
        return t.encode() + struct.pack(fmt, v)
    if t in 'di':
        a = np.asarray(v, {'d': '<f8', 'i': '<i4'}[t]).tobytes()
        z = zlib.compress(a)
        return t.encode() + struct.pack('<III', len(v), 1, len(z)) + z
    raise ValueError(t)


def _node(name, props=(), children=(), base=0):
    """Record with absolute end offset; `base` = file offset where this record starts."""
    pbytes = b''.join(_prop(t, v) for t, v in props)
    head_len = 13 + len(name)
    body = b''
    off = base + head_len + len(pbytes)
    for c in children:
        rec = c(off)
        body += rec
        off += len(rec)
    if children:
        body += b'\x00' * 13
        off += 13
    return struct.pack('<IIIB', off, len(props), len(pbytes), len(name)) + name.encode() + pbytes + body


def N(name, props=(), children=()):
    return lambda base: _node(name, props, children, base)


def P70(name, kind, *vals):
    return N('P', [('S', name), ('S', kind), ('S', ''), ('S', 'A')] + [('D', float(v)) for v in vals])


def uv_sphere(lat=6, lon=8, radius=1.0):
    verts = [(0.0, 0.0, radius)]
    for i in range(1, lat):
        th = np.pi * i / lat
        for j in range(lon):
            ph = 2 * np.pi * j / lon
            verts.append((radius * np.sin(th) * np.cos(ph), radius * np.sin(th) * np.sin(ph), radius * np.cos(th)))
    verts.append((0.0, 0.0, -radius))
    polys = []
    ring = lambda i, j: 1 + (i - 1) * lon + (j % lon)
    for j in range(lon):
        polys.append([0, ring(1, j), ring(1, j + 1)])
    for i in range(1, lat - 1):
        for j in range(lon):
            polys.append([ring(i, j), ring(i + 1, j), ring(i + 1, j + 1), ring(i, j + 1)])   # quads
    south = len(verts) - 1
    for j in range(lon):
        polys.append([south, ring(lat - 1, j + 1), ring(lat - 1, j)])
    return np.array(verts, np.float64), polys


def write(path, verts, polys, T=(0, 0, 0), R=(0, 0, 0), S=(1, 1, 1), unit=1.0):
    idx = []
    for p in polys:
        idx += p[:-1] + [~p[-1]]
    gid, mid = 1001, 2002
    tops = [
        N('GlobalSettings', [], [N('Properties70', [], [N('P', [('S', 'UnitScaleFactor'), ('S', 'double'), ('S', 'Number'), ('S', ''), ('D', float(unit))])])]),
        N('Objects', [], [
            N('Geometry', [('L', gid), ('S', b'MESH_Fixture\x00\x01Geometry'), ('S', 'Mesh')],
              [N('Vertices', [('d', verts.reshape(-1))]), N('PolygonVertexIndex', [('i', idx)])]),
            N('Model', [('L', mid), ('S', b'Fixture\x00\x01Model'), ('S', 'Mesh')],
              [N('Properties70', [], [P70('Lcl Translation', 'Lcl Translation', *T), P70('Lcl Rotation', 'Lcl Rotation', *R),
                                      P70('Lcl Scaling', 'Lcl Scaling', *S)])]),
        ]),
        N('Connections', [], [N('C', [('S', 'OO'), ('L', mid), ('L', 0)]), N('C', [('S', 'OO'), ('L', gid), ('L', mid)])]),
    ]
    out = b'Kaydara FBX Binary  \x00\x1a\x00' + struct.pack('<I', 7400)
    for t in tops:
        out += t(len(out))
    out += b'\x00' * 13
    Path(path).write_bytes(out)


def expected_triangles(verts, polys, T, R, S, unit):
    rx, ry, rz = np.radians(R)
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    P = ((Rz @ Ry @ Rx) @ (verts * np.array(S)).T).T + np.array(T)
    P = (P * unit).astype(np.float32)
    tris = []
    for p in polys:
        for k in range(1, len(p) - 1):
            tris.append([P[p[0]], P[p[k]], P[p[k + 1]]])
    return np.array(tris, np.float32)


FIXTURE = dict(T=(0.5, -0.25, 1.0), R=(20.0, -35.0, 50.0), S=(1.5, 1.0, 0.75), unit=2.0)

if __name__ == "__main__":
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "tests" / "golden" / "fbx"
    out.mkdir(parents=True, exist_ok=True)
    v, p = uv_sphere()
    write(out / "uv_sphere_trs.fbx", v, p, **FIXTURE)
    v2, p2 = uv_sphere(12, 16, 0.8)
    write(out / "ball.fbx", v2, p2)
    print("wrote", sorted(x.name for x in out.iterdir()))
