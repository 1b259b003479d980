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


def _pint(name, v):
    return N('P', [('S', name), ('S', 'int'), ('S', 'Integer'), ('S', ''), ('I', int(v))])


def write(path, verts, polys, T=(0, 0, 0), R=(0, 0, 0), S=(1, 1, 1), unit=1.0, axes=None):
    """axes: None (no axis properties: the file counts as already in the importer's target system) or
    dict(up=(axis, sign), front=(axis, sign), coord=(axis, sign)) -> GlobalSettings UpAxis/.../CoordAxisSign."""
    idx = []
    for p in polys:
        idx += p[:-1] + [~p[-1]]
    gid, mid = 1001, 2002
    gprops = [N('P', [('S', 'UnitScaleFactor'), ('S', 'double'), ('S', 'Number'), ('S', ''), ('D', float(unit))])]
    if axes is not None:
        gprops = [_pint('UpAxis', axes['up'][0]), _pint('UpAxisSign', axes['up'][1]), _pint('FrontAxis', axes['front'][0]),
                  _pint('FrontAxisSign', axes['front'][1]), _pint('CoordAxis', axes['coord'][0]), _pint('CoordAxisSign', axes['coord'][1])] + gprops
    tops = [
        N('GlobalSettings', [], [N('Properties70', [], gprops)]),
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


def axis_matrix(axes):
    """Rows = the file's coord, front and up vectors: p' = A p carries the file's system onto (X coord, +Y front, +Z up),
    the importer's target FbxAxisSystem(eZAxis, eParityOdd, eLeftHanded) (core-mesh-parser.cpp:636-655)."""
    A = np.zeros((3, 3))
    for row, key in enumerate(('coord', 'front', 'up')):
        A[row, axes[key][0]] = 1.0 if axes[key][1] >= 0 else -1.0
    return A


def expected_triangles(verts, polys, T, R, S, unit, axes=None):
    rx, ry, rz = np.radians(R)
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    P = ((Rz @ Ry @ Rx) @ (verts * np.array(S)).T).T + np.array(T)
    P = P * unit
    mirrored = False
    if axes is not None:
        A = axis_matrix(axes)
        P = (A @ P.T).T
        mirrored = np.linalg.det(A) < 0          # a right-handed file: the map to the left-handed target is a reflection
    P = P.astype(np.float32)
    tris = []
    for p in polys:
        for k in range(1, len(p) - 1):
            b, c = (k + 1, k) if mirrored else (k, k + 1)   # ... which reverses the winding
            tris.append([P[p[0]], P[p[b]], P[p[c]]])
    return np.array(tris, np.float32)


FIXTURE = dict(T=(0.5, -0.25, 1.0), R=(20.0, -35.0, 50.0), S=(1.5, 1.0, 0.75), unit=2.0)
# three axis systems a file may declare (GlobalSettings), each on the same transformed sphere
AXIS_FIXTURES = {
    "blender_zup_rh": dict(up=(2, 1), front=(1, -1), coord=(0, 1)),    # Blender's exports: Z up, -Y front, X coord (right-handed)
    "maya_yup_rh": dict(up=(1, 1), front=(2, 1), coord=(0, 1)),        # FBX default / Maya: Y up, +Z front, X coord (right-handed)
    "target_zup_lh": dict(up=(2, 1), front=(1, 1), coord=(0, 1)),      # the importer's own target: nothing to convert
    "xup_lh": dict(up=(0, 1), front=(2, 1), coord=(1, 1)),             # X up, +Z front, +Y coord: det[c u f] = -1, left-handed, a pure rotation
}

if __name__ == "__main__":
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "tests" / "golden" / "fbx"
    out.mkdir(parents=True, exist_ok=True)
    v, p = uv_sphere()
    write(out / "uv_sphere_trs.fbx", v, p, **FIXTURE)
    v2, p2 = uv_sphere(12, 16, 0.8)
    write(out / "ball.fbx", v2, p2)
    for name, axes in AXIS_FIXTURES.items():
        write(out / f"uv_sphere_{name}.fbx", v, p, axes=axes, **FIXTURE)
    print("wrote", sorted(x.name for x in out.iterdir()))
