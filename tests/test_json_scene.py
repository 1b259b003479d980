"""JSON scene front-end: host C++ loader (host/dmt_json_scene.cpp) vs the independent numpy restatement
of the reference's parser semantics (oracle/json_scene_ref.py), plus the rejection rules of core-parser.cpp.
The fixture scene is this repo's own file in the reference's schema (tests/golden/json_scene/).  No GPU needed."""
import json
import shutil
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import json_scene_ref  # noqa: E402

SCENE = GOLDEN / "json_scene" / "three_boxes.json"


def test_loader_matches_restatement(pkg, O):
    got = pkg.host_scene.load_json(SCENE)
    ref = json_scene_ref.load(SCENE, O)
    assert got.tri_count == 26 and np.array_equal(got.mat_id, ref["mat_id"])
    # world members are visited in key order: ground (plane, chalk), left (cube, glass), right (cube, gold)
    assert got.mat_id.tolist() == [2] * 2 + [0] * 12 + [1] * 12
    for k in ("xs", "ys", "zs"):
        # transforms are float32 products in glm's order on both sides; cosf/sinf may differ in the last ulp
        assert np.abs(getattr(got, k) - ref[k]).max() < 2e-6, k
    assert np.array_equal(got.bsdfs, ref["bsdfs"])            # packed records: byte-identical
    assert got.lights.shape == (2, 32) and got.inf_lights.shape[0] == 0
    assert np.array_equal(got.lights, ref["lights"])
    cam = got.camera
    f = cam[:24].view(np.float32)
    assert np.array_equal(f[:3], ref["camera"]["dir"]) and np.array_equal(f[3:6], ref["camera"]["pos"])
    assert (got.width, got.height) == (96, 64) and cam[32:36].view(np.int32)[0] == 8
    assert cam[36:44].view(np.float32).tolist() == [24.0, 36.0]
    assert (got.max_depth, got.spp) == (6, 8)


def test_fractional_metallic_becomes_a_record_pair(pkg, O, tmp_path):
    """core-material.cpp:272-286: metallic <= 0 -> dielectric, >= 1 -> conductor, in between BOTH lobes (blended per hit):
    the loader emits the dielectric record tagged BS_GGX_BLEND with the fraction in its weight field, followed by the
    conductor record; later materials' indices shift; packed bytes equal the numpy restatement's."""
    def edit(d):
        d["materials"][1]["metallic"] = 0.3          # 'gold' of the fixture: was 1.0
    p = _variant(tmp_path, edit)
    got = pkg.host_scene.load_json(p)
    ref = json_scene_ref.load(p, O)
    assert got.bsdfs.shape[0] == 4 and np.array_equal(got.bsdfs, ref["bsdfs"])
    types = [int(r[6:8].view(np.uint16)[0]) for r in got.bsdfs]
    assert types == [1, 4, 2, 0]                     # glass, gold (blend dielectric + conductor), chalk
    assert abs(float(got.bsdfs[1, 0:2].view(np.float16)[0]) - 0.3) < 2e-4
    assert np.array_equal(got.mat_id, ref["mat_id"]) and got.mat_id.tolist() == [3] * 2 + [0] * 12 + [1] * 12
    # the two GGX records of the pair share the material's alphas
    assert np.array_equal(got.bsdfs[1, 12:18], got.bsdfs[2, 12:18])


def test_envlight_png_is_loaded_as_bytes_over_255(pkg):
    got = pkg.host_scene.load_json(SCENE)
    assert got.env_rgb.shape == (16, 32, 3)
    # the fixture PNG was written with rows 40 + 8 y / 60 + 4 x / 120 and three white texels
    y, x = 5, 9
    assert got.env_rgb[y, x].tolist() == [np.float32((40 + 8 * y) / 255.0), np.float32((60 + 4 * x) / 255.0),
                                          np.float32(120 / 255.0)]
    assert (got.env_rgb == 1.0).all(axis=2).sum() == 3


def _variant(tmp_path, edit):
    d = json.loads(SCENE.read_text())
    edit(d)
    shutil.copy(SCENE.parent / "sky_32x16.png", tmp_path / "sky_32x16.png")
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(d))
    return p


@pytest.mark.parametrize("edit, needle", [
    (lambda d: d.pop("lights"), "lacking some keys"),                                        # core-parser.cpp:1371
    (lambda d: d.update(extra=1), "extraneous key"),
    (lambda d: d["camera"].update(position=[0, 0, 0]), "at most 4 members"),                 # :741 (5 members)
    (lambda d: d["film"].update(samples=0), "positive integer"),                             # :290-294
    (lambda d: d["film"].update(samples=2.5), "positive integer"),
    (lambda d: d["materials"][0].pop("roughness"), "should specify a 'roughness'"),          # :480-486
    (lambda d: d["materials"][0].update(shiny=True), "extraneous key 'shiny'"),              # :446-459
    (lambda d: d["materials"][1].update(eta=[1, 1, 1]), "both 'eta' and 'etak'"),            # :600
    (lambda d: d["materials"][0].update({"oren-nayar-dielectric": None}), "only one of"),    # :665
    (lambda d: d["materials"].append(dict(d["materials"][0])), "Duplicate material name"),   # :421
    (lambda d: d["materials"][0].update(diffuse="paint"), "existing named texture"),         # :497
    (lambda d: d["objects"][0].update(material="nope"), "unknown material"),
    (lambda d: d["objects"][0].update(shape="torus"), "unrecognized shape"),
    (lambda d: d["objects"][0].update(type="FBX", path="x.fbx") or d["objects"][0].pop("shape"), "cannot open"),
    (lambda d: d["lights"][0].update(radius=1), "extraneous key 'radius'"),                  # :1001-1012
    (lambda d: d["lights"][1].update(type="area"), "unrecognized type"),
    (lambda d: d["transforms"][0].update(name=""), "non-empty 'name'"),                      # :836
    (lambda d: d["world"]["room"]["left"].update(instances=["ghost"]), "unknown object"),
    (lambda d: d.update(envlight="missing.png"), "envlight"),
])
def test_rejections(pkg, tmp_path, edit, needle):
    with pytest.raises(ValueError) as e:
        pkg.host_scene.load_json(_variant(tmp_path, edit))
    assert needle in str(e.value), str(e.value)


def test_defaults_and_clamps(pkg, O, tmp_path):
    """cone-angle clamps to [10, 120], falloff to [1, 80] (core-parser.cpp:1040-1052); lights default to intensity 1;
    a transform without srt is the identity; max-depth defaults to 5, samples to 1 (core-types.h:28-30)."""
    def edit(d):
        d["camera"].pop("max-depth")
        d["film"].pop("samples")
        d["lights"][0].update({"cone-angle": 500, "falloff-percentage": 0.01})
        d["lights"][0].pop("radiant-intensity")
        d["transforms"].append({"name": "id"})
        d["world"]["id"] = {"lights": ["bulb"]}
    p = _variant(tmp_path, edit)
    got, ref = pkg.host_scene.load_json(p), json_scene_ref.load(p, O)
    assert (got.max_depth, got.spp) == (5, 1)
    assert got.lights.shape[0] == 3 and np.array_equal(got.lights, ref["lights"])


def test_json_reader_details(pkg, tmp_path):
    """escapes, exponents, nested empties; a repeated key keeps the last value (nlohmann's behaviour)."""
    text = SCENE.read_text().replace('"samples": 8', '"samples": 8, "samples": 3').replace('"resolutionX": 96', '"resolutionX": 9.6e1')
    text = text.replace('"name": "bulb"', '"name": "b\\u0075lb"')
    shutil.copy(SCENE.parent / "sky_32x16.png", tmp_path / "sky_32x16.png")
    p = tmp_path / "s.json"
    p.write_text(text)
    got = pkg.host_scene.load_json(p)
    assert got.spp == 3 and got.width == 96 and got.lights.shape[0] == 2
    (tmp_path / "bad.json").write_text(text[:-3])
    with pytest.raises(ValueError):
        pkg.host_scene.load_json(tmp_path / "bad.json")


# ---------------------------------------------------------------------------------------------
# binary FBX reader (replaces the FBX SDK call of MeshFbxParser::ImportFBX).  Parity unpinned: no SDK
# here; checked against files written by tools/make_fbx_fixture.py and the structure of the reference's asset.
# ---------------------------------------------------------------------------------------------
sys.path.insert(0, str(ROOT / "tools"))
import make_fbx_fixture as fbxfix  # noqa: E402


def test_fbx_reader_fixture(pkg, tmp_path):
    v, polys = fbxfix.uv_sphere()
    got = pkg.host_scene.read_fbx(GOLDEN / "fbx" / "uv_sphere_trs.fbx")
    want = fbxfix.expected_triangles(v, polys, **fbxfix.FIXTURE)
    assert got.shape == want.shape == (80, 3, 3)         # 8 + 8 pole triangles + 32 quads fan-triangulated
    assert np.abs(got - want).max() < 1e-6
    # the committed fixture is what the writer produces (regenerable)
    fbxfix.write(tmp_path / "again.fbx", v, polys, **fbxfix.FIXTURE)
    assert (tmp_path / "again.fbx").read_bytes() == (GOLDEN / "fbx" / "uv_sphere_trs.fbx").read_bytes()


@pytest.mark.parametrize("name", sorted(fbxfix.AXIS_FIXTURES))
def test_fbx_axis_system_conversion(pkg, tmp_path, name):
    """core-mesh-parser.cpp:630-655: the importer converts every file to (Z up, parity-odd = +Y front, left-handed) before
    baking the node's transform.  Files declaring four different systems: the reader's triangles equal the numpy
    restatement's (change of basis on the global transform, winding reversed when the map is a reflection), the committed
    fixtures are reproducible, and an outward-wound sphere stays outward-wound in every system.  Unpinned (no SDK)."""
    axes = fbxfix.AXIS_FIXTURES[name]
    v, polys = fbxfix.uv_sphere()
    got = pkg.host_scene.read_fbx(GOLDEN / "fbx" / f"uv_sphere_{name}.fbx")
    want = fbxfix.expected_triangles(v, polys, axes=axes, **fbxfix.FIXTURE)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-5
    fbxfix.write(tmp_path / "again.fbx", v, polys, axes=axes, **fbxfix.FIXTURE)
    assert (tmp_path / "again.fbx").read_bytes() == (GOLDEN / "fbx" / f"uv_sphere_{name}.fbx").read_bytes()
    # same sphere without a node transform: geometric normals point away from the centre in the converted mesh
    fbxfix.write(tmp_path / "plain.fbx", v, polys, axes=axes)
    t = pkg.host_scene.read_fbx(tmp_path / "plain.fbx")
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    assert (np.einsum("ij,ij->i", n, t.mean(axis=1)) > 0).all()
    A = fbxfix.axis_matrix(axes)
    mirrored = np.linalg.det(A) < 0
    assert mirrored == (name.endswith("_rh"))             # right-handed files are reflected into the left-handed target
    # the semantic directions land where the target puts them: file "up" -> +Z, "front" -> +Y, "coord" -> +X
    e = np.eye(3)
    assert np.allclose(A @ (e[axes["up"][0]] * axes["up"][1]), [0, 0, 1]) and np.allclose(A @ (e[axes["front"][0]] * axes["front"][1]), [0, 1, 0])
    assert np.allclose(A @ (e[axes["coord"][0]] * axes["coord"][1]), [1, 0, 0])


def test_fbx_axis_system_must_be_a_permutation(pkg, tmp_path):
    v, polys = fbxfix.uv_sphere()
    fbxfix.write(tmp_path / "bad.fbx", v, polys, axes=dict(up=(2, 1), front=(2, -1), coord=(0, 1)))
    with pytest.raises(ValueError, match="not a permutation"):
        pkg.host_scene.read_fbx(tmp_path / "bad.fbx")


def test_reference_fbx_assets_axis_conversion(pkg, monkeypatch):
    """The reference's own meshes (data fixtures).  scenes/sphere.fbx declares Blender's system (Z up, -Y front, X coord,
    right-handed): (x, y, z) -> (x, -y, z).  scenes/teapot.fbx declares Y up, +Z front, X coord (right-handed): its height
    runs along +Y in the file and along +Z -- the renderer's up -- after the conversion, (x, y, z) -> (x, z, y); read raw
    (DMT_FBX_AXIS=off) it lies on its side.  Both maps are reflections, so v1 <-> v2."""
    cases = ((GOLDEN / "c3" / "sphere.fbx", lambda r: r * np.array([1, -1, 1], np.float32)),
             (GOLDEN / "scene_test" / "res" / "fbx" / "teapot.fbx", lambda r: r[..., [0, 2, 1]]))
    for f, conv_of in cases:
        conv = pkg.host_scene.read_fbx(f)
        monkeypatch.setenv("DMT_FBX_AXIS", "off")
        raw = pkg.host_scene.read_fbx(f)
        monkeypatch.delenv("DMT_FBX_AXIS")
        assert conv.shape == raw.shape
        assert np.allclose(conv[:, [0, 2, 1]], conv_of(raw), atol=1e-6)
    lo, hi = conv.reshape(-1, 3).min(0), conv.reshape(-1, 3).max(0)       # the teapot: stands on z = 0, 1.55 tall, spout along +x
    assert abs(lo[2]) < 1e-6 and abs(hi[2] - 1.5497) < 1e-3 and hi[0] > 1.6 and abs(hi[1] + lo[1]) < 1e-5


def test_fbx_reader_rejects_garbage(pkg, tmp_path):
    (tmp_path / "a.fbx").write_bytes(b"; FBX 7.4.0 project file\n")
    with pytest.raises(ValueError, match="not a binary FBX"):
        pkg.host_scene.read_fbx(tmp_path / "a.fbx")
    good = (GOLDEN / "fbx" / "ball.fbx").read_bytes()
    (tmp_path / "b.fbx").write_bytes(good[:200])
    with pytest.raises(ValueError):
        pkg.host_scene.read_fbx(tmp_path / "b.fbx")
    with pytest.raises(ValueError, match="cannot open"):
        pkg.host_scene.read_fbx(tmp_path / "missing.fbx")
    # forged sizes: an array record that claims 2^32 - 1 elements in a handful of stored bytes must be rejected
    # BEFORE anything is allocated, and a child record may not end beyond its parent
    import struct
    hdr = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", 7400)
    def rec(name, props, children=b"", end_delta=0):
        body = bytes([len(name)]) + name + props + children
        # end offset is absolute: patched by the caller through `at`
        return body, end_delta
    def node(at, name, nprops, props, children=b"", end_delta=0):
        n = 12 + 1 + len(name) + len(props) + len(children)
        return struct.pack("<III", at + n + end_delta, nprops, len(props)) + bytes([len(name)]) + name + props + children
    arr = b"d" + struct.pack("<III", 0xFFFFFFFF, 0, 16) + b"\0" * 16              # 4 G doubles "stored" in 16 raw bytes
    (tmp_path / "c.fbx").write_bytes(hdr + node(27, b"Objects", 1, arr) + b"\0" * 13)
    with pytest.raises(ValueError, match="array size inconsistent"):
        pkg.host_scene.read_fbx(tmp_path / "c.fbx")
    arr = b"d" + struct.pack("<III", 0x0FFFFFFF, 1, 16) + b"\0" * 16              # 2 GiB claimed from a 16-byte deflate stream
    (tmp_path / "d.fbx").write_bytes(hdr + node(27, b"Objects", 1, arr) + b"\0" * 13)
    with pytest.raises(ValueError, match="array size inconsistent"):
        pkg.host_scene.read_fbx(tmp_path / "d.fbx")
    child = node(27 + 12 + 1 + 7, b"Geometry", 0, b"", end_delta=40)            # child ends 40 bytes past its parent
    (tmp_path / "e.fbx").write_bytes(hdr + node(27, b"Objects", 0, b"", child) + b"\0" * 64)
    with pytest.raises(ValueError, match="bad record end offset"):
        pkg.host_scene.read_fbx(tmp_path / "e.fbx")


def test_fbx_reader_on_the_reference_asset(pkg):
    """scenes/sphere.fbx of the reference (BASELINE config 3): present in the build container only."""
    asset = Path("/root/reference/scenes/sphere.fbx")
    if not asset.exists():
        pytest.skip("reference assets are not on this machine")
    t = pkg.host_scene.read_fbx(asset)
    assert t.shape == (480, 3, 3)                         # 1440 polygon-vertex indices, all triangles
    r = np.linalg.norm(t.reshape(-1, 3), axis=1)
    assert abs(r.max() - 100.0) < 1e-3                    # unit mesh x Lcl Scaling 100 (Blender's centimetres)
    # closed surface: every undirected edge is shared by exactly two triangles
    q = np.round(t.reshape(-1, 3) * 1e3).astype(np.int64)
    ids = {tuple(p): i for i, p in enumerate(map(tuple, np.unique(q, axis=0)))}
    tri = np.array([ids[tuple(p)] for p in q]).reshape(-1, 3)
    edges = {}
    for a, b, c in tri:
        for e in ((a, b), (b, c), (c, a)):
            k = (min(e), max(e))
            edges[k] = edges.get(k, 0) + 1
    assert set(edges.values()) == {2}


def test_json_scene_with_fbx_object(pkg):
    """BASELINE config 3 in miniature: an FBX mesh under an env map (scenes/fbx_example.json's shape)."""
    s = pkg.host_scene.load_json(GOLDEN / "json_scene" / "ball_envmap.json")
    ball = pkg.host_scene.read_fbx(GOLDEN / "fbx" / "ball.fbx")
    assert s.tri_count == ball.shape[0] + 2
    assert s.mat_id.tolist() == [1] * 2 + [0] * ball.shape[0]   # key order: position-floor < position-light < position-obj
    moved = ball + np.array([0, 3, 0], np.float32)
    got = np.stack([s.xs[2:, :3], s.ys[2:, :3], s.zs[2:, :3]], -1)   # [tri, vertex, xyz]
    assert np.abs(got - moved).max() < 1e-6
    assert s.env_rgb is not None and s.lights.shape[0] == 1 and s.max_depth == 8


def test_c3_fbx_reader_on_the_reference_asset(pkg):
    """sphere.fbx through the binary reader: 480 triangles, closed, radius 100 file units (Lcl Scaling 100 x unit mesh)."""
    T = pkg.host_scene.read_fbx(GOLDEN / "c3" / "sphere.fbx")
    assert T.shape == (480, 3, 3)
    r = np.linalg.norm(T.reshape(-1, 3), axis=1)
    assert abs(r.max() - 100.0) < 1e-3 and r.min() > 55.0
    # closed: every undirected edge is shared by exactly two triangles
    V, inv = np.unique(np.round(T.reshape(-1, 3), 3), axis=0, return_inverse=True)
    assert len(V) == 242
    F = inv.reshape(-1, 3)
    e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert np.all(counts == 2)


def test_c3_scene_files_load(pkg):
    """BASELINE configs[2] fixtures: the reference's fbx_example.json (literal, env-map name fixed) and its metres
    reading both load; same mesh, 100x apart, same veranda map (see tests/test_configs_gpu.py for the GPU renders)."""
    lit = pkg.host_scene.load_json(GOLDEN / "c3" / "fbx_example_literal.json")
    met = pkg.host_scene.load_json(GOLDEN / "c3" / "c3_sphere_veranda.json")
    assert lit.tri_count == met.tri_count == 480 and lit.spp == 32 and met.spp == 2048 and lit.max_depth == met.max_depth == 12
    assert lit.env_rgb.shape == (512, 1024, 3) and np.array_equal(lit.env_rgb, met.env_rgb)
    c = np.array([0, 2, -1], np.float32)
    P = lambda s: np.stack([s.xs[:, :3], s.ys[:, :3], s.zs[:, :3]], -1).reshape(-1, 3) - c
    assert np.allclose(P(lit), 100 * P(met), rtol=1e-5, atol=1e-4)
    assert abs(np.linalg.norm(P(met), axis=1).max() - 1.0) < 1e-5


def test_reference_scene_example_loads_unmodified(pkg):
    """The reference's scenes/scene_example.json, byte for byte (tests/golden/c3/ holds it next to the veranda map it
    names): a unit cube (12 triangles) with a GGX dielectric, the spot light placed by the world, the env map."""
    s = pkg.host_scene.load_json(GOLDEN / "c3" / "scene_example.json")
    assert s.tri_count == 12 and (s.width, s.height, s.spp, s.max_depth) == (256, 256, 32, 12)
    assert s.lights.shape == (1, 32) and s.env_rgb.shape == (512, 1024, 3) and s.bsdfs.shape == (1, 32)
    P = np.stack([s.xs[:, :3], s.ys[:, :3], s.zs[:, :3]], -1).reshape(-1, 3)
    assert np.allclose(P.min(0), [-0.5, 1.5, -1.5]) and np.allclose(P.max(0), [0.5, 2.5, -0.5])


def test_reference_scene_test_json_loads_unmodified(pkg):
    """The reference's scenes/scene_test.json, byte for byte, with its teapot.fbx and chipped-paint textures (data fixtures
    under tests/golden/scene_test/).  The env map it names (res/envLight/autumn_field_puresky_4k.png) does not exist in
    the reference tree; a small synthetic stand-in sits at that path.  Image textures + normal map: SURVEY 8f-1."""
    d = GOLDEN / "scene_test"
    s = pkg.host_scene.load_json(d / "scene_test.json")
    assert s.tri_count == 9216 and (s.width, s.height, s.spp, s.max_depth) == (256, 256, 32, 12)
    assert s.tex_desc.tolist() == [[0, 1024, 1024], [1 << 20, 1024, 1024], [2 << 20, 1024, 1024]]   # albedo, normal, roughness
    assert s.tex_rgba.shape == (3 << 20, 4) and (s.tex_rgba[:, 3] == 255).all()
    rough = s.tex_rgba[2 << 20:]
    assert (rough[:, 0] == rough[:, 1]).all() and (rough[:, 0] == rough[:, 2]).all()               # grey PNG replicated
    assert s.mat_tex.shape == (1, 4) and s.mat_tex[0, :3].tolist() == [0, 2, 1]                       # diffuse, roughness, normal
    assert s.mat_tex[0, 3:].view(np.float32)[0] == 1.0                                               # "ggx-anisotropy": 0 -> 1
    assert s.tri_uv.shape == (9216, 6) and 0.0 <= s.tri_uv.min() and s.tri_uv.max() <= 2.0 and s.tri_uv.std() > 0.1
    # an untextured scene uploads no texture tables at all
    assert pkg.host_scene.load_json(GOLDEN / "c3" / "scene_example.json").tex_desc is None


def test_texture_rules_of_the_parser(pkg, tmp_path):
    """core-parser.cpp:306-560: texture / material cross checks."""
    import shutil
    src = GOLDEN / "scene_test"
    shutil.copytree(src, tmp_path / "s")
    base = json.loads((src / "scene_test.json").read_text())
    def load(mut):
        j = json.loads(json.dumps(base))
        mut(j)
        (tmp_path / "s" / "m.json").write_text(json.dumps(j))
        return pkg.host_scene.load_json(tmp_path / "s" / "m.json")
    with pytest.raises(ValueError, match="should point to a 'diffuse' texture"):
        load(lambda j: j["materials"][0].__setitem__("diffuse", "chippedPaintNormal"))
    with pytest.raises(ValueError, match="existing named texture"):
        load(lambda j: j["materials"][0].__setitem__("roughness", "nope"))
    with pytest.raises(ValueError, match="expect 1 channel"):
        load(lambda j: j["textures"][2].__setitem__("path", "./res/textures/chippedPaint/Paint_Chipped_1K_albedo.png"))
    # a 'metallic' MAP (1-channel): the material becomes a record pair -- dielectric tagged BS_GGX_BLEND (4), then its
    # conductor -- with one texture row per record; the metallic map sits in the first slot of the SECOND row
    def metallic_tex(j):
        j["textures"].append({"name": "m", "type": "metallic", "path": "./res/textures/chippedPaint/Paint_Chipped_1K_roughness.png"})
        j["materials"][0]["metallic"] = "m"
    sm = load(metallic_tex)
    assert sm.bsdfs.shape[0] == 2 and sm.bsdfs[0, 6:8].view(np.uint16)[0] == 4 and sm.bsdfs[1, 6:8].view(np.uint16)[0] == 2
    assert sm.mat_tex.shape == (2, 4)
    assert sm.mat_tex[0, :3].tolist() == [0, 2, 1] and sm.mat_tex[1, :3].tolist() == [3, 2, 1]
    assert set(sm.mat_id.tolist()) == {0}
    s = load(lambda j: j["materials"][0].pop("normal"))
    assert s.mat_tex[0, :3].tolist() == [0, 2, 0xFFFFFFFF]

