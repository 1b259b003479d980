"""PBRT-v4 subset front-end: host C++ loader vs the independent numpy restatement, and its rejections.
The fixture is this repo's own scene file with the parameters of the reference's scenes/cornell-box.pbrt."""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import pbrt_scene_ref  # noqa: E402

SCENE = GOLDEN / "pbrt" / "cornell_box.pbrt"


def test_loader_matches_restatement(pkg, O):
    got = pkg.host_scene.load_pbrt(SCENE)
    ref = pbrt_scene_ref.load(SCENE, O)
    assert got.tri_count == 36 and np.array_equal(got.mat_id, ref["mat_id"])
    for k in ("xs", "ys", "zs"):
        assert np.abs(getattr(got, k) - ref[k]).max() < 1e-6, k
    assert np.array_equal(got.bsdfs, ref["bsdfs"])              # Lambert R/pi as packed Oren-Nayar, roughness 0
    assert np.array_equal(got.area_tri, ref["area_tri"]) and np.array_equal(got.area_le, ref["area_le"])
    assert got.lights.shape[0] == 0 and got.inf_lights.shape[0] == 0
    assert (got.width, got.height, got.spp, got.max_depth) == (ref["width"], ref["height"], ref["spp"], ref["max_depth"])
    f = got.camera[:24].view(np.float32)
    assert np.array_equal(f[:3], ref["dir"]) and np.array_equal(f[3:6], ref["pos"])
    assert got.camera[36:44].view(np.float32).tolist() == [float(ref["focal"]), 36.0]


def test_emitter_faces_into_the_room(pkg):
    """One of the two emitter triangles of the scene faces down into the room, the other up into the ceiling gap
    (index order 0 1 2 / 0 2 3 on a strip-ordered quad): pbrt's own picture shows exactly that."""
    s = pkg.host_scene.load_pbrt(SCENE)
    n = []
    for t in s.area_tri:
        p = np.stack([s.xs[t, :3], s.ys[t, :3], s.zs[t, :3]], -1)     # [vertex, xyz]
        n.append(np.cross(p[1] - p[0], p[2] - p[0])[1])
    assert n[0] < 0 < n[1]


def test_if_the_reference_scene_is_here_it_loads_identically(pkg):
    ref_scene = Path("/root/reference/scenes/cornell-box.pbrt")
    if not ref_scene.exists():
        pytest.skip("reference assets are not on this machine")
    a, b = pkg.host_scene.load_pbrt(ref_scene), pkg.host_scene.load_pbrt(SCENE)
    for k in ("xs", "ys", "zs", "mat_id", "bsdfs", "area_tri", "area_le", "camera"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


@pytest.mark.parametrize("edit, needle", [
    (lambda t: t.replace('Camera "perspective"', 'Camera "orthographic"'), 'only "perspective"'),
    (lambda t: t.replace('[0.63 0.06 0.06] "string type" "diffuse"', '[0.63 0.06 0.06] "string type" "conductor"'), 'only "diffuse" materials'),
    (lambda t: t.replace('  NamedMaterial "matte_green"', '  NamedMaterial "matte_blue"'), "is not defined"),
    (lambda t: t.replace("WorldBegin", "WorldBegin\nLightSource \"goniometric\""), 'LightSource "goniometric" is not supported'),
    (lambda t: t.replace("WorldBegin", "WorldBegin\nLightSource \"infinite\" \"string filename\" \"sky.exr\""), "image maps are not supported"),
    (lambda t: t.replace("WorldBegin", "WorldBegin\nMakeNamedMedium \"fog\""), "'MakeNamedMedium' is not supported"),
    (lambda t: t.replace('NamedMaterial "matte_red"\n  Shape "trianglemesh"', 'NamedMaterial "matte_red"\n  Shape "sphere"'), 'only "trianglemesh"'),
    (lambda t: t.replace("  0 0 1\nCamera", "  0 1 0\nCamera"), "up = +z"),
    (lambda t: t.replace("      0 2 3\n", "      0 2 9\n", 1), "index out of range"),
    (lambda t: t.replace("AttributeEnd", "", 1) + "\nAttributeEnd\nAttributeEnd\n", "AttributeEnd without AttributeBegin"),
    (lambda t: t.replace('"rgb L" [20 20 20]', '"rgb L" [20 20]'), '"rgb L" with three values'),
])
def test_rejections(pkg, tmp_path, edit, needle):
    p = tmp_path / "s.pbrt"
    p.write_text(edit(SCENE.read_text()))
    with pytest.raises(ValueError) as e:
        pkg.host_scene.load_pbrt(p)
    assert needle in str(e.value), str(e.value)


def test_light_sources_map_to_the_reference_light_records(pkg, O, tmp_path):
    """LightSource point / spot / distant / infinite -> the reference's packed Light records (CC/private/light.cu:271-307),
    positions and directions transformed by the CTM and mirrored with the geometry."""
    text = SCENE.read_text().replace("WorldBegin", """WorldBegin
AttributeBegin
  Translate 0.25 1 0.5
  LightSource "point" "rgb I" [3 2 1] "point3 from" [0.1 0 0]
  LightSource "spot" "rgb I" [5 5 4] "point3 from" [0 0 0] "point3 to" [0 1 -1] "float coneangle" 40 "float conedeltaangle" 10 "float scale" 2
AttributeEnd
LightSource "distant" "rgb L" [1 1 2] "point3 from" [0 0 1] "point3 to" [1 0 0]
LightSource "infinite" "rgb L" [0.2 0.3 0.4]
""")
    p = tmp_path / "lights.pbrt"
    p.write_text(text)
    s = pkg.host_scene.load_pbrt(p)
    # the camera's right axis is +x here, so mirroring negates x
    want = [O.make_point_light([3, 2, 1], [-0.35, 1, 0.5], 1e-3),
            O.make_spot_light([10, 10, 8], [-0.25, 1, 0.5], [0, 2 ** -0.5, -2 ** -0.5], float(np.float32(np.cos(np.radians(30.0)))),
                              float(np.float32(np.cos(np.radians(40.0)))), 1e-3),
            O.make_directional_light([1, 1, 2], [-2 ** -0.5, 0, -2 ** -0.5], 0.0)]
    assert s.lights.shape[0] == 3
    for got, w in zip(s.lights, want):
        assert np.array_equal(got, w), (got, w)
    assert s.inf_lights.shape[0] == 1 and np.array_equal(s.inf_lights[0], O.make_env_light([0.2, 0.3, 0.4]))
    assert len(s.area_tri) == 2                       # the scene's own emitter is still there
