"""CPU tests: the oracle against the reference's own known-answer test, the survey anchors and
the committed golden vectors.  No GPU."""
import json

import numpy as np
import pytest

from conftest import GOLDEN, film_rmse, golden

PINS = json.loads((GOLDEN / "reference_pins.json").read_text())


# ---- the reference's own test: T/tests/triangle_intersect.cu:12-68,164-186 --------------------
def reference_triangle_soup(count):
    """generateTriangleSoup: 4 shapes cycling (XY ccw, XY cw, slanted, degenerate), base x = i."""
    xs = np.zeros((count, 4), np.float32)
    ys = np.zeros((count, 4), np.float32)
    zs = np.zeros((count, 4), np.float32)
    i = np.arange(count)
    base = i.astype(np.float32)
    t = i % 4
    v0 = np.stack([base, np.zeros(count), np.zeros(count)], 1)
    v1 = np.where((t == 0)[:, None], np.stack([base + 1, 0 * base, 0 * base], 1),
         np.where((t == 1)[:, None], np.stack([base, 0 * base + 1, 0 * base], 1),
         np.where((t == 2)[:, None], np.stack([base + 1, 0 * base, 0 * base + 0.5], 1), v0)))
    v2 = np.where((t == 0)[:, None], np.stack([base, 0 * base + 1, 0 * base], 1),
         np.where((t == 1)[:, None], np.stack([base + 1, 0 * base, 0 * base], 1),
         np.where((t == 2)[:, None], np.stack([base, 0 * base + 1, 0 * base + 0.25], 1), v0)))
    for arr, c in ((xs, 0), (ys, 1), (zs, 2)):
        arr[:, 0], arr[:, 1], arr[:, 2] = v0[:, c], v1[:, c], v2[:, c]
    return xs, ys, zs


RAY_A = (np.array([0.25, 0.25, 5.0], np.float32), np.array([0.0, 0.0, -1.0], np.float32))
RAY_B = (np.array([-5.0, 5.0, 5.0], np.float32), np.array([0.0, 0.0, -1.0], np.float32))


@pytest.mark.parametrize("ray", [RAY_A, RAY_B], ids=["rayA", "rayB"])
def test_reference_triangle_kat(O, ray):
    """The reference asserts triangleIntersect (device routine) == hostIntersectMT on 65,536
    generated triangles; both restatements must agree the same way."""
    xs, ys, zs = reference_triangle_soup(1 << 16)
    hit, t, pos, nrm, err = O.triangle_intersect(xs, ys, zs, *ray)
    expected = O.host_intersect_mt(xs, ys, zs, *ray)
    assert np.array_equal(hit, expected)
    if ray is RAY_A:
        assert hit[0] == 1 and hit.sum() == 1          # only triangle 0 covers (0.25, 0.25)
        assert t[0] == np.float32(5.0)
        assert np.allclose(pos[0], [0.25, 0.25, 0.0])
        assert np.allclose(np.abs(nrm[0]), [0, 0, 1])
    else:
        assert hit.sum() == 0


def test_degenerate_and_empty_soup(O):
    xs, ys, zs = reference_triangle_soup(8)
    hit, *_ = O.triangle_intersect(xs[3:4], ys[3:4], zs[3:4], *RAY_A)   # degenerate triangle
    assert hit[0] == 0
    idx, t = O.closest_hit(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32),
                           np.zeros((0, 4), np.float32), [RAY_A[0]], [RAY_A[1]])
    assert idx[0] == -1 and np.isinf(t[0])


# ---- survey anchors (reference sources executed on CPU) ---------------------------
def test_survey_anchor_sampler(O):
    a = PINS["survey_anchors"]
    assert O.halton_params(512, 512).tolist() == a["computeParams_512x512"]
    for res in (512, 1024, 4096):  # all configs give the same scales
        assert O.halton_params(res, res).tolist() == a["computeParams_512x512"]
    hi, _, _ = O.sampler_stream(512, 512, [17], [42], [3], 1)
    assert int(hi[0]) == a["halton_index_px17_py42_s3"]


@pytest.mark.parametrize("res,spp,key", [(64, 4, "film_mean_64x64_4spp"), (128, 16, "film_mean_128x128_16spp")])
def test_survey_anchor_film_means(O, res, spp, key):
    a = PINS["survey_anchors"]
    scene = O.cornell_box(res, res)
    mean, m2 = O.render(scene, spp, rtl_args=a["rtl_args"])
    got = mean[..., :3].mean(axis=(0, 1))
    assert np.isfinite(mean).all()
    assert np.allclose(got, a[key], atol=0.6e-4), (got, a[key])   # 4 recorded digits
    assert np.all(m2[..., 3] == spp)


def test_argument_order_matters(O):
    """The two sampler-argument orders give visibly different images; the HIP kernel and every
    golden film use left-to-right (= the CUDA build, pinned by the published figure)."""
    scene = O.cornell_box(64, 64)
    a, _ = O.render(scene, 4, rtl_args=False)
    b, _ = O.render(scene, 4, rtl_args=True)
    assert film_rmse(a, b) > 0.05


@pytest.mark.slow
def test_published_sqrt_mse_figure(O):
    """docs/notes.txt:36-37: mean of output-2048_sqrt_mse.png = 0.018148823657066993 (CUDA build).
    ~160 s on 8 cores; run with --runslow.  The GPU twin of this test runs by default."""
    p = PINS["published"]
    scene = O.cornell_box(256, 256)
    mean, m2 = O.render(scene, 2048)
    _, se = O.pixels_from_film(mean, m2)
    got = (se.astype(np.float64) / 255.0).mean(axis=2).mean()
    assert abs(got - p["avg_sqrt_mse_256x256_2048spp"]) < 2e-6


# ---- golden vectors (regression pins of the oracle itself) -------------------------------------
def test_golden_scene_packing(O):
    g = golden("cornell_scene.npz")
    sc = O.cornell_box()
    for k, v in (("xs", sc.xs), ("ys", sc.ys), ("zs", sc.zs), ("mat_id", sc.mat_id), ("bsdfs", sc.bsdfs),
                 ("lights", sc.lights), ("inf_lights", sc.inf_lights), ("camera", sc.camera)):
        assert np.array_equal(g[k], v), k
    # 2 spheres x 8 + 5 quads x 2 = 26 triangles, 7 materials, spot + env
    assert sc.tri_count == 26 and sc.bsdfs.shape[0] == 7
    assert sc.mat_id.tolist() == [0] * 8 + [1] * 8 + [2, 2, 3, 3, 4, 4, 5, 5, 6, 6]
    assert sc.lights.shape[0] == 1 and sc.inf_lights.shape[0] == 1
    # the octahedral encoder clamps before rounding: the spot direction packs to 0x00010001
    assert sc.lights[0, 20:24].view(np.uint32)[0] == 0x00010001


def test_golden_sampler_streams(O):
    g = golden("sampler_streams.npz")
    for res in (512, 1024, 4096):
        hi, p2, d = O.sampler_stream(res, res, g[f"r{res}_px"], g[f"r{res}_py"], g[f"r{res}_s"], 24)
        assert np.array_equal(hi, g[f"r{res}_hidx"])
        assert np.array_equal(p2, g[f"r{res}_pix2d"])
        assert np.array_equal(d, g[f"r{res}_dims"])
        # dimension wraps to 2 after 9 (rng.cu:236): a path only ever sees 8 distinct values
        assert np.array_equal(d[:, :8], d[:, 8:16]) and np.array_equal(d[:, :8], d[:, 16:24])
        assert (d >= 0).all() and (d < 1).all()


def test_golden_films(O):
    g = golden("films.npz")
    s64 = O.cornell_box(64, 64)
    m, v = O.render(s64, 4)
    assert np.array_equal(m, g["f64_spp4_mean"]) and np.array_equal(v, g["f64_spp4_m2"])
    m, v = O.render(s64, 16, max_depth=4)
    assert np.array_equal(m, g["f64_spp16_depth4_mean"])


def test_film_is_resumable_and_thread_invariant(O):
    """sampleOffset continues a render (megakernel.cu:57,103): 4 x 4 spp == 16 spp, bit for bit;
    the tile/thread schedule does not change a single bit (stateless sampler)."""
    scene = O.cornell_box(32, 32)
    full = O.render(scene, 16, threads=1)
    film = None
    for k in range(4):
        film = O.render(scene, 4, sample_offset=4 * k, film=film, threads=3)
    assert np.array_equal(full[0], film[0]) and np.array_equal(full[1], film[1])


def test_ragged_resolution_and_regions(O):
    """Non-square, non-multiple-of-8 image; a region render touches only its pixels."""
    scene = O.cornell_box(37, 21)
    mean, m2 = O.render(scene, 2)
    assert mean.shape == (21, 37, 4) and np.all(m2[..., 3] == 2)
    part = O.render(scene, 2, region=(5, 3, 20, 11))
    assert np.array_equal(part[0][3:11, 5:20], mean[3:11, 5:20])
    outside = np.ones((21, 37), bool)
    outside[3:11, 5:20] = False
    assert np.all(part[1][outside] == 0)


def test_half_codec_round_half_up(O):
    """CC/private/encoding.cu:93-115 adds half an ulp and truncates (ties go UP, not to even)."""
    L = O.lib()
    # 1 + 2^-11 is exactly between fp16(1.0)=0x3C00 and 0x3C01
    tie = np.float32(1.0 + 2.0 ** -11)
    assert L.oracle_float_to_half(float(tie)) == 0x3C01
    for h in (0x0000, 0x8000, 0x0001, 0x03FF, 0x0400, 0x3C00, 0x7BFF, 0xFBFF):
        assert L.oracle_float_to_half(L.oracle_half_to_float(h)) == h
    assert L.oracle_float_to_half(1e-10) == 0
    # overflow quirk (encoding.cu:88-91): finite values beyond fp16 range take the NaN/Inf branch, so a
    # non-zero mantissa gives the NaN pattern 0x7E00 and only exact powers of two give +inf
    assert L.oracle_float_to_half(1e6) == 0x7E00 and L.oracle_float_to_half(65536.0) == 0x7C00
