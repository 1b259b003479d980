"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and through size-independent properties at
larger sizes.  Integer / index work must be bit-exact; floating point within the stated tolerance.

Float tolerance: the north star asks for per-pixel RMSE < 1e-3 against the reference arithmetic at
equal sample indices (BASELINE config: 8 bounces).  The device build contracts a*b+c into FMAs and
uses the device libm, so single values differ from the oracle's by a few ulp.  Up to the config's
depth cap of 8 that stays at float-rounding level and the tests assert RMSE < 1e-4.

At the reference's own cap of 32 bounces a second effect dominates (DESIGN.md "Deep paths are
chaotic"): the sampler only has 8 dimensions and a bounce consumes 7, so from depth 3 on every bounce
of a path reuses the SAME random numbers; the walls of the box then act as a fixed sequence of
oblique projections that multiplies any position error by ~2 per bounce, and paths whose throughput
has a component equal to 1.0 (green wall) never enter Russian roulette and run to depth 32.  A 1e-7
rounding difference is O(1) after ~25 bounces, so ~1 % of the paths end on different triangles on
ANY two float implementations (the reference's own -use_fast_math CUDA build vs its host code
included).  For depth 32 the tests therefore assert (a) exact agreement of everything discrete,
(b) per-path agreement for the 98 %+ of paths that are not chaotic, (c) film differences far below
the estimator's own standard error, and (d) the published 2048-spp figure of the CUDA build.
"""
import json

import numpy as np
import pytest

from conftest import GOLDEN, film_rmse, golden
from test_oracle_pins import RAY_A, RAY_B, reference_triangle_soup

pytestmark = pytest.mark.gpu

PINS = json.loads((GOLDEN / "reference_pins.json").read_text())
RMSE_TOL = 1e-3          # north_star / BASELINE.json
REL = 2e-5               # per-value relative tolerance for function-level float comparisons


def close(a, b, rel=REL, abs_=1e-6):
    return np.allclose(a, b, rtol=rel, atol=abs_, equal_nan=True)


# ---- integer / index work: bit exact -------------------------------------------------------------
def test_half_codec_exhaustive(renderer, O):
    halves = np.arange(65536, dtype=np.uint16)
    _, f = renderer.test_half(halves=halves)
    L = O.lib()
    ref = np.array([L.oracle_half_to_float(int(h)) for h in halves], np.float32)
    assert np.array_equal(f.view(np.uint32), ref.view(np.uint32))
    rng = np.random.default_rng(1)
    x = np.concatenate([
        rng.standard_normal(50000).astype(np.float32),
        (rng.random(20000, dtype=np.float32) * 1e-4),
        (rng.standard_normal(5000) * 1e5).astype(np.float32),
        # exact ties between neighbouring halves (software codec rounds them UP)
        (1.0 + (2 * np.arange(512) + 1) * 2.0 ** -11).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65520.0, 65536.0, 1e6, 6e-8, 3e-8, 2.9e-8, 6.1e-5, np.inf, -np.inf], np.float32),
    ])
    h, _ = renderer.test_half(floats=x)
    ref = np.array([L.oracle_float_to_half(float(v)) for v in x], np.uint16)
    assert np.array_equal(h, ref)


@pytest.mark.parametrize("res", [512, 1024, 4096])
def test_sampler_bit_exact_vs_golden_and_oracle(renderer, O, res):
    g = golden("sampler_streams.npz")
    hi, p2, d = renderer.test_sampler(res, res, g[f"r{res}_px"], g[f"r{res}_py"], g[f"r{res}_s"], 24)
    assert np.array_equal(hi, g[f"r{res}_hidx"])
    assert np.array_equal(p2.view(np.uint32), g[f"r{res}_pix2d"].view(np.uint32))
    assert np.array_equal(d.view(np.uint32), g[f"r{res}_dims"].view(np.uint32))
    rng = np.random.default_rng(res)
    n = 4096
    pxs, pys = rng.integers(0, res, n), rng.integers(0, res, n)
    ss = rng.integers(0, 4096, n)
    a = renderer.test_sampler(res, res, pxs, pys, ss, 10)
    b = O.sampler_stream(res, res, pxs, pys, ss, 10)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_sampler_ragged_resolution(renderer, O):
    """width/height below 128 and non powers: scales stop early (rng.cu:194)."""
    for w, h in ((37, 21), (64, 64), (100, 300), (1, 1)):
        rng = np.random.default_rng(w * 1000 + h)
        n = 512
        pxs, pys, ss = rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 2000, n)
        a = renderer.test_sampler(w, h, pxs, pys, ss, 9)
        b = O.sampler_stream(w, h, pxs, pys, ss, 9)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (w, h)


# ---- the reference's own kernel test, on the GPU --------------------------------------------------
@pytest.mark.parametrize("ray", [RAY_A, RAY_B], ids=["rayA", "rayB"])
def test_reference_triangle_kat_on_gpu(renderer, O, ray):
    """T/tests/triangle_intersect.cu:164-186: device hit flags == host Moeller-Trumbore for all
    65,536 generated triangles (4 shapes incl. degenerate) and both rays."""
    xs, ys, zs = reference_triangle_soup(1 << 16)
    hit, t, pos, nrm, err = renderer.test_triangle_intersect(xs, ys, zs, *ray)
    expected = O.host_intersect_mt(xs, ys, zs, *ray)
    assert np.array_equal(hit, expected)
    ohit, ot, opos, onrm, oerr = O.triangle_intersect(xs, ys, zs, *ray)
    assert np.array_equal(hit, ohit)
    m = hit == 1
    assert close(t[m], ot[m]) and close(pos[m], opos[m]) and close(nrm[m], onrm[m]) and close(err[m], oerr[m], abs_=1e-12)


def test_triangle_intersect_random_soup(renderer, O):
    rng = np.random.default_rng(7)
    n = 20000
    c = rng.uniform(-1, 1, (n, 3))
    v = c[:, None, :] + rng.uniform(-0.5, 0.5, (n, 3, 3))
    xs = np.zeros((n, 4), np.float32); ys = np.zeros((n, 4), np.float32); zs = np.zeros((n, 4), np.float32)
    xs[:, :3], ys[:, :3], zs[:, :3] = v[..., 0], v[..., 1], v[..., 2]
    o = np.array([0.1, -3.0, 0.2], np.float32)
    d = np.array([0.05, 1.0, -0.02], np.float32); d /= np.linalg.norm(d)
    hit, t, pos, nrm, err = renderer.test_triangle_intersect(xs, ys, zs, o, d)
    ohit, ot, opos, onrm, oerr = O.triangle_intersect(xs, ys, zs, o, d)
    # a ray within float rounding of an edge may flip; count them (none expected at this size)
    assert (hit != ohit).sum() <= 2
    m = (hit == 1) & (ohit == 1)
    assert m.sum() > 100
    assert close(t[m], ot[m], rel=1e-4) and close(pos[m], opos[m], rel=1e-4, abs_=1e-5)
    assert close(nrm[m], onrm[m]) and close(err[m], oerr[m], rel=1e-3, abs_=1e-10)


def test_closest_hit_lowest_index_on_ties(renderer, O):
    """Duplicate triangles: the strict `<` of megakernel.cu:126 keeps the lowest index."""
    g = golden("cornell_scene.npz")
    xs = np.concatenate([g["xs"], g["xs"]]); ys = np.concatenate([g["ys"], g["ys"]]); zs = np.concatenate([g["zs"], g["zs"]])
    mat = np.concatenate([g["mat_id"], g["mat_id"]])
    renderer.upload_triangles(xs, ys, zs, mat)
    rng = np.random.default_rng(3)
    n = 4096
    o = np.tile(np.array([0, 0.1, 0.3], np.float32), (n, 1))
    d = rng.normal(size=(n, 3)).astype(np.float32); d[:, 1] = np.abs(d[:, 1]) + 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    idx, t = renderer.test_closest_hit(o, d)
    oidx, ot = O.closest_hit(xs, ys, zs, o, d)
    assert (idx != oidx).sum() <= 2
    assert idx.max() < 26            # never the duplicate
    m = idx == oidx
    assert close(t[m], ot[m], rel=1e-5)


# ---- function-level float parity -------------------------------------------------------------------
@pytest.mark.parametrize("res", [64, 512, 1024])
def test_camera_rays(renderer, O, res):
    g = golden("camera_rays.npz")
    scene = O.cornell_box(res, res)
    renderer.set_camera(scene.camera)
    o, d = renderer.test_camera_rays(g[f"r{res}_px"], g[f"r{res}_py"], g[f"r{res}_s"])
    assert close(o, g[f"r{res}_o"]) and close(d, g[f"r{res}_d"], rel=1e-6, abs_=2e-7)
    rng = np.random.default_rng(res)
    pxs, pys, ss = rng.integers(0, res, 2048), rng.integers(0, res, 2048), rng.integers(0, 1024, 2048)
    o, d = renderer.test_camera_rays(pxs, pys, ss)
    oo, od = O.camera_rays(scene, pxs, pys, ss)
    assert close(o, oo) and close(d, od, rel=1e-6, abs_=2e-7)


def _decode_prepared(rec32, n):
    """weight / multiScatter / energyScale of the oracle's prepared 32-byte records."""
    rec = rec32.reshape(n, 32)
    h = rec.view(np.uint16).reshape(n, 16)
    f = rec.view(np.float32).reshape(n, 8)
    weight = h[:, 0:3].view(np.float16).astype(np.float32)
    ms = h[:, 7:10].view(np.float16).astype(np.float32)
    return weight, ms, f[:, 2]


BSDF_NAMES = [f"cornell{i}" for i in range(7)] + ["gold", "gold_aniso", "lambert", "glass_iso"]


@pytest.mark.parametrize("name", BSDF_NAMES)
def test_bsdf_prepare_sample_eval(renderer, name):
    g = golden("bsdf_lattice.npz")
    n = g["ns"].shape[0]
    prep, samp, ev = renderer.test_bsdf(g[f"{name}_rec"], g["ns"], g["wo"], g["u2"], g["uc"], g["wi"])
    gw, gms, gescale = _decode_prepared(g[f"{name}_prepared"], n)
    btype = int(g[f"{name}_rec"].view(np.uint16)[3])
    # fp16-quantised terms: equal up to one fp16 ulp on the rare rounding-boundary case
    assert np.mean(prep[:, 0:3] != gw) < 0.02 and close(prep[:, 0:3], gw, rel=1.1e-3)
    if btype == 0:
        assert np.mean(prep[:, 3:6] != gms) < 0.02 and close(prep[:, 3:6], gms, rel=1.1e-3, abs_=1e-7)
    if btype in (1, 2):
        assert close(prep[:, 6], gescale, rel=1e-5)
    gs, ge = g[f"{name}_sample"], g[f"{name}_eval"]
    # discrete outputs (lobe choice, delta flag, validity) must agree except on knife-edge cases
    flags_equal = (samp[:, 8:10] == gs[:, 8:10]).all(axis=1) & ((samp[:, 6] != 0) == (gs[:, 6] != 0))
    assert flags_equal.mean() > 0.99
    m = flags_equal
    # wi / f / pdf inherit the fp16 quantisation of the weights only through f: 2e-3 relative
    assert close(samp[m, 0:3], gs[m, 0:3], rel=1e-4, abs_=2e-6)
    assert close(samp[m, 3:6], gs[m, 3:6], rel=2e-3, abs_=1e-6)
    assert close(samp[m, 6], gs[m, 6], rel=1e-4, abs_=1e-7) and close(samp[m, 7], gs[m, 7])
    assert close(ev, ge, rel=2e-3, abs_=1e-6)


LIGHT_NAMES = ["cornell_spot", "wide_spot", "point_small", "point_big", "directional", "env"]


@pytest.mark.parametrize("name", LIGHT_NAMES)
def test_light_sample_eval(renderer, name):
    g = golden("light_lattice.npz")
    out = renderer.test_light(g[f"{name}_rec"], g["pos"], g["nrm"], g["u2"], g["hadt"])
    ref = g[f"{name}_out"]
    valid_equal = out[:, 13] == ref[:, 13]
    assert valid_equal.mean() > 0.99
    m = valid_equal & (ref[:, 13] == 1)
    assert m.sum() > 0
    # point_small: distance = d cos - sqrt(r^2 - d^2 + d^2 cos^2) cancels catastrophically for r << d
    # (light.cu:56-59), so its distance / Le only agree to ~r/d
    rel_d = 5e-3 if name == "point_small" else 1e-4
    assert close(out[m, 3:6], ref[m, 3:6], rel=1e-4, abs_=2e-6)                # direction
    assert close(out[m, 6:8], ref[m, 6:8], rel=1e-4)                             # pdf, delta
    assert close(out[m, 8], ref[m, 8], rel=rel_d) and close(out[m, 9], ref[m, 9])  # distance, factor
    assert close(out[m, 10:13], ref[m, 10:13], rel=2 * rel_d, abs_=1e-7)         # Le
    finite = np.isfinite(ref[m, 0:3]).all(axis=1)
    assert close(out[m][finite, 0:3], ref[m][finite, 0:3], rel=rel_d, abs_=1e-5)  # pLight


# ---- per-path and film parity ---------------------------------------------------------------------
def _load_cornell(renderer, O, w, h, max_depth=32):
    scene = O.cornell_box(w, h)
    renderer.upload_scene(scene)
    renderer.set_limits(max_depth)
    renderer.set_partition(0, 1)
    renderer.film_clear()
    return scene


def test_path_radiance_samples(renderer, O):
    g = golden("path_samples.npz")
    _load_cornell(renderer, O, 64, 64)
    L = renderer.test_trace_samples(g["px"], g["py"], g["s"])
    ref = g["L"]
    same = np.isclose(L, ref, rtol=1e-3, atol=1e-5).all(axis=1)
    assert same.mean() >= 0.96, f"{(~same).sum()} of {len(same)} paths diverge"   # ~1.3 % are chaotic
    assert np.isfinite(L).all()


@pytest.mark.parametrize("w,h,spp,depth", [(64, 64, 4, 32), (64, 64, 64, 32), (64, 64, 16, 4), (37, 21, 8, 32), (128, 128, 16, 8)])
def test_film_vs_oracle(renderer, O, w, h, spp, depth):
    scene = _load_cornell(renderer, O, w, h, depth)
    renderer.render(spp)
    mean, m2 = renderer.download_film()
    rmean, rm2 = O.render(scene, spp, max_depth=depth)
    assert np.isfinite(mean).all() and np.isfinite(m2).all()
    assert np.array_equal(m2[..., 3], rm2[..., 3])                 # sample counts: exact
    assert np.all(mean[..., 3] == 0)
    rmse = film_rmse(mean, rmean)
    if depth <= 8:
        assert rmse < RMSE_TOL, rmse
        assert rmse < 1e-4, rmse                                    # observed ~1e-6
        bad = np.abs(mean[..., :3] - rmean[..., :3]).max(axis=2) > 1e-4
        assert bad.mean() < 0.01, bad.mean()
        assert film_rmse(np.sqrt(np.maximum(m2, 0)), np.sqrt(np.maximum(rm2, 0))) < 1e-2
    else:
        # depth 32: chaotic paths (module docstring).  The difference must be far below the
        # estimator's own standard error sqrt(M2)/N and must not shift the image mean.
        stderr = np.sqrt(np.maximum(rm2[..., :3], 0)) / np.maximum(rm2[..., 3:4], 1)
        assert rmse < 0.05 * float(stderr.mean()), (rmse, float(stderr.mean()))
        assert rmse < 3 * RMSE_TOL, rmse
        bias = np.abs(mean[..., :3].mean(axis=(0, 1)) - rmean[..., :3].mean(axis=(0, 1))).max()
        assert bias < 0.02 * float(stderr.mean()), (bias, float(stderr.mean()))
        # the non-chaotic majority of pixels still agrees closely (a pixel is "off" as soon as ONE
        # of its spp samples is a chaotic path, ~1.3 % of the samples)
        agree = np.abs(mean[..., :3] - rmean[..., :3]).max(axis=2) < 1e-4
        assert agree.mean() > (0.9 if spp <= 4 else 0.8 if spp <= 8 else 0.35), agree.mean()


def test_film_vs_golden(renderer, O):
    g = golden("films.npz")
    _load_cornell(renderer, O, 64, 64)
    renderer.render(64)
    mean, _ = renderer.download_film()
    assert film_rmse(mean, g["f64_spp64_mean"]) < 3 * RMSE_TOL       # depth 32, see module docstring
    _load_cornell(renderer, O, 64, 64, 4)
    renderer.render(16)
    mean, _ = renderer.download_film()
    assert film_rmse(mean, g["f64_spp16_depth4_mean"]) < 1e-4
    _load_cornell(renderer, O, 512, 512)
    renderer.render(4, region=(0, 200, 512, 232))
    mean, m2 = renderer.download_film()
    assert film_rmse(mean[200:232], g["f512_band_spp4_mean"]) < 3 * RMSE_TOL
    assert np.all(m2[:200, :, 3] == 0) and np.all(m2[232:, :, 3] == 0)   # untouched outside the region


def test_empty_and_degenerate_inputs(renderer, O):
    scene = _load_cornell(renderer, O, 16, 16)
    renderer.render(0)                                   # zero samples: no-op
    renderer.render(4, region=(5, 5, 5, 9))              # empty region: no-op
    mean, m2 = renderer.download_film()
    assert not mean.any() and not m2.any()
    # empty scene: every ray misses, film = environment colour exactly
    renderer.upload_triangles(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32),
                              np.zeros((0, 4), np.float32), np.zeros(0, np.uint32))
    renderer.render(3)
    mean, m2 = renderer.download_film()
    env = np.float16(0.1).astype(np.float32)
    assert np.all(mean[..., :3] == env) and np.all(m2[..., 3] == 3) and np.all(m2[..., :3] == 0)
    # depth cap 0: first hit terminates with no light gathered
    renderer.upload_scene(scene)
    renderer.set_limits(0)
    renderer.film_clear()
    renderer.render(2)
    mean, _ = renderer.download_film()
    rmean, _ = O.render(scene, 2, max_depth=0)
    assert film_rmse(mean, rmean) < 1e-6
    renderer.set_limits(32)


def test_error_behaviour(renderer, pkg, O):
    scene = O.cornell_box(16, 16)
    with pkg.Renderer(0) as r2:
        with pytest.raises(pkg.DmtError):          # render before upload
            r2.render(1)
        r2.upload_scene(scene)
        bad = scene.mat_id.copy(); bad[3] = 99
        r2.upload_triangles(scene.xs, scene.ys, scene.zs, bad)
        with pytest.raises(pkg.DmtError):          # material index outside the BSDF array
            r2.render(1)
        with pytest.raises(pkg.DmtError):          # sample index beyond the 32-bit Halton index
            r2.upload_triangles(scene.xs, scene.ys, scene.zs, scene.mat_id)
            r2.render(10, sample_offset=6000000)
        with pytest.raises(pkg.DmtError):
            r2.set_partition(2, 2)


# ---- size-independent properties at larger sizes --------------------------------------------------
def test_resumable_bit_exact(renderer, O):
    """kspp batching (main.cu:141-155): 8 launches of 8 spp give the same bits as one of 64."""
    _load_cornell(renderer, O, 256, 256)
    renderer.render(64)
    a = renderer.download_film()
    renderer.film_clear()
    for k in range(8):
        renderer.render(8, sample_offset=8 * k)
    b = renderer.download_film()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_tile_partition_is_exact(renderer, O):
    """Multi-GPU split: rendering the tile sets of ranks 0..3 one after the other into one film
    (disjoint pixels) equals the single-context film bit for bit; so does a sum of separate films."""
    _load_cornell(renderer, O, 200, 120)
    renderer.render(8)
    full = renderer.download_film()
    renderer.film_clear()
    parts = []
    for rank in range(4):
        renderer.set_partition(rank, 4)
        renderer.film_clear()
        renderer.render(8)
        parts.append(renderer.download_film())
    renderer.set_partition(0, 1)
    counts = sum(p[1][..., 3] for p in parts)
    assert np.all(counts == 8)                                    # each pixel owned exactly once
    from cuda_optix_pathtracing_amd import multigpu               # host-side mirror of the kernel's tile map
    for rank in range(4):
        assert np.array_equal(parts[rank][1][..., 3] > 0, multigpu.owned_pixel_mask(200, 120, rank, 4))
    assert np.array_equal(sum(p[0] for p in parts), full[0])      # x + 0 == x: exact gather by sum
    assert np.array_equal(sum(p[1] for p in parts), full[1])


@pytest.mark.parametrize("accel", [0, 1])
def test_concurrent_contexts_partition_is_exact(renderer, pkg, O, accel):
    """Multi-GPU rehearsal on one device: one context per rank (dmt_set_partition(r, W)), each with its OWN stream and
    film, launched back to back without synchronising in between, so their persistent kernels, work counters, tile
    counters and staging areas are live at the same time.  Each launch is sized to a fraction of the machine (items <
    resident waves), so the kernels really co-reside.  Sum of the films == the single-context film, bit for bit; each
    pixel is owned by exactly one rank; repeated with several passes (resumable films per rank)."""
    W = 3
    w, h, spp = 256, 192, 16
    scene = O.cornell_box(w, h) if accel == 0 else pkg.host_scene.random_triangle_scene(5000, width=w, height=h)
    renderer.upload_scene(scene)
    renderer.set_limits(8)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    renderer.film_clear()
    for k in range(2):
        renderer.render(spp, sample_offset=k * spp)
    renderer.sync()
    full = renderer.download_film()
    renderer.set_accel(0)
    ranks = [pkg.Renderer(0) for _ in range(W)]
    try:
        for r, ctx in enumerate(ranks):
            ctx.upload_scene(scene)
            ctx.set_limits(8)
            ctx.set_accel(accel)
            ctx.set_partition(r, W)
            ctx.film_clear()
        for k in range(2):
            for ctx in ranks:                      # asynchronous launches on W different streams
                ctx.render(spp, sample_offset=k * spp)
        for ctx in ranks:
            ctx.sync()                             # raises unless every sample chunk was folded exactly once
        parts = [ctx.download_film() for ctx in ranks]
    finally:
        for ctx in ranks:
            ctx.close()
    counts = sum(p[1][..., 3] for p in parts)
    assert np.all(counts == 2 * spp)
    for r in range(W):
        assert np.array_equal(parts[r][1][..., 3] > 0, pkg.multigpu.owned_pixel_mask(w, h, r, W))
    assert np.array_equal(sum(p[0] for p in parts), full[0])
    assert np.array_equal(sum(p[1] for p in parts), full[1])


def test_region_split_is_exact(renderer, O):
    _load_cornell(renderer, O, 96, 80)
    renderer.render(4)
    full = renderer.download_film()
    renderer.film_clear()
    for reg in ((0, 0, 50, 33), (50, 0, 96, 33), (0, 33, 96, 80)):
        renderer.render(4, region=reg)
    split = renderer.download_film()
    assert np.array_equal(full[0], split[0]) and np.array_equal(full[1], split[1])


def test_published_sqrt_mse_figure_on_gpu(renderer, O):
    """docs/notes.txt:36-37: `dmt-mk v2 ... 0.018148823657066993` = scripts/rmse.py default mode on
    the CUDA build's output-2048_sqrt_mse.png (256x256, 2048 spp, kspp 4, depth 32).  The HIP path
    must land on the same figure: this ties the whole pipeline (scene, sampler, BSDFs, lights, film,
    8-bit writer) to the reference's own GPU output."""
    p = PINS["published"]
    _load_cornell(renderer, O, 256, 256)
    for k in range(8):
        renderer.render(256, sample_offset=256 * k)
    mean, m2 = renderer.download_film()
    assert np.all(m2[..., 3] == 2048)
    _, se = O.pixels_from_film(mean, m2)
    got = (se.astype(np.float64) / 255.0).mean(axis=2).mean()
    assert abs(got - p["avg_sqrt_mse_256x256_2048spp"]) < 2e-6, got
    assert abs(got - p["oracle_ltr_measured"]) < 2e-6, got


def test_full_size_invariants(renderer, O):
    """BASELINE config 2 resolution (1024x1024), reduced spp: finite, correct counts, mean matches
    the oracle on a sampled band, image statistics match the 256x256 render."""
    scene = _load_cornell(renderer, O, 1024, 1024, 8)
    renderer.render(16)
    mean, m2 = renderer.download_film()
    assert np.isfinite(mean).all() and np.all(m2[..., 3] == 16) and (m2[..., :3] >= 0).all()
    rmean, _ = O.render(scene, 16, max_depth=8, region=(0, 500, 1024, 516))
    assert film_rmse(mean[500:516], rmean[500:516]) < 1e-4


# ---- BVH (DMT_ACCEL_BVH): must reproduce brute force exactly ---------------------------------------
def _random_soup(n, seed, spread=3.0, size=0.35):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, (n, 1, 3)) + np.array([0, 6.0, 0])
    v = c + rng.uniform(-size, size, (n, 3, 3))
    xs = np.zeros((n, 4), np.float32); ys = np.zeros((n, 4), np.float32); zs = np.zeros((n, 4), np.float32)
    xs[:, :3], ys[:, :3], zs[:, :3] = v[..., 0], v[..., 1], v[..., 2]
    return xs, ys, zs


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:, 1] = np.abs(d[:, 1]) + 0.3
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("ntri", [1, 5, 26, 777, 20000])
def test_bvh_closest_hit_equals_brute_force(renderer, O, ntri):
    xs, ys, zs = _random_soup(ntri, ntri)
    if ntri == 777:          # duplicates: ties must resolve to the lowest ORIGINAL index
        xs[400:] = xs[:377]; ys[400:] = ys[:377]; zs[400:] = zs[:377]
    renderer.upload_triangles(xs, ys, zs, np.zeros(ntri, np.uint32))
    o, d = _rays(8192, ntri + 1)
    renderer.set_accel(0)
    bi, bt = renderer.test_closest_hit(o, d)
    renderer.set_accel(1)
    ai, at = renderer.test_closest_hit(o, d)
    renderer.set_accel(0)
    assert np.array_equal(ai, bi)
    assert np.array_equal(at.view(np.uint32), bt.view(np.uint32))      # same routine on the same record: bit equal
    oi, ot = O.closest_hit(xs, ys, zs, o[:2048], d[:2048])
    assert (ai[:2048] != oi).sum() <= 1                               # vs the CPU oracle: float-rounding edge cases only
    if ntri >= 777:
        assert (ai >= 0).mean() > 0.02
    if ntri == 777:
        assert ai.max() < 400 or (ai[ai >= 377] < 400).all()          # never the duplicate copy


def test_bvh_empty_slots_axis_parallel_rays(renderer, O):
    """A flat floor (one triangle pair -> a root with one leaf and three EMPTY slots) under rays exactly parallel to an
    axis: the other two axes drop out of the slab test as NaN, and on the floor's flat axis 255 quantisation steps
    vanish against the plane distance, so the inverted boxes of the empty slots "hit" (near == far).  Their implicit
    references must stay inside the pair array (guard pairs) and the result must equal brute force."""
    def tri(p0, p1, p2):
        return [p0[0], p1[0], p2[0], 0.0], [p0[1], p1[1], p2[1], 0.0], [p0[2], p1[2], p2[2], 0.0]
    for nfloor in (1, 2, 3):                                   # 1..3 triangles: the last node's leaves end the array
        t = [tri((-1, 0, -1), (1, 0, -1), (1, 0, 1)), tri((-1, 0, -1), (1, 0, 1), (-1, 0, 1)), tri((2, 0, 2), (3, 0, 2), (3, 0, 3))][:nfloor]
        xs = np.array([a[0] for a in t], np.float32).reshape(-1); ys = np.array([a[1] for a in t], np.float32).reshape(-1)
        zs = np.array([a[2] for a in t], np.float32).reshape(-1)
        renderer.upload_triangles(xs, ys, zs, np.zeros(nfloor, np.uint32))
        g = np.linspace(-1.5, 3.5, 41, dtype=np.float32)
        gx, gz = np.meshgrid(g, g)
        n = gx.size
        o = np.stack([gx.ravel(), np.full(n, 5.0, np.float32), gz.ravel()], axis=1).astype(np.float32)
        d = np.tile(np.array([0.0, -1.0, 0.0], np.float32), (n, 1))
        # the same from below, and rays along x / z skimming over the floor (miss) and in its plane
        o = np.concatenate([o, o * np.array([1, -1, 1], np.float32), np.stack([np.full(n, -9.0, np.float32), gz.ravel() * 0, gx.ravel()], axis=1)])
        d = np.concatenate([d, -d, np.tile(np.array([1.0, 0.0, 0.0], np.float32), (n, 1))])
        renderer.set_accel(0)
        bi, bt = renderer.test_closest_hit(o, d)
        renderer.set_accel(1)
        ai, at = renderer.test_closest_hit(o, d)
        renderer.set_accel(0)
        assert np.array_equal(ai, bi) and np.array_equal(at.view(np.uint32), bt.view(np.uint32))
        assert (ai >= 0).sum() > 100


def test_bvh_film_bit_exact_vs_brute_force(renderer, pkg, O):
    """Whole path tracer through the BVH == brute force, bit for bit (closest hits, shadow rays, ties)."""
    for scene, res, spp in ((pkg.host_scene.cornell_box(64, 64), 64, 16),
                            (pkg.host_scene.random_triangle_scene(4000, width=48, height=48), 48, 4)):
        renderer.upload_scene(scene)
        renderer.set_limits(8)
        films = []
        for mode in (0, 1):
            renderer.set_accel(mode)
            renderer.film_clear()
            renderer.render(spp)
            films.append(renderer.download_film())
        renderer.set_accel(0)
        assert np.array_equal(films[0][0], films[1][0]) and np.array_equal(films[0][1], films[1][1])
        assert films[0][0][..., :3].max() > 0


def test_bvh_million_triangles(renderer, pkg, O):
    """BASELINE config 4 scene (1 M random triangles): BVH == brute force, bit for bit, on two 64 x 64 windows x 4 spp
    at the config's bounce cap of 8 (brute force costs 1 M tests per ray: ~2e11 tests), and vs the CPU oracle on 32
    camera rays."""
    scene = pkg.host_scene.random_triangle_scene(1_000_000, width=1024, height=1024)
    assert pkg.bvh_validate(scene.xs, scene.ys, scene.zs)["ok"]
    renderer.upload_scene(scene)
    renderer.set_limits(8)
    for region in ((480, 480, 544, 544), (3, 950, 67, 1014)):   # centre of the frame; a window off the tile grid near a corner
        films = []
        for mode in (0, 1):
            renderer.set_accel(mode)
            renderer.film_clear()
            renderer.render(4, region=region)
            renderer.sync()
            films.append(renderer.download_film())
        assert np.array_equal(films[0][0], films[1][0]) and np.array_equal(films[0][1], films[1][1])
        x0, y0, x1, y1 = region
        assert np.all(films[1][1][y0:y1, x0:x1, 3] == 4) and films[1][1][..., 3].sum() == 4 * 64 * 64
        assert films[1][0][..., :3].max() > 0
    o, d = _rays(32, 99)
    o[:] = 0
    ai, at = renderer.test_closest_hit(o, d)            # accel still BVH
    oi, ot = O.closest_hit(scene.xs, scene.ys, scene.zs, o, d)
    assert np.array_equal(ai, oi)
    # a larger BVH-only render stays finite and fully sampled
    renderer.film_clear()
    renderer.render(4, region=(256, 256, 768, 768))
    renderer.sync()
    mean, m2 = renderer.download_film()
    assert np.isfinite(mean).all() and np.all(m2[256:768, 256:768, 3] == 4)
    renderer.set_accel(0)


@pytest.mark.parametrize("kind", ["plain", "env", "area", "env_area", "partition"])
def test_bvh_wavefront_equals_megakernel(renderer, pkg, O, kind):
    """DMT_ACCEL_BVH launches run as a megakernel (small launches) or as the device-side wavefront of csrc/wavefront.hpp
    (generate / trace / shade / fold kernels over path-state arrays).  Same samples, same arithmetic (the wavefront's
    shading IS the megakernel's path_shade), same fold order: the films must be bit-identical, for every light kind, for
    passes that split tiles as well as samples, with a region off the tile grid, with a tile partition, and resumable."""
    spp, depth, region, part = 24, 8, None, (0, 1)
    if kind == "plain":
        sc = pkg.host_scene.random_triangle_scene(6000, width=200, height=136)
        region = (3, 5, 197, 131)
    elif kind == "partition":
        sc = pkg.host_scene.random_triangle_scene(6000, width=200, height=136)
        part = (1, 3)
    elif kind == "env":
        sc = pkg.host_scene.sphere_envmap_scene(96, 96, lat=8, lon=16, env_height=16)
    else:
        sc = O.cornell_box(72, 72)
        sc.lights = sc.lights[:0] if kind == "area" else sc.lights
        for a in (sc.xs, sc.ys, sc.zs):
            a[[0, 1, 16, 17]] = a[[0, 1, 16, 17]][:, [0, 2, 1, 3]]
        sc.set_area_lights([0, 1, 16, 17, 20, 21], [[6, 6, 5], [6, 6, 5], [12, 14, 20], [12, 14, 20], [20, 15, 10], [20, 15, 10]])
        if kind == "env_area":
            sc.env_rgb, sc.env_quat, sc.env_scale = pkg.host_scene.synthetic_sky(16), np.array([0, 0, 0, 1], np.float32), 1.0
        depth = 6
    renderer.upload_scene(sc)
    renderer.set_limits(depth)
    renderer.set_accel(1)
    renderer.set_partition(*part)
    films = []
    try:
        for strategy, paths in ((1, 0), (2, 1 << 22), (2, 20000)):   # megakernel; wavefront in one pass; in many (tile x sample) passes
            renderer.set_bvh_strategy(strategy, paths)
            renderer.film_clear()
            renderer.render(spp - 7, region=region)
            renderer.render(7, sample_offset=spp - 7, region=region)       # resumable
            renderer.sync()
            films.append(renderer.download_film())
    finally:
        renderer.set_bvh_strategy(0, 1 << 22)
        renderer.set_accel(0)
        renderer.set_partition(0, 1)
        renderer.clear_envmap()
        renderer.upload_area_lights([], np.zeros((0, 3), np.float32))
    assert films[0][0][..., :3].max() > 0 and films[0][1][..., 3].max() == spp
    for f in films[1:]:
        assert np.array_equal(films[0][1][..., 3], f[1][..., 3])
        assert np.array_equal(films[0][0], f[0]) and np.array_equal(films[0][1], f[1])


def test_bvh_overflow_stack_variant(renderer):
    """The traversal stack keeps 16 entries per lane in LDS and the rest in a global overflow area that ordinary scenes
    rarely reach.  csrc/variants/libdmt_hip_stack2.so is the same library compiled with 2 LDS entries: every non-trivial
    traversal then runs through the overflow path.  Closest hits and whole films must still equal brute force bit for
    bit, and the device counters must show that overflow pushes happened.  Runs in a subprocess (the library is
    selected at load time through DMT_HIP_LIB)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    lib = root / "cuda-optix-pathtracing_amd" / "csrc" / "variants" / "libdmt_hip_stack2.so"
    assert lib.exists(), f"{lib} is missing: run __graft_entry__.build()"
    env = dict(os.environ, DMT_HIP_LIB=str(lib))
    p = subprocess.run([sys.executable, str(root / "tests" / "_bvh_worker.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["lib"] == str(lib)
    assert out["closest_equal"] and out["film_equal"], out
    assert out["hit_share"] > 0.02 and out["film_max"] > 0
    assert out["overflow_pushes"] > 0.05 * out["node_visits"], out      # the overflow path really ran
    # the default build on the same scene: the 16 LDS entries suffice (the counter is the same code)
    assert (root / "cuda-optix-pathtracing_amd" / "csrc" / "libdmt_hip.so").exists()


def test_bvh_shade_threshold_is_scheduling_only(pkg, monkeypatch):
    """The BVH megakernel shades when `shadeThreshold` lanes of a wave have finished their rays; the library picks the
    value by tree size (bvhShadeThreshold) and DMT_BVH_SHADE_THRESHOLD overrides it.  Scheduling only: films must not
    change by a bit between 1 (shade at once), the defaults and 64 (shade only when every lane waits), and must
    equal brute force."""
    scene = pkg.host_scene.random_triangle_scene(30000, width=96, height=64)
    films = []
    for thr in (None, "1", "17", "64"):
        if thr is None:
            monkeypatch.delenv("DMT_BVH_SHADE_THRESHOLD", raising=False)
        else:
            monkeypatch.setenv("DMT_BVH_SHADE_THRESHOLD", thr)
        with pkg.Renderer(0) as r:
            r.upload_scene(scene)
            r.set_limits(8)
            r.set_accel(1)
            r.render(24)
            r.sync()
            films.append(r.download_film())
    monkeypatch.delenv("DMT_BVH_SHADE_THRESHOLD", raising=False)
    with pkg.Renderer(0) as r:
        r.upload_scene(scene)
        r.set_limits(8)
        r.set_accel(0)
        r.render(24)
        r.sync()
        brute = r.download_film()
    assert brute[0][..., :3].max() > 0 and brute[1][..., 3].max() == 24
    for f in films:
        assert np.array_equal(brute[0], f[0]) and np.array_equal(brute[1], f[1])


def test_row_band_scheduling_is_bit_exact(pkg, O, monkeypatch):
    """A GPU that owns fewer tiles than it has resident waves schedules row bands of its tiles (2 or 4 items per
    8x8 tile).  Scheduling only: the film must not change by a bit, partial tiles and partitions included."""
    scene = O.cornell_box(200, 136)
    films = []
    for shift in ("0", "1", "2"):
        monkeypatch.setenv("DMT_SUB_SHIFT", shift)
        with pkg.Renderer(0) as r:
            r.upload_scene(scene)
            r.set_limits(8)
            r.set_partition(1, 3)
            r.render(40, region=(3, 5, 197, 131))
            r.sync()
            films.append(r.download_film())
    for f in films[1:]:
        assert np.array_equal(films[0][0], f[0]) and np.array_equal(films[0][1], f[1])
    assert films[0][1][..., 3].max() == 40


def test_sample_chunking_is_bit_exact(renderer, O):
    """In-launch sample chunks (work items = tile x chunk, folded in chunk order per tile through the hand-over words)
    are a scheduling knob only: any chunk size gives the same film bits as one item per tile."""
    _load_cornell(renderer, O, 200, 136, 8)
    films = []
    for chunk in (0, 1, 7, 16, 64, 1000):
        renderer.set_chunk(chunk)
        renderer.film_clear()
        renderer.render(96)
        renderer.sync()
        films.append(renderer.download_film())
    renderer.set_chunk(0)
    for f in films[1:]:
        assert np.array_equal(films[0][0], f[0]) and np.array_equal(films[0][1], f[1])
    assert np.all(films[0][1][..., 3] == 96)


@pytest.mark.parametrize("accel", [0, 1])
def test_fold_handover_small_frame(renderer, pkg, O, accel):
    """The ordered fold without waiting (dmt_hip.hip: item_complete / fold_chain).  A 64 x 64 frame is 64 tiles for ~4 000
    resident waves, so with 1-sample chunks hundreds of chunks of every tile are traced at the same time and finish out
    of order: most of them are HANDED OVER to the wave that folds their predecessor, which folds them in a chain.  The
    film must equal the one-item-per-tile film bit for bit, every launched chunk must be folded exactly once, and no wave
    may have left the launch for want of a staging slab."""
    w, h, spp = 64, 64, 256
    scene = O.cornell_box(w, h) if accel == 0 else pkg.host_scene.random_triangle_scene(3000, width=w, height=h)
    renderer.upload_scene(scene)
    renderer.set_limits(8)
    renderer.set_accel(accel)
    renderer.set_chunk(1000)                   # one item per tile: no hand-over word is ever used
    renderer.film_clear()
    renderer.sched_diag(reset=True)
    renderer.render(spp)
    ref = renderer.download_film()
    d0 = renderer.sched_diag(reset=True)
    assert d0["handed_over"] == 0 and d0["folds"] == d0["launched"]
    for chunk in (1, 2, 5):
        renderer.set_chunk(chunk)
        renderer.film_clear()
        renderer.render(spp)
        got = renderer.download_film()
        d = renderer.sched_diag(reset=True)
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), chunk
        assert d["folds"] == d["launched"] > 0, d
        assert d["handed_over"] > d["launched"] // 4, d       # the hand-over path really ran ...
        assert d["folded_for_others"] == d["handed_over"], d  # ... and every handed-over chunk was folded by a chain
        assert d["early_exits"] == 0, d
    renderer.set_chunk(0)
    renderer.set_accel(0)


def test_cli_writes_reference_named_pngs(renderer, O, tmp_path):
    """The dmt-megakernel-compatible executable (host/main.cpp): same flags, same output names, and the
    8-bit images equal the oracle film put through the reference's quantisation."""
    import subprocess
    from pathlib import Path
    from PIL import Image
    exe = Path(__file__).resolve().parent.parent / "cuda-optix-pathtracing_amd" / "host" / "dmt-megakernel-hip"
    assert exe.exists(), "run __graft_entry__.build()"
    out = subprocess.run([str(exe), "--width", "48", "--height", "40", "--spp", "12", "--kspp", "4", "--max-depth", "6",
                          "--out", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Parsed Configuration" in out.stdout and "Total Execution Time" in out.stdout
    img = np.asarray(Image.open(tmp_path / "output-12.png").convert("RGB"))
    se = np.asarray(Image.open(tmp_path / "output-12_sqrt_mse.png").convert("RGB"))
    scene = O.cornell_box(48, 40)
    mean, m2 = O.render(scene, 12, max_depth=6)
    a, b = O.pixels_from_film(mean, m2)
    assert img.shape == (40, 48, 3)
    # 8-bit truncation: a float difference of 1e-6 can move a value across an integer boundary
    assert (np.abs(img.astype(int) - a.astype(int)) <= 1).all() and (img != a).mean() < 0.01
    assert (np.abs(se.astype(int) - b.astype(int)) <= 1).all() and (se != b).mean() < 0.02
    bad = subprocess.run([str(exe), "--spp", "2", "--kspp", "4"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "invalid spp" in bad.stderr      # Config::validate (host_utils.cuh:48-51)


# ---------------------------------------------------------------------------------------------
# A18: environment-map light (semantics of the reference's CPU renderer, core-light.cpp:394-491,
# core-render.cpp:154-163,290-299,357-369).  No reference-side vectors exist: parity is oracle <-> HIP.
# ---------------------------------------------------------------------------------------------
def _envmap(h=32, seed=7):
    rng = np.random.default_rng(seed)
    w = 2 * h
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.2 + 0.3 * y / h, 0.25 + 0.2 * x / w, 0.4 + 0.0 * x], -1).astype(np.float32)
    for _ in range(6):
        img[rng.integers(0, h), rng.integers(0, w)] += rng.random(3).astype(np.float32) * 40
    return img


def test_envmap_functions_vs_oracle(renderer, O):
    img = _envmap(64)
    q = np.array([0.2, -0.1, 0.3, 0.9], np.float32)
    rng = np.random.default_rng(11)
    n = 8192
    u = rng.random((n, 2)).astype(np.float32)
    wi = rng.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    _load_cornell(renderer, O, 64, 64)
    renderer.upload_envmap(img, q)
    try:
        g = renderer.test_envmap(u, wi)
    finally:
        renderer.clear_envmap()
    a = O.envmap_sample(img, q, u)
    e = O.envmap_eval(img, q, wi)
    # the two binary searches and the table reads are exact; uv and pdf carry v_rcp-based divisions (1-2 ulp), and
    # a uv within an ulp of a texel border may pick the neighbouring texel
    assert np.array_equal(g["ok"], a["ok"])
    assert np.abs(g["uv"] - a["uv"]).max() < 3e-7
    assert np.allclose(g["pdf"], a["pdf"], rtol=2e-6, atol=0)
    assert (g["Le"] == a["Le"]).all(axis=1).mean() > 0.999
    assert np.abs(g["wi"] - a["wi"]).max() < 2e-6                     # sinf/cosf + quaternion with FMA contraction
    # evaluation by direction: the texel can flip for directions within rounding of a texel border
    same = (g["Le_dir"] == e["Le"]).all(axis=1)
    assert same.mean() > 0.999
    assert np.allclose(g["pdf_dir"][same], e["pdf"][same], rtol=2e-6, atol=0)


@pytest.mark.parametrize("accel", [0, 1])
def test_envmap_film_vs_oracle(renderer, pkg, O, accel):
    """Open random-triangle scene (most rays leave it, so both env-map code paths -- seen by a path ray with MIS,
    sampled by NEE -- carry most of the image), brute force and BVH kernels."""
    scene = pkg.host_scene.random_triangle_scene(300, width=48, height=40)
    img = _envmap(32, seed=3)
    osc = O.Scene(scene.xs, scene.ys, scene.zs, scene.mat_id, scene.bsdfs, scene.lights, scene.inf_lights, scene.camera)
    osc.set_envmap(img, (0.1, 0.2, -0.3, 0.9))
    renderer.upload_scene(osc)
    renderer.set_limits(6)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
    om, om2 = O.render(osc, 32, max_depth=6, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3])
    assert om[..., :3].max() > 0.5                      # the map is bright: this is not a black image
    rmse = float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2)))
    rel = rmse / float(om[..., :3].mean())
    assert rel < 2e-3, (rmse, rel)                      # hot texels (x40) make absolute RMSE scale with the map


def test_envmap_clear_restores_constant_environment(renderer, O):
    _load_cornell(renderer, O, 48, 48, 8)
    renderer.render(8); renderer.sync()
    base = renderer.download_film()
    renderer.upload_envmap(_envmap(16))
    renderer.film_clear(); renderer.render(8); renderer.sync()
    withmap = renderer.download_film()
    renderer.clear_envmap()
    renderer.film_clear(); renderer.render(8); renderer.sync()
    again = renderer.download_film()
    assert np.array_equal(base[0], again[0]) and np.array_equal(base[1], again[1])
    assert not np.array_equal(base[0], withmap[0])


def test_cli_time_report_and_gpus_partition(tmp_path):
    """cli/CLIManager.cpp's --time / short options, and --gpus N: N contexts render interleaved tile sets concurrently
    and the gathered image equals the one-context image byte for byte (here both contexts share the box's one GPU)."""
    import os
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "cuda-optix-pathtracing_amd" / "host" / "dmt-megakernel-hip"
    args = ["--width", "72", "--height", "56", "--spp", "16", "--kspp", "8", "--max-depth", "8", "-d", "gpu", "-t"]
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(), two.mkdir()
    r1 = subprocess.run([str(exe), *args, "-o", str(one)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    assert "Timing report:" in r1.stdout and "kernels (HIP events" in r1.stdout and "PNG encode + write" in r1.stdout
    r2 = subprocess.run([str(exe), *args, "--gpus", "3", "-o", str(two)], capture_output=True, text=True, timeout=120,
                        env=dict(os.environ, DMT_CLI_SHARE_DEVICE="1"))
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "(3 GPUs)" in r2.stdout
    for name in ("output-16.png", "output-16_sqrt_mse.png"):
        assert (one / name).read_bytes() == (two / name).read_bytes(), name


# ---------------------------------------------------------------------------------------------
# JSON scene front-end end to end: file -> host loader -> HIP render == oracle render of the same arrays
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name, accel", [("three_boxes.json", 0), ("three_boxes.json", 1), ("ball_envmap.json", 1)])
def test_json_scene_renders_like_the_oracle(renderer, pkg, O, name, accel):
    """three_boxes: primitives, three material kinds, spot + point light, env map.  ball_envmap: BASELINE config 3 in
    miniature (FBX mesh, conductor, env map: the NEE + MIS path), through the BVH kernel."""
    hs = pkg.host_scene.load_json(GOLDEN / "json_scene" / name)
    osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera)
    osc.set_envmap(hs.env_rgb)
    renderer.upload_scene(hs)
    renderer.set_limits(hs.max_depth)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(16)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
    om, om2 = O.render(osc, 16, max_depth=hs.max_depth, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3])
    assert om[..., :3].mean() > 0.02
    rmse = float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2)))
    assert rmse < 1e-3 * max(1.0, float(om[..., :3].mean())), rmse


@pytest.mark.parametrize("accel", [0, 1])
def test_fractional_metallic_blend_vs_oracle(renderer, pkg, O, tmp_path, accel):
    """SURVEY 8f-1 / core-material.cpp:272-286, :383-394: a material that is metallic by a fraction evaluates and samples BOTH
    GGX lobes and blends them (record pair, BS_GGX_BLEND; *_tex kernels).  three_boxes.json with gold at 0.35 and glass at
    0.6: film vs the oracle, and the blend must differ from both pure readings.  Parity unpinned by the reference (its CPU
    renderer cannot be built): oracle <-> HIP on identical arrays."""
    import json, shutil
    src = GOLDEN / "json_scene"
    films = {}
    for tag, gold, glass in (("blend", 0.35, 0.6), ("pure", 1.0, 0.0)):
        d = json.loads((src / "three_boxes.json").read_text())
        d["materials"][1]["metallic"] = gold
        d["materials"][0]["metallic"] = glass
        shutil.copy(src / "sky_32x16.png", tmp_path / "sky_32x16.png")
        (tmp_path / f"{tag}.json").write_text(json.dumps(d))
        hs = pkg.host_scene.load_json(tmp_path / f"{tag}.json")
        assert hs.bsdfs.shape[0] == (5 if tag == "blend" else 3)
        osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera)
        osc.set_envmap(hs.env_rgb)
        renderer.upload_scene(hs)
        renderer.set_limits(hs.max_depth)
        renderer.set_accel(accel)
        renderer.set_partition(0, 1)
        try:
            renderer.film_clear()
            renderer.render(32)
            renderer.sync()
            mean, m2 = renderer.download_film()
        finally:
            renderer.set_accel(0)
            renderer.clear_envmap()
        om, om2 = O.render(osc, 32, max_depth=hs.max_depth, threads=8)[:2]
        assert np.isfinite(mean).all() and np.array_equal(m2[..., 3], om2[..., 3])
        # The reference's blended pdf is (1 - pdfD) pdfC + pdfD metallic (argument order of its float lerp; kept as written),
        # which is negative or tiny wherever a lobe's pdf exceeds 1: single samples of the blend reach +-400 in a scene whose
        # pure reading stays below 20.  An absolute film RMSE is meaningless there; the check is per pixel, RELATIVE to the
        # RMS of the pixel's own samples (from the oracle's Welford M2): |gpu - oracle| <= 1e-3 max(1, rms of the samples).
        rms = np.sqrt(om[..., :3].astype(np.float64) ** 2 + om2[..., :3] / 32.0)
        rel = np.abs(mean[..., :3] - om[..., :3]) / np.maximum(1.0, rms)
        # Where the blended pdf passes through zero a sample is divided by ~0 (pixels with samples of 1e4 and more) and the
        # last bits decide: a permille of the values differs by more (measured: 99 % within 1e-5, 0.14 % beyond 1e-3).
        assert float(np.quantile(rel, 0.99)) < 1e-4, (tag, float(np.quantile(rel, 0.99)))
        assert float((rel > 1e-3).mean()) < 5e-3 and float((rel > 1e-2).mean()) < 1e-3, (tag, float((rel > 1e-3).mean()), float((rel > 1e-2).mean()))
        if tag == "pure":
            assert float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2))) < 1e-3
        films[tag] = mean
    assert film_rmse(films["blend"], films["pure"]) > 1e-2      # the fraction changes the picture


def test_textured_metallic_vs_oracle(renderer, pkg, O, tmp_path):
    """A 1-channel 'metallic' MAP on the reference's scene_test.json teapot (the roughness PNG doubles as the map): the
    fraction is sampled per hit (core-material.cpp:209-216) and drives the same blend; BVH + env map + texture kernel."""
    import json, shutil
    shutil.copytree(GOLDEN / "scene_test", tmp_path / "s")
    j = json.loads((GOLDEN / "scene_test" / "scene_test.json").read_text())
    j["textures"].append({"name": "m", "type": "metallic", "path": "./res/textures/chippedPaint/Paint_Chipped_1K_roughness.png"})
    j["materials"][0]["metallic"] = "m"
    (tmp_path / "s" / "m.json").write_text(json.dumps(j))
    hs = pkg.host_scene.load_json(tmp_path / "s" / "m.json")
    assert hs.bsdfs.shape[0] == 2 and hs.mat_tex.shape == (2, 4)
    osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera)
    osc.set_envmap(hs.env_rgb)
    osc.set_textures(hs.tex_rgba, hs.tex_desc, hs.mat_tex, hs.tri_uv)
    w, h, spp = hs.width, hs.height, 16
    renderer.upload_scene(hs)
    renderer.set_limits(hs.max_depth)
    renderer.set_accel(1)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(spp)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
        renderer.upload_textures(None, None, None, None)
    assert np.isfinite(mean).all() and np.all(m2[..., 3] == spp)
    y0, y1 = h // 2 - 8, h // 2 + 8
    om, om2 = O.render(osc, spp, max_depth=hs.max_depth, region=(0, y0, w, y1), threads=16)[:2]
    assert np.array_equal(m2[y0:y1, :, 3], om2[y0:y1, :, 3])
    rms = np.sqrt(om[y0:y1, :, :3].astype(np.float64) ** 2 + om2[y0:y1, :, :3] / float(spp))   # see the blend test above
    rel = np.abs(mean[y0:y1, :, :3] - om[y0:y1, :, :3]) / np.maximum(1.0, rms)
    # On top of the singular pdf, the per-hit roughness lands in a 16-bit alpha (ggxCommon): a last-bit difference of the bilinear
    # lookup flips the quantised alpha of a sharp lobe -- ~0.4 % of the SAMPLES differ by ~0.3 % (tools/diag_blend_tex.py), so
    # at 16 spp a few per cent of the pixels carry one.  Bulk exact, tails bounded:
    assert float(np.median(rel)) < 1e-6 and float(np.quantile(rel, 0.9)) < 1e-4, (float(np.median(rel)), float(np.quantile(rel, 0.9)))
    assert float((rel > 1e-2).mean()) < 1e-2, float((rel > 1e-2).mean())


def test_oren_nayar_albedo_texture_in_a_reference_format_scene(renderer, pkg, O, tmp_path):
    """Round-2 advisor: the albedo path of apply_material_textures was only covered by a hand-built scene.  Here a JSON scene in
    the reference's schema -- three_boxes.json with the chalk floor's `diffuse` replaced by the reference's chipped-paint
    albedo PNG (a data fixture) on an `oren-nayar-dielectric` material -- goes through the loader, the *_tex kernels and the
    oracle; the texture must change the floor."""
    import json, shutil
    src = GOLDEN / "json_scene"
    shutil.copy(src / "sky_32x16.png", tmp_path / "sky_32x16.png")
    shutil.copy(GOLDEN / "scene_test" / "res" / "textures" / "chippedPaint" / "Paint_Chipped_1K_albedo.png", tmp_path / "albedo.png")
    films = {}
    for tag in ("textured", "plain"):
        d = json.loads((src / "three_boxes.json").read_text())
        if tag == "textured":
            d["textures"] = [{"name": "paint", "type": "diffuse", "path": "./albedo.png"}]
            chalk = [m for m in d["materials"] if "oren-nayar-dielectric" in m][0]
            chalk["diffuse"] = "paint"
        (tmp_path / f"{tag}.json").write_text(json.dumps(d))
        hs = pkg.host_scene.load_json(tmp_path / f"{tag}.json")
        osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera)
        osc.set_envmap(hs.env_rgb)
        if tag == "textured":
            assert hs.tex_desc is not None and hs.mat_tex[:, 0].tolist().count(0) == 1
            osc.set_textures(hs.tex_rgba, hs.tex_desc, hs.mat_tex, hs.tri_uv)
        renderer.upload_scene(hs)
        renderer.set_limits(hs.max_depth)
        renderer.set_accel(1)
        renderer.set_partition(0, 1)
        try:
            renderer.film_clear()
            renderer.render(16)
            renderer.sync()
            mean, m2 = renderer.download_film()
            if tag == "textured":
                with pytest.raises(pkg.DmtError, match="counting kernels"):      # the stats build has no texture variant: refused, not mislabelled
                    renderer.render_stats(1)
        finally:
            renderer.set_accel(0)
            renderer.clear_envmap()
            renderer.upload_textures(None, None, None, None)
        om, om2 = O.render(osc, 16, max_depth=hs.max_depth, threads=8)[:2]
        assert np.array_equal(m2[..., 3], om2[..., 3])
        assert float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2))) < 1e-3 * max(1.0, float(om[..., :3].mean()))
        films[tag] = mean
    assert film_rmse(films["textured"], films["plain"]) > 5e-3


def test_cli_renders_a_json_scene(tmp_path):
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "cuda-optix-pathtracing_amd" / "host" / "dmt-megakernel-hip"
    scene = GOLDEN / "json_scene" / "three_boxes.json"
    r = subprocess.run([str(exe), "--scene", str(scene), "--kspp", "4", "--out", str(tmp_path), "--bvh"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Width:     96" in r.stdout and "SPP:       8" in r.stdout
    assert (tmp_path / "output-8.png").stat().st_size > 500 and (tmp_path / "output-8_sqrt_mse.png").exists()
    r = subprocess.run([str(exe), "--scene", str(tmp_path / "missing.json")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr


def test_sphere_envmap_workload_scene(renderer, pkg, O):
    """The synthetic stand-in for BASELINE config 3 that bench.py offers (sphere + ground under an HDR sky), small."""
    sc = pkg.host_scene.sphere_envmap_scene(48, 48, lat=8, lon=16, env_height=16)
    osc = O.Scene(sc.xs, sc.ys, sc.zs, sc.mat_id, sc.bsdfs, sc.lights, sc.inf_lights, sc.camera)
    osc.set_envmap(sc.env_rgb)
    renderer.upload_scene(sc)
    renderer.set_limits(8)
    renderer.set_accel(1)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(24)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
    om, om2 = O.render(osc, 24, max_depth=8, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3])
    scale = float(om[..., :3].mean())
    assert scale > 0.2
    assert float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2))) < 2e-3 * scale


# ---------------------------------------------------------------------------------------------
# emissive triangles (diffuse area lights, pbrt-v4 semantics).  No reference implementation: parity is
# oracle <-> HIP only.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("accel", [0, 1])
def test_area_lights_film_vs_oracle(renderer, O, accel):
    sc = O.cornell_box(56, 56)
    # no point/spot lights: light the box with two of its own triangles (both facings occur in the soup)
    sc.lights = sc.lights[:0]
    emit = [0, 1, 16, 17, 20, 21]
    # emission is one-sided on normalize(cross(p1 - p0, p2 - p0)) (pbrt's convention), which for the reference's
    # generators points out of the room: rewind four of the emitters so that both facings are exercised
    for a in (sc.xs, sc.ys, sc.zs):
        a[[0, 1, 16, 17]] = a[[0, 1, 16, 17]][:, [0, 2, 1, 3]]
    sc.set_area_lights(emit, [[6, 6, 5], [6, 6, 5], [12, 14, 20], [12, 14, 20], [20, 15, 10], [20, 15, 10]])
    renderer.upload_scene(sc)
    renderer.set_limits(6)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.upload_area_lights([], np.zeros((0, 3), np.float32))
    om, om2 = O.render(sc, 32, max_depth=6, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3])
    scale = float(om[..., :3].mean())
    plain = O.render(O.cornell_box(56, 56), 4, max_depth=6, threads=8)[0]
    assert scale > 0.2 and abs(scale - float(plain[..., :3].mean())) > 0.1   # the emitters light the room
    assert float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2))) < 1e-3 * max(1.0, scale)


def test_area_lights_are_cleared_by_a_new_soup(renderer, O):
    sc = O.cornell_box(32, 32)
    sc.set_area_lights([0], [[5, 5, 5]])
    renderer.upload_scene(sc)
    plain = O.cornell_box(32, 32)
    renderer.upload_scene(plain)                          # new triangles: the emissive list must not survive
    renderer.set_limits(4); renderer.film_clear(); renderer.render(4); renderer.sync()
    a = renderer.download_film()
    om = O.render(plain, 4, max_depth=4, threads=4)[0]
    assert float(np.sqrt(np.mean((a[0][..., :3] - om[..., :3]) ** 2))) < 1e-4


# ---------------------------------------------------------------------------------------------
# PBRT-v4 subset end to end: file -> loader -> HIP render, checked against the oracle (tight) and
# against pbrt's OWN rendering of the same scene (scenes/pbrt-output.png of the reference, kept as a golden image;
# sRGB 8-bit, spectral renderer: supports a coarse claim only)
# ---------------------------------------------------------------------------------------------
def test_pbrt_cornell_box_vs_oracle_and_pbrt_image(renderer, pkg, O):
    from PIL import Image
    hs = pkg.host_scene.load_pbrt(GOLDEN / "pbrt" / "cornell_box.pbrt")
    renderer.upload_scene(hs)
    renderer.set_limits(hs.max_depth)
    renderer.set_accel(1)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(256)
        renderer.sync()
        mean, m2 = renderer.download_film()
        # oracle on a band of rows, same samples
        osc = O.Scene(hs.xs, hs.ys, hs.zs, hs.mat_id, hs.bsdfs, hs.lights, hs.inf_lights, hs.camera)
        osc.set_area_lights(hs.area_tri, hs.area_le)
        renderer.film_clear()
        renderer.render(16, region=(0, 120, 256, 136))
        renderer.sync()
        band = renderer.download_film()[0][120:136, :, :3]
    finally:
        renderer.set_accel(0)
        renderer.upload_area_lights([], np.zeros((0, 3), np.float32))
    om = O.render(osc, 16, max_depth=hs.max_depth, region=(0, 120, 256, 136), threads=8)[0][120:136, :, :3]
    assert float(np.sqrt(np.mean((band - om) ** 2))) < 1e-3 * max(1.0, float(om.mean()))

    lin = np.clip(mean[..., :3], 0, 1)
    srgb = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * np.power(lin, 1 / 2.4) - 0.055)
    img = np.clip(srgb * 255 + 0.5, 0, 255)
    ref = np.asarray(Image.open(GOLDEN / "pbrt" / "pbrt_render_256.png"))[..., :3].astype(np.float64)
    mad = float(np.abs(img - ref).mean())
    mad_mirrored = float(np.abs(img[:, ::-1] - ref).mean())
    assert mad < 16.0 and mad_mirrored > 2.5 * mad, (mad, mad_mirrored)     # same picture, same orientation
    g1, g2 = img.mean(axis=2).ravel(), ref.mean(axis=2).ravel()
    assert float(np.corrcoef(g1, g2)[0, 1]) > 0.9


@pytest.mark.parametrize("accel", [0, 1])
def test_env_map_and_area_lights_together(renderer, pkg, O, accel):
    """Both optional light kinds at once: the env map takes half of the NEE samples, the uploaded lights and the
    emissive triangles share the other half (k_megakernel_env_area / k_megakernel_bvh_env_area)."""
    scene = pkg.host_scene.random_triangle_scene(200, width=40, height=40)
    osc = O.Scene(scene.xs, scene.ys, scene.zs, scene.mat_id, scene.bsdfs, scene.lights, scene.inf_lights, scene.camera)
    osc.set_envmap(_envmap(16, seed=5))
    osc.set_area_lights(np.arange(0, 200, 7), np.tile([[8.0, 6.0, 4.0]], (29, 1)))
    renderer.upload_scene(osc)
    renderer.set_limits(6)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
        renderer.upload_area_lights([], np.zeros((0, 3), np.float32))
    om, om2 = O.render(osc, 32, max_depth=6, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3])
    scale = float(om[..., :3].mean())
    assert scale > 0.2
    assert float(np.sqrt(np.mean((mean[..., :3] - om[..., :3]) ** 2))) < 2e-3 * scale


# ---------------------------------------------------------------------------------------------
# image textures (albedo / roughness / normal map) of JSON materials.  No reference implementation on the
# megakernel path and no reference-side vectors: parity is oracle <-> HIP only (unpinned).
# ---------------------------------------------------------------------------------------------
def _textured_cornell(O, pkg, res, env=False):
    sc = O.cornell_box(res, res)
    rng = np.random.default_rng(5)
    n = sc.tri_count
    # three textures back to back: 16x16 albedo checker with noise, 8x8 roughness, 32x32 normal-map bumps
    def rgba(a):
        out = np.zeros(a.shape[:2] + (4,), np.uint8)
        out[..., :a.shape[2]] = a
        out[..., 3] = 255
        return out.reshape(-1, 4)
    yy, xx = np.mgrid[0:16, 0:16]
    albedo = (np.stack([((xx + yy) % 2) * 150 + 60, ((xx // 2 + yy) % 2) * 120 + 90, (yy * 13) % 200 + 40], -1) + rng.integers(0, 15, (16, 16, 3)))
    rough = np.repeat(rng.integers(10, 250, (8, 8, 1)), 3, axis=2)
    yy, xx = np.mgrid[0:32, 0:32]
    nx, ny = 0.35 * np.sin(xx * 0.7), 0.35 * np.cos(yy * 0.9)
    nz = np.sqrt(np.maximum(0, 1 - nx * nx - ny * ny))
    nmap = np.stack([(nx * 0.5 + 0.5) * 255, (ny * 0.5 + 0.5) * 255, nz * 255], -1)
    tex = [rgba(albedo.astype(np.uint8)), rgba(rough.astype(np.uint8)), rgba(nmap.astype(np.uint8))]
    desc = np.array([[0, 16, 16], [256, 8, 8], [256 + 64, 32, 32]], np.int32)
    none = 0xFFFFFFFF
    mt = np.full((sc.bsdfs.shape[0], 4), none, np.uint32)
    mt[:, 3] = np.float32(1.0).view(np.uint32)
    mt[0] = [0, 1, 2, np.float32(1.0).view(np.uint32)]            # Oren-Nayar: albedo + roughness + normal map
    mt[1] = [none, 1, 2, np.float32(0.6).view(np.uint32)]         # GGX dielectric: roughness (anisotropy 0.6) + normal map
    mt[3] = [0, none, none, np.float32(1.0).view(np.uint32)]      # albedo only
    mt[5] = [none, none, 2, np.float32(1.0).view(np.uint32)]      # normal map only
    uv = rng.uniform(-0.5, 2.5, (n, 6)).astype(np.float32)        # beyond [0, 1]: exercises the mirror wrap
    sc.set_textures(np.concatenate(tex), desc, mt, uv)
    if env:
        sc.set_envmap(pkg.host_scene.synthetic_sky(16))
    return sc


@pytest.mark.parametrize("accel, env", [(0, False), (1, False), (0, True), (1, True)])
def test_textured_materials_vs_oracle(renderer, pkg, O, accel, env):
    sc = _textured_cornell(O, pkg, 64, env)
    renderer.upload_scene(sc)
    renderer.set_limits(6)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
        L = renderer.test_trace_samples(np.arange(64, dtype=np.int32) % 64, (np.arange(64, dtype=np.int32) * 7) % 64, np.arange(64, dtype=np.int32) % 5)
    finally:
        renderer.set_accel(0)
        renderer.clear_envmap()
        renderer.upload_textures(None, None, None, None)
    om, om2 = O.render(sc, 32, max_depth=6, threads=8)[:2]
    plain = O.cornell_box(64, 64)
    if env:
        plain.set_envmap(pkg.host_scene.synthetic_sky(16))
    pm = O.render(plain, 32, max_depth=6, threads=8)[0]
    assert np.array_equal(m2[..., 3], om2[..., 3]) and np.isfinite(mean).all() and np.isfinite(L).all()
    scale = float(om[..., :3].mean())
    assert film_rmse(om, pm) > 5e-3 * scale                      # the textures do change the picture
    assert film_rmse(mean, om) < RMSE_TOL * max(1.0, scale), film_rmse(mean, om)


def test_texture_upload_is_validated(renderer, pkg, O):
    sc = _textured_cornell(O, pkg, 32)
    renderer.upload_scene(sc)
    bad = sc.tex_desc.copy()
    bad[2, 1] = 4096                                             # descriptor runs past the texel array
    with pytest.raises(pkg.DmtError, match="outside the texel array"):
        renderer.upload_textures(sc.tex_rgba, bad, sc.mat_tex, sc.tri_uv)
    mt = sc.mat_tex.copy()
    mt[0, 0] = 7                                                 # texture index that does not exist
    with pytest.raises(pkg.DmtError, match="does not exist"):
        renderer.upload_textures(sc.tex_rgba, sc.tex_desc, mt, sc.tri_uv)
    renderer.upload_textures(sc.tex_rgba, sc.tex_desc, sc.mat_tex, sc.tri_uv[:-1])   # wrong triangle count: caught at render time
    with pytest.raises(pkg.DmtError, match="do not match"):
        renderer.render(1)
    renderer.upload_textures(None, None, None, None)
    renderer.render(1)                                           # cleared: plain kernels again
    renderer.sync()


# ---------------------------------------------------------------------------------------------
# light tree (opt-in; the uniform pick stays the parity mode).  Parity unpinned: oracle <-> HIP only.
# ---------------------------------------------------------------------------------------------
def _many_lights_cornell(O, pkg, res, nlights=48, env=False):
    from test_light_tree import _lights
    sc = O.cornell_box(res, res)
    # radius 2e-4: "effectively delta" lights (radius / distance < 1e-3, light.cu:29,131), the kind the JSON front-end
    # makes.  For those NEE is Le f / pmf and any selection probability gives the same expected image.  For larger
    # lights the reference weights NEE by a power heuristic that is NOT divided by the light pdf (megakernel.cu:233-238,
    # kept for parity): its expectation depends on the selection probabilities, so there the tree changes the picture.
    L = _lights(pkg, nlights, 11, spread=0.9, radius=2e-4)
    pos = L[:, 8:20].copy().view(np.float32).reshape(-1, 3)
    pos[:] = pos * [1.0, 0.25, 0.9] + [0.0, 0.5, 0.6]             # inside the box (y 1..3, z -0.3..1.5), near the ceiling
    L[:, 8:20] = pos.view(np.uint8).reshape(-1, 12)
    sc.lights = L
    if env:
        sc.set_envmap(pkg.host_scene.synthetic_sky(16))
    return sc


@pytest.mark.parametrize("accel, env", [(0, False), (1, False), (1, True)])
def test_light_tree_film_vs_oracle(renderer, pkg, O, accel, env):
    sc = _many_lights_cornell(O, pkg, 64, env=env)
    sc.light_sampling = 1
    renderer.upload_scene(sc)
    renderer.set_limits(5)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    renderer.set_light_sampling(1)
    try:
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
        renderer.set_light_sampling(0)
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        umean, um2 = renderer.download_film()
    finally:
        renderer.set_light_sampling(0)
        renderer.set_accel(0)
        renderer.clear_envmap()
    om, om2 = O.render(sc, 32, max_depth=5, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3]) and np.isfinite(mean).all()
    scale = float(om[..., :3].mean())
    assert film_rmse(mean, om) < RMSE_TOL * max(1.0, scale), film_rmse(mean, om)
    # same expected image as the uniform pick (both unbiased): the 32-spp means agree within the noise of the noisier one ...
    stderr_u = float((np.sqrt(np.maximum(um2[..., :3], 0)) / 32).mean())
    assert abs(float(mean[..., :3].mean()) - float(umean[..., :3].mean())) < 0.05 * scale
    # ... and the tree has clearly less variance with 48 lights of very different flux and distance
    assert float(m2[..., :3].mean()) < 0.6 * float(um2[..., :3].mean()), (float(m2[..., :3].mean()), float(um2[..., :3].mean()), stderr_u)


@pytest.mark.parametrize("accel, env", [(0, False), (1, False), (0, True), (1, True)])
def test_reference_light_tree_film_vs_oracle(renderer, pkg, O, accel, env):
    """DMT_LIGHTS_TREE_REFERENCE (csrc/light_tree_ref.hpp): the reference's tree with its own semantics --
    normal cones, orientation term, SAOH splits, adaptive cuts of up to FOUR lights = up to four shadow rays per bounce --
    in the *_ltree2 kernels against the oracle's own restatement, 64 lights.  All but the last selected light of a bounce
    are shadow-tested in line (brute force: a triangle pass; BVH: a whole any-hit traversal).  Parity unpinned by the
    reference (experimental code without outputs): oracle <-> HIP."""
    sc = _many_lights_cornell(O, pkg, 64, env=env)
    sc.light_sampling = 2
    renderer.upload_scene(sc)
    renderer.set_limits(5)
    renderer.set_accel(accel)
    renderer.set_partition(0, 1)
    try:
        renderer.set_light_sampling(2)
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        mean, m2 = renderer.download_film()
        renderer.set_light_sampling(1)
        renderer.film_clear()
        renderer.render(32)
        renderer.sync()
        smean, sm2 = renderer.download_film()
    finally:
        renderer.set_light_sampling(0)
        renderer.set_accel(0)
        renderer.clear_envmap()
    om, om2 = O.render(sc, 32, max_depth=5, threads=8)[:2]
    assert np.array_equal(m2[..., 3], om2[..., 3]) and np.isfinite(mean).all()
    scale = float(om[..., :3].mean())
    assert scale > 0.01
    assert film_rmse(mean, om) < RMSE_TOL * max(1.0, scale), film_rmse(mean, om)
    # several lights per bounce: less variance than the single-light tree on the same scene
    assert float(m2[..., :3].mean()) < float(sm2[..., :3].mean()), (float(m2[..., :3].mean()), float(sm2[..., :3].mean()))


def test_light_tree_converges_to_the_uniform_image(renderer, pkg, O):
    """Unbiasedness on the GPU: 2 048 spp with the tree vs 2 048 spp with the uniform pick, 48 x 48 pixels."""
    sc = _many_lights_cornell(O, pkg, 48)
    renderer.upload_scene(sc)
    renderer.set_limits(4)
    renderer.set_accel(0)
    renderer.set_partition(0, 1)
    films = []
    try:
        for mode in (0, 1):
            renderer.set_light_sampling(mode)
            renderer.film_clear()
            renderer.render(2048)
            renderer.sync()
            films.append(renderer.download_film())
    finally:
        renderer.set_light_sampling(0)
    (um, um2), (tm, tm2) = films
    se = np.sqrt(np.maximum(um2[..., :3], 0)) / 2048 + np.sqrt(np.maximum(tm2[..., :3], 0)) / 2048   # standard errors of both means
    z = np.abs(um[..., :3] - tm[..., :3]) / np.maximum(se, 1e-9)
    assert np.percentile(z, 99) < 5.0 and z.mean() < 1.5, (float(np.percentile(z, 99)), float(z.mean()))
    assert float(tm2[..., :3].mean()) < 0.6 * float(um2[..., :3].mean())


def test_light_tree_falls_back_where_it_does_not_apply(renderer, pkg, O):
    """One light, a directional light in the list, emissive triangles: the uniform pick runs (bit-identical films)."""
    import ctypes as C
    H = pkg.host_scene.load_host_library()
    f3 = lambda v: np.ascontiguousarray(v, np.float32).ctypes.data_as(C.c_void_p)
    sc = O.cornell_box(40, 40)
    distant = np.zeros(32, np.uint8)
    H.dmt_host_make_directional_light(f3([0.4, 0.4, 0.3]), f3([0.2, 0.6, -0.7]), C.c_float(0.0), distant.ctypes.data_as(C.c_void_p))
    cases = [sc.lights.copy(), np.concatenate([sc.lights, distant[None]])]
    renderer.set_limits(5)
    renderer.set_accel(0)
    renderer.set_partition(0, 1)
    try:
        for lights in cases:
            sc.lights = lights
            renderer.upload_scene(sc)
            films = []
            for mode in (0, 1):
                renderer.set_light_sampling(mode)
                renderer.film_clear()
                renderer.render(8)
                renderer.sync()
                films.append(renderer.download_film())
            assert np.array_equal(films[0][0], films[1][0]) and np.array_equal(films[0][1], films[1][1])
    finally:
        renderer.set_light_sampling(0)


def test_textures_and_area_lights_are_refused_together(renderer, pkg, O):
    sc = _textured_cornell(O, pkg, 32)
    sc.set_area_lights([20], [[5, 5, 5]])
    renderer.upload_scene(sc)
    try:
        with pytest.raises(pkg.DmtError, match="not supported"):
            renderer.render(1)
    finally:
        renderer.upload_textures(None, None, None, None)
        renderer.upload_area_lights([], np.zeros((0, 3), np.float32))

