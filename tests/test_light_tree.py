"""Light tree (csrc/light_tree.hpp): the product's builder / probabilities against the oracle's independent
restatement, and the properties that make the estimator unbiased.  No reference-side vectors exist (the reference's light
tree is experimental CPU code that its build disables): parity unpinned, uniform pick remains the parity mode.  No GPU."""
import ctypes as C

import numpy as np
import pytest


def _lights(pkg, n, seed, spread=3.0, radius=0.02):
    H = pkg.host_scene.load_host_library()
    rng = np.random.default_rng(seed)
    f3 = lambda v: np.ascontiguousarray(v, np.float32).ctypes.data_as(C.c_void_p)
    out = []
    for i in range(n):
        rec = np.zeros(32, np.uint8)
        pos = rng.uniform(-spread, spread, 3) + [0, 6, 0]
        col = rng.uniform(0.05, 6.0, 3)
        if i % 3 == 0:
            H.dmt_host_make_spot_light(f3(col), f3(pos), f3([0, 0, -1]), C.c_float(0.9), C.c_float(0.7), C.c_float(radius / 2), rec.ctypes.data_as(C.c_void_p))
        else:
            H.dmt_host_make_point_light(f3(col), f3(pos), C.c_float(radius), rec.ctypes.data_as(C.c_void_p))
        out.append(rec)
    return np.stack(out)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 301])
def test_probabilities_sum_to_one_and_match_the_oracle(pkg, O, n):
    L = _lights(pkg, n, n)
    rng = np.random.default_rng(100 + n)
    for _ in range(8):
        p = rng.uniform(-4, 4, 3) + [0, 6, 0]
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        a, nodes, depth = pkg.light_tree_pmfs(L, p, nrm)
        b, onodes = O.light_tree_pmfs(L, p, nrm)
        assert nodes == onodes == 2 * n - 1 and depth <= 2 * int(np.ceil(np.log2(max(n, 2)))) + 4
        assert abs(float(a.sum()) - 1.0) < 1e-5 and (a > 0).all()          # every light reachable: unbiased NEE
        assert np.allclose(a, b, rtol=1e-5, atol=1e-9)                     # two independent builders, same tree


def test_importance_follows_flux_and_distance(pkg):
    L = _lights(pkg, 40, 7)
    pos = L[:, 8:20].copy().view(np.float32).reshape(-1, 3)
    p = pos[5] + np.array([0.05, 0.0, 0.0], np.float32)                    # right next to light 5
    a, _, _ = pkg.light_tree_pmfs(L, p, [1, 0, 0])
    assert a.argmax() == 5 and a[5] > 10.0 / 40                            # ten times the uniform probability
    far = np.array([60.0, 6.0, 0.0], np.float32)                           # far away: probability ~ flux
    a, _, _ = pkg.light_tree_pmfs(L, far, [-1, 0, 0])
    half = lambda h: h.view(np.float16).astype(np.float32)
    lum = (half(L[:, 0:6].copy()).reshape(-1, 3) * [0.2126, 0.7152, 0.0722]).sum(1)
    assert np.corrcoef(a, lum / lum.sum())[0, 1] > 0.97


def test_degenerate_layouts(pkg, O):
    L = _lights(pkg, 9, 1)
    same = L.copy()
    same[:, 8:20] = same[0, 8:20]                                           # all lights at one point
    line = L.copy()
    line[:, 12:20] = line[0, 12:20]                                         # collinear along x
    for recs in (same, line):
        a, nodes, _ = pkg.light_tree_pmfs(recs, [0.3, 2.0, 0.1], [0, 1, 0])
        b, _ = O.light_tree_pmfs(recs, [0.3, 2.0, 0.1], [0, 1, 0])
        assert nodes == 17 and abs(float(a.sum()) - 1.0) < 1e-5 and np.allclose(a, b, rtol=1e-5, atol=1e-9)
    a, nodes, depth = pkg.light_tree_pmfs(L[:1], [0, 0, 0], [0, 1, 0])
    assert nodes == 1 and depth == 1 and a.tolist() == [1.0]


# ---- DMT_LIGHTS_TREE_REFERENCE: the reference's tree with its own semantics (csrc/light_tree_ref.hpp) --------------------
@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 301])
def test_reference_tree_selection_matches_the_oracle(pkg, O, n):
    """The product's builder + cut + selection (the function the *_ltree2 kernels call, run on the host) against the oracle's
    own restatement of core-light-tree-builder.cpp: same cut size, same lights, same probabilities; up to
    LightTreeMaxSplitSize = 4 lights per shading point, each with pmf <= start pmf."""
    L = _lights(pkg, n, 1000 + n)
    rng = np.random.default_rng(7 + n)
    k = 256
    p = rng.uniform(-5, 5, (k, 3)) + [0, 6, 0]
    nrm = rng.normal(size=(k, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    u = rng.uniform(0, 1, k).astype(np.float32)
    for start in (1.0, 0.5):
        ai, ap, ac, nodes, depth = pkg.light_tree_ref_select(L, p, nrm, u, start)
        bi, bp, bc, onodes = O.light_tree_ref_select(L, p, nrm, u, start)
        assert nodes == onodes == 2 * n - 1 and depth <= 60
        assert np.array_equal(ac, bc) and np.array_equal(ai, bi)
        assert np.allclose(ap, bp, rtol=2e-5, atol=1e-9)
        assert ac.max() <= 4 and (ap <= start * (1 + 1e-6)).all()
        valid = np.arange(4)[None, :] < ac[:, None]
        assert (ap[valid] > 0).all() and (ai[valid] >= 0).all() and (ai[valid] < n).all() and (ai[~valid] == -1).all()
        for row, c in zip(ai, ac):                      # the walks below different cut nodes end in different lights
            assert len(set(row[:c].tolist())) == c
    if n >= 17:
        assert ac.max() >= 2                            # the adaptive cut really splits somewhere


def test_reference_tree_orientation_term(pkg):
    """lbImportance's cone test (core-light-tree-builder.cpp:124-130): a shading point BEHIND a spot light (outside its
    emission falloff) gives that light importance 0 -- it is never selected there -- while a point light next to it is."""
    import ctypes as C
    H = pkg.host_scene.load_host_library()
    f3 = lambda v: np.ascontiguousarray(v, np.float32).ctypes.data_as(C.c_void_p)
    spot, point = np.zeros(32, np.uint8), np.zeros(32, np.uint8)
    H.dmt_host_make_spot_light(f3([5, 5, 5]), f3([0, 0, 0]), f3([0, 0, -1]), C.c_float(0.95), C.c_float(0.8), C.c_float(0.01), spot.ctypes.data_as(C.c_void_p))
    H.dmt_host_make_point_light(f3([1, 1, 1]), f3([3, 0, 0]), C.c_float(0.01), point.ctypes.data_as(C.c_void_p))
    L = np.stack([spot, point])
    u = np.linspace(0.01, 0.99, 64).astype(np.float32)
    below = np.tile([0.0, 0.0, -4.0], (64, 1))           # in the cone's axis
    above = np.tile([0.0, 0.0, 4.0], (64, 1))            # behind the spot light
    up = np.tile([0.0, 0.0, 1.0], (64, 1))
    bi, _, bc, _, _ = pkg.light_tree_ref_select(L, below, up, u)
    ai, _, ac, _, _ = pkg.light_tree_ref_select(L, above, -up, u)
    assert (bi[np.arange(4)[None, :] < bc[:, None]] == 0).any()
    assert not (ai[np.arange(4)[None, :] < ac[:, None]] == 0).any() and (ai[:, 0] == 1).all()


def test_reference_tree_degenerate_layouts(pkg, O):
    L = _lights(pkg, 9, 1)
    same = L.copy()
    same[:, 8:20] = same[0, 8:20]                                           # all lights at one point: no plane separates them
    rng = np.random.default_rng(3)
    p = rng.uniform(-3, 3, (32, 3)) + [0, 6, 0]
    nrm = np.tile([0.0, -1.0, 0.0], (32, 1))
    u = rng.uniform(0, 1, 32).astype(np.float32)
    ai, ap, ac, nodes, depth = pkg.light_tree_ref_select(same, p, nrm, u)
    bi, bp, bc, onodes = O.light_tree_ref_select(same, p, nrm, u)
    assert nodes == onodes == 17 and np.array_equal(ai, bi) and np.array_equal(ac, bc) and np.allclose(ap, bp, rtol=2e-5, atol=1e-9)
    assert np.isfinite(ap).all()
