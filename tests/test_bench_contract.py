"""CPU checks of bench.py's static contract: the default workload is BASELINE's configuration, the algorithmic-bytes
model is SURVEY 8(d)'s, the roofline record is a fraction <= 1 for every workload on every code path (with and
without CPU-leg counters, with and without a PMC record), and the JSON line carries the keys the driver reads
(checked on the source: running it needs a GPU)."""
import importlib.util
import itertools
import json
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_default_workload_is_baseline_config_2():
    b = _bench()
    assert b.DEFAULT_WORKLOAD == "cornell_1024x1024_1024spp_8bounces"
    assert b.WORKLOADS[b.DEFAULT_WORKLOAD][:4] == (1024, 1024, 1024, 8)
    assert b.WORKLOADS[b.C1_WORKLOAD][:4] == (512, 512, 64, 4)          # BASELINE configs[0]: the CPU-baseline frame
    for name in ("cornell_512x512_64spp_4bounces", "random1M_1024x1024_512spp_8bounces",
                 "sphere_fbx_veranda_256x256_2048spp_12bounces", "cornell_4096x4096_4096spp_8bounces"):
        assert name in b.WORKLOADS                        # BASELINE configs 1, 4, 3 (named assets), 5
    # config 3's assets travel with the repo (data fixtures)
    for f in ("sphere.fbx", "veranda_polyhaven_1k.png", "c3_sphere_veranda.json"):
        assert (ROOT / "tests" / "golden" / "c3" / f).exists()


def test_algorithmic_bytes_model():
    b = _bench()
    stats = {"samples": 10, "tri_tests": 1000, "bounces": 30, "hits": 30}
    # B_film + sum N_tris * 48 + N_bounces * 64 + N_hits * 4, per sample; film = 64 B per pixel per launch
    assert abs(b.algorithmic_bytes_per_sample(stats, 16) - (64 / 16 + (1000 * 48 + 30 * 64 + 30 * 4) / 10)) < 1e-9
    stats["node_visits"] = 500
    assert abs(b.algorithmic_bytes_per_sample(stats, 16, 128.0, 64.0) -
               (64 / 16 + (1000 * 64 + 500 * 128 + 30 * 64 + 30 * 4) / 10)) < 1e-9
    assert b.HBM_PEAK_GBS == 8000.0 and b.FP32_VALU_PEAK_TFLOPS == 157.3
    assert abs(b.flops_per_sample(stats) - (1000 * 60 + 500 * 60 + 30 * 620) / 10) < 1e-9


BVH_STATS = {"samples": 1e6, "tri_tests": 55e6, "bounces": 2.85e6, "hits": 2.85e6, "node_visits": 146e6,
             "closest_rays": 3.5e6, "shadow_rays": 2.8e6}
PMC = {"kspp": 0, "hbm_bytes_per_launch": 36.5e9, "SQ_ACTIVE_INST_VALU": 2.08e11, "kernel_ms": 324.0, "clock_ghz": 2.3,
       "lane_utilisation": 0.7, "wait_any_share": 0.2, "valu_insts_per_sample": 187.0}


@pytest.mark.parametrize("name", sorted(_bench().WORKLOADS))
def test_roofline_record_on_every_path(name):
    """every workload through the fallback branch (no CPU leg: N > 1 or --no-cpu-baseline) and through the
    counted branch, with and without a PMC record; `frac` is a fraction of the bound that limits the kernel, never > 1."""
    b = _bench()
    w, h, spp, depth, kind = b.WORKLOADS[name]
    bvh = kind != "cornell"
    for world, have_stats, have_pmc in itertools.product((1, 8), (False, True), (False, True)):
        stats = (BVH_STATS if bvh else b.WORKLOAD_STATS.get(name)) if have_stats or not bvh else None
        if not bvh and not have_stats:
            stats = b.WORKLOAD_STATS.get(name)            # what main() falls back to
        pmc = dict(PMC, kspp=spp) if have_pmc else None
        # a fast kernel: 8 Gsamples/s (below the VALU roof) puts the SURVEY 8(d) cache-level rate far above the HBM peak
        samples = float(w) * h * spp / world
        ms = samples / 8e9 * 1e3
        if pmc:
            pmc["kernel_ms"] = ms
            pmc["hbm_bytes_per_launch"] = 0.05 * b.HBM_PEAK_GBS * 1e9 * ms * 1e-3
            pmc["SQ_ACTIVE_INST_VALU"] = 0.9 * b.SIMDS * ms * 1e-3 * 2.3e9 / 4
        r = b.build_roofline(name, stats, ms, samples, spp, pmc, {"vgprs": 128, "lds_bytes": 1, "blocks_per_cu": 4, "cu_count": 256})
        json.dumps(r)
        assert r["bound"] == ("valu_issue" if bvh else "valu")   # the bound the counters show
        assert r["frac"] is None or 0 <= r["frac"] <= 1.0
        assert r["pmc_record"]["matches_loaded_library"] is False      # the synthetic record carries no library hash
        if bvh:
            assert (r["frac"] is not None) == have_pmc and r["peak"] == 1.0
            if have_pmc:
                assert abs(r["frac"] - 0.9) < 1e-3 and r["traffic"] == pmc["hbm_bytes_per_launch"]
                assert abs(r["hbm_view"]["frac_of_spec_peak"] - 0.05) < 1e-3
        else:
            assert r["unit"] == "TFLOP/s" and (r["frac"] is not None) == (stats is not None)
            if stats is not None:
                assert r["cache_level_rate"]["GB/s"] > b.HBM_PEAK_GBS       # the rate round 1 mislabelled as an HBM fraction
                assert 0 < r["valu_view"]["frac_tests_only"] < r["valu_view"]["frac"]   # both FLOP models are reported
        if have_pmc:
            assert abs(r["issue_view"]["valu_busy"] - 0.9) < 1e-3
    # the Cornell workloads all have fallback counters (round 1: KeyError for the 4096^2 frame)
    if not bvh:
        assert name in b.WORKLOAD_STATS


def test_json_line_keys_present_in_source():
    src = (ROOT / "bench.py").read_text()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(rf'"{key}"\s*:', src), key
    for key in ("bound", "achieved", "peak", "frac", "traffic", "cores", "kind", "sample", "cpu"):
        assert re.search(rf'"{key}"\s*:', src), key
    for key in ("secondary", "per_rank", "fold_handover", "kernel_ms_per_step", "combine_ms_per_step", "final_barrier_wait_ms"):
        assert re.search(rf'"{key}"\s*:', src), key
    assert "r.sync()" in src and "sys.exit(main())" in src      # a launch that lost a sample chunk / a bad film fails the run


def test_issue_view_uses_the_cycles_of_its_own_pass():
    """SQ_BUSY_CYCLES (shader cycles x 32 shader engines) of the same PMC pass is the denominator; a saturated pipe reads
    slightly above 1 raw and clips to 1 (bench.py explains why)."""
    b = _bench()
    name = b.DEFAULT_WORKLOAD
    pmc = {"kspp": 1024, "hbm_bytes_per_launch": 36.66e9, "SQ_ACTIVE_INST_VALU": 198412231802.0, "SQ_INSTS_VALU": 191378109987.0,
           "SQ_BUSY_CYCLES": 23838067799.0, "kernel_ms": 314.5189, "clock_ghz": 2.0}
    r = b.build_roofline(name, b.WORKLOAD_STATS[name], 312.3, 1024.0 ** 3, 1024, pmc)
    iv = r["issue_view"]
    assert abs(iv["raw"] - 1.0405) < 2e-3 and iv["valu_busy"] == 1.0 and abs(iv["clock_ghz"] - 2.3685) < 2e-3
    assert abs(iv["valu_insts_x4_per_simd_cycle"] - 1.0036) < 2e-3
    assert abs(r["valu_view"]["frac"] - 0.27) < 0.01 and abs(r["valu_view"]["frac_tests_only"] - 0.228) < 0.01
