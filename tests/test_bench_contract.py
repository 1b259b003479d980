"""CPU checks of bench.py's static contract: the default workload is BASELINE's configuration, the algorithmic-bytes
model is SURVEY 8(d)'s, and the JSON line carries the keys the driver reads (checked on the source: running it needs a GPU)."""
import importlib.util
import re
from pathlib import Path

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
    for name in ("cornell_512x512_64spp_4bounces", "random1M_1024x1024_512spp_8bounces",
                 "sphere_envmap_1024x1024_2048spp_8bounces", "cornell_4096x4096_4096spp_8bounces"):
        assert name in b.WORKLOADS                        # BASELINE configs 1, 4, 3 (synthetic assets), 5


def test_algorithmic_bytes_model():
    b = _bench()
    stats = {"samples": 10, "tri_tests": 1000, "bounces": 30, "hits": 30}
    # B_film + sum N_tris * 48 + N_bounces * 64 + N_hits * 4, per sample; film = 64 B per pixel per launch
    assert abs(b.algorithmic_bytes_per_sample(stats, 16) - (64 / 16 + (1000 * 48 + 30 * 64 + 30 * 4) / 10)) < 1e-9
    assert b.BVH_NODE_BYTES == 128 and b.HBM_PEAK_GBS == 8000.0


def test_json_line_keys_present_in_source():
    src = (ROOT / "bench.py").read_text()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(rf'"{key}"\s*:', src), key
    for key in ("bound", "achieved", "peak", "frac", "traffic", "cores", "kind", "sample"):
        assert re.search(rf'"{key}"\s*:', src), key
