#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

Contract (see the task): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
by `python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL).  Rank 0 prints
ONE JSON line.

Workload (BASELINE.json configs[1]): the reference's cornellBox() scene, 1024x1024, 1024 spp,
bounce cap 8.  One "step" = one full pass of the hot path over that frame: every sample of every
pixel traced and folded into the film (and, for N > 1, the film gathered on rank 0 by an RCCL
reduce over zero-initialised frames -- the tile sets are disjoint, so the sum is an exact gather).
The frame is partitioned by interleaved 8x8 tiles (tile j -> rank j mod N), total work is fixed:
"scaling": "strong".  Scene and film are resident in HBM before the timed region.

PyTorch is plumbing here (device buffers for the film, streams, torch.distributed); every sample is
traced by csrc/libdmt_hip.so through the C ABI.  The CPU oracle is used ONLY in the cpu_baseline
leg (rank 0, N == 1), where it is the thing timed on the host cores and, as a by-product, supplies
the per-sample work counters for the flop / byte models and a band to check the GPU film against.

Roofline record (`roofline`): the bound that limits each workload's kernel, as a fraction <= 1.
  * Cornell workloads (26 triangles, k_megakernel): the scene lives in SGPRs / the scalar cache and the only compulsory
    HBM traffic is the film, so the kernel is VALU bound (the reference's authors found the same on their GPU:
    98 FLOP/B, docs/dmt-mk_roofline_point.txt).  bound = "valu": achieved = useful fp32 FLOP/s from the FLOP model
    below over the live HIP-event kernel time, peak = 157.3 TFLOP/s (MI355X_MICROARCH.md).  Next to it: `issue_view`
    (VALU pipe busy from the committed PMC pass) and `hbm_view` (PMC FETCH/WRITE bytes over the live time).
  * BVH workloads (k_megakernel_bvh*): bound = "valu_issue".  The counters say what limits the per-lane traversal on this
    chip: the VALU pipes are busy ~80 % of the time at half the lanes, while the fabric moves 8-10 % of the HBM peak on a
    59 MB and on a 0.96 GB BVH alike.  frac = issue_view.valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x
    cycles of the same PMC pass), peak 1.0; `hbm_view` (PMC FETCH_SIZE x 2 + WRITE_SIZE over the live kernel time, against
    8 TB/s spec and the guide's 6.29 TB/s copy rate) stands beside it.  No PMC record -> frac null.
  * The PMC records are COMMITTED measurements (profiles/pmc_summary.json, one per workload, with the sha256 of the library
    they were taken on); `pmc_record.matches_loaded_library` says whether the library that ran now is the same build.
  * `cache_level_rate` keeps SURVEY 8(d)'s algorithmic-bytes figure (bytes the algorithm asks for per sample x samples /
    time).  It is served by the scalar cache / L2 / Infinity Cache, NOT by HBM, and may exceed the HBM peak; it is
    labelled as such and never used as `frac`.
FLOP model, stated both ways: `frac` counts 60 flop per triangle test + 60 per BVH node visit (4 slab tests)
+ 620 per bounce, where the 620 is a FILLER backed out of the authors' own nvprof total (17.9 kFLOP per sample on the same
scene at 250.7 tests and 4.65 bounces per sample, docs/dmt-mk_roofline_point.txt:2-6), i.e. shading + sampler flops this
build did not count itself; `frac_tests_only` counts the triangle tests and node visits alone.
A `secondary` array (N = 1, default workload only) carries BASELINE config 4 (1 M random triangles, BVH kernel) timed
for a few steps after the headline's timed region, so that the driver's record holds a number for it as well.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# multi-process GPU work on this pool needs dmabuf IPC (RCCL and CUDA-tensor sharing fail with the legacy mode); the driver's
# environment exports it already -- this only covers a shell that does not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HBM_MEASURED_GBS = 6290.0      # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy); this box: copy 4.7, triad 5.9, fill 6.8
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector
SIMDS = 1024                   # 256 CUs x 4
FLOP_PER_TRI_TEST = 60.0
FLOP_PER_NODE_VISIT = 60.0
FLOP_PER_BOUNCE = 620.0

WORKLOADS = {
    # name: (width, height, spp, max_depth, scene)
    "cornell_1024x1024_1024spp_8bounces": (1024, 1024, 1024, 8, "cornell"),
    "cornell_256x256_2048spp_32bounces": (256, 256, 2048, 32, "cornell"),   # the reference's own published run
    "cornell_512x512_64spp_4bounces": (512, 512, 64, 4, "cornell"),         # BASELINE configs[0]
    # BASELINE configs[3]: 1 M random triangles (SURVEY 8d generator); BVH traversal.  102 MB of nodes + leaves:
    # resident in the 256 MiB Infinity Cache
    "random1M_1024x1024_512spp_8bounces": (1024, 1024, 512, 8, "random1M"),
    # the same generator at 16 M triangles and the SAME density (the cube of centroids 16^(1/3) times wider): nodes +
    # leaves (0.96 GB) are four times the 256 MiB Infinity Cache -> the HBM-footprint point
    "random16M_1024x1024_64spp_8bounces": (1024, 1024, 64, 8, "random16M"),
    # BASELINE configs[2] on its named assets: the reference's scenes/sphere.fbx under scenes/veranda_polyhaven_1k.png,
    # scene description scenes/fbx_example.json (committed as data under tests/golden/c3/), film 256x256 as the JSON says
    "sphere_fbx_veranda_256x256_2048spp_12bounces": (256, 256, 2048, 12, "c3_assets"),
    # BASELINE configs[2] with synthetic assets at 1024^2 (tessellated sphere + ground under a synthetic HDR sky)
    "sphere_envmap_1024x1024_2048spp_8bounces": (1024, 1024, 2048, 8, "sphere_env"),
    # BASELINE configs[4]: the frame the reference quotes for 8 GPUs; runs on any N (strong scaling)
    "cornell_4096x4096_4096spp_8bounces": (4096, 4096, 4096, 8, "cornell"),
}
RANDOM_SCENE_TRIANGLES = {"random1M": (1_000_000, 1.0), "random16M": (16_000_000, 16.0 ** (1.0 / 3.0))}
BVH_NODE_BYTES = 64            # csrc/bvh.hpp Bvh4Node: one 64-byte sector (48 bytes read)
BVH_LEAF_BYTES_PER_TRI = 40    # csrc/bvh.hpp TriPair: 80 B per pair
DEFAULT_WORKLOAD = "cornell_1024x1024_1024spp_8bounces"
C1_WORKLOAD = "cornell_512x512_64spp_4bounces"

# Per-sample work counters of the Cornell workloads, counted by the CPU restatement on rows spread evenly over
# the frame (same procedure as the cpu_baseline leg below, which recounts them live when it runs).
# Used for the flop / byte models when the cpu_baseline leg is skipped (N > 1, --no-cpu-baseline).
WORKLOAD_STATS = {
    "cornell_1024x1024_1024spp_8bounces": {"samples": 1.0, "closest_rays": 3.6101, "shadow_rays": 3.1595,
                                           "tri_tests": 173.8335, "bounces": 3.1595, "hits": 3.1595},
    "cornell_256x256_2048spp_32bounces": {"samples": 1.0, "closest_rays": 5.1006, "shadow_rays": 4.6513,
                                          "tri_tests": 250.6595, "bounces": 4.6513, "hits": 4.6513},
    "cornell_512x512_64spp_4bounces": {"samples": 1.0, "closest_rays": 3.1490, "shadow_rays": 2.6932,
                                       "tri_tests": 150.0929, "bounces": 2.6932, "hits": 2.6932},
    # same scene, camera and cap as the 1024^2 workload: per-sample counters agree to 3 digits (4 rows x 64 spp counted)
    "cornell_4096x4096_4096spp_8bounces": {"samples": 1.0, "closest_rays": 3.6101, "shadow_rays": 3.1595,
                                           "tri_tests": 173.8335, "bounces": 3.1595, "hits": 3.1595},
}


def algorithmic_bytes_per_sample(stats, spp_per_launch, node_bytes=0.0, tri_bytes=48.0):
    """B_sample = B_film + sum_rays[N_nodes * S_node + N_tris * 48] + N_bounces * (32 + 32) + N_hits * 4,
    counted on the same rays (brute force: no BVH nodes; BVH: S_node and the leaf bytes per triangle of csrc/bvh.hpp)."""
    n = float(stats["samples"])
    b_film = 64.0 / spp_per_launch
    return b_film + (stats["tri_tests"] * tri_bytes + stats.get("node_visits", 0.0) * node_bytes +
                     stats["bounces"] * 64.0 + stats["hits"] * 4.0) / n


def flops_per_sample(stats):
    n = float(stats["samples"])
    return (stats["tri_tests"] * FLOP_PER_TRI_TEST + stats.get("node_visits", 0.0) * FLOP_PER_NODE_VISIT +
            stats["bounces"] * FLOP_PER_BOUNCE) / n


def kernel_name(scene_kind):
    if scene_kind == "cornell":
        return "k_megakernel"
    return "k_megakernel_bvh_env" if scene_kind in ("sphere_env", "c3_assets") else "k_megakernel_bvh"


def library_sha16():
    """sha256 (first 16 hex digits) of the HIP library that is loaded (DMT_HIP_LIB or the in-tree build)."""
    import hashlib
    so = os.environ.get("DMT_HIP_LIB") or str(ROOT / "cuda-optix-pathtracing_amd" / "csrc" / "libdmt_hip.so")
    try:
        return hashlib.sha256(Path(so).read_bytes()).hexdigest()[:16]
    except OSError:
        return None


def load_pmc(workload, kspp):
    """profiles/pmc_summary.json record of this workload (written by tools/pmc_summarize.py from rocprofv3 --pmc passes)."""
    pmc = ROOT / "profiles" / "pmc_summary.json"
    try:
        j = json.loads(pmc.read_text())["workloads"].get(workload)
        return j if j and j.get("kspp") == kspp else None
    except Exception:
        return None


def build_roofline(workload, stats, avg_ms, samples_per_launch, kspp, pmc, info=None, world=1):
    """The `roofline` object of the JSON line (pure function: exercised on CPU by tests/test_bench_contract.py).
    stats: per-workload work counters or None; pmc: load_pmc() record or None."""
    scene_kind = WORKLOADS[workload][4]
    use_bvh = scene_kind != "cornell"
    secs = avg_ms * 1e-3
    # the PMC record is a 1-GPU launch over the whole frame; a rank of an N-GPU run renders (and moves) 1/N of it
    traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
    if traffic is not None and world > 1:
        traffic = traffic / world
    hbm_view = None
    if traffic is not None and secs > 0:
        gbs = traffic / secs / 1e9
        hbm_view = {"traffic_bytes_per_launch": traffic, "GB/s": round(gbs, 2),
                    "frac_of_spec_peak": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_peak": round(gbs / HBM_MEASURED_GBS, 4),
                    "source": "profiles/pmc_summary.json: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, separate --pmc passes; live kernel time"}
    valu_view = issue_view = cache_rate = None
    if stats is not None and secs > 0:
        tf = flops_per_sample(stats) * samples_per_launch / secs / 1e12
        tests_only = dict(stats, bounces=0.0)
        tf0 = flops_per_sample(tests_only) * samples_per_launch / secs / 1e12
        valu_view = {"achieved_tflops": round(tf, 3), "peak_tflops": FP32_VALU_PEAK_TFLOPS, "frac": round(tf / FP32_VALU_PEAK_TFLOPS, 4),
                     "flops_per_sample": round(flops_per_sample(stats), 1),
                     "flops_model": "60 x triangle tests + 60 x BVH node visits + 620 x bounces; the 620 per bounce is a filler backed out "
                                    "of the authors' nvprof total, not counted by this build (see module docstring)",
                     "achieved_tflops_tests_only": round(tf0, 3), "frac_tests_only": round(tf0 / FP32_VALU_PEAK_TFLOPS, 4),
                     "flops_per_sample_tests_only": round(flops_per_sample(tests_only), 1)}
        b_sample = algorithmic_bytes_per_sample(stats, kspp, BVH_NODE_BYTES if use_bvh else 0.0,
                                                BVH_LEAF_BYTES_PER_TRI if use_bvh else 48.0)
        cache_rate = {"algorithmic_bytes_per_sample": round(b_sample, 1),
                      "GB/s": round(b_sample * samples_per_launch / secs / 1e9, 1),
                      "note": "SURVEY 8(d) algorithmic bytes over kernel time: served by the scalar cache / L2 / Infinity Cache, "
                              "not an HBM rate (can exceed the HBM peak); HBM traffic is hbm_view"}
    if pmc and pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("kernel_ms"):
        # cycles of the SAME pass: SQ_BUSY_CYCLES counts shader cycles per shader engine (8 XCDs x 4 = 32 of them); the clock of a
        # separate GRBM pass is the fallback for records without it
        if pmc.get("SQ_BUSY_CYCLES"):
            cycles = pmc["SQ_BUSY_CYCLES"] / 32.0
            clk = cycles / (pmc["kernel_ms"] * 1e-3) / 1e9
        else:
            clk = pmc.get("clock_ghz") or 2.4
            cycles = pmc["kernel_ms"] * 1e-3 * clk * 1e9
        busy = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cycles)
        insts = pmc["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cycles) if pmc.get("SQ_INSTS_VALU") else None
        issue_view = {"valu_busy": round(min(busy, 1.0), 4), "raw": round(busy, 4), "clock_ghz": round(clk, 4),
                      "valu_insts_x4_per_simd_cycle": round(insts, 4) if insts is not None else None,
                      "formula": "SQ_ACTIVE_INST_VALU x 4 (the counter is in quad-cycles) / (1024 SIMDs x SQ_BUSY_CYCLES / 32) of the same PMC pass",
                      "raw_above_one": "the counter adds up, per wave, the quad-cycles during which that wave has a VALU instruction in flight; on a "
                                       "saturated SIMD the issue of one wave's instruction overlaps the last stage of another wave's, and a "
                                       "simple wave64 instruction holds the pipe for ~2.5 cycles, not 4 (tools/ubench), so neither this sum nor "
                                       "the instruction count x 4 is an exact cycle count: the Cornell kernel reads 1.10 and 1.07 of the SIMD "
                                       "cycles (1.04 and 1.00 while its polynomial code was packed, 4.5-cycle instructions).  valu_busy clips "
                                       "at 1; below saturation raw is an upper bound",
                      "lane_utilisation": pmc.get("lane_utilisation"), "wait_any_share": pmc.get("wait_any_share"),
                      "valu_insts_per_sample": pmc.get("valu_insts_per_sample")}
    out = {"kernel": kernel_name(scene_kind), "avg_launch_ms": round(avg_ms, 4), "traffic": traffic,
           "hbm_view": hbm_view, "valu_view": valu_view, "issue_view": issue_view, "cache_level_rate": cache_rate}
    if not use_bvh:
        out.update({"bound": "valu", "unit": "TFLOP/s", "peak": FP32_VALU_PEAK_TFLOPS,
                    "achieved": valu_view["achieved_tflops"] if valu_view else None,
                    "frac": valu_view["frac"] if valu_view else None,
                    "note": "26-triangle scene is SGPR / scalar-cache resident, compulsory HBM traffic is the film only: VALU bound "
                            "(SURVEY 8d).  frac = useful fp32 FLOP/s / 157.3 TF; the VALU pipe itself is ~saturated (issue_view): "
                            "the gap is lane utilisation, integer/address/compare work and non-FMA instructions"})
    else:
        out.update({"bound": "valu_issue", "unit": "fraction of VALU issue cycles", "peak": 1.0,
                    "achieved": issue_view["valu_busy"] if issue_view else None,
                    "frac": issue_view["valu_busy"] if issue_view else None,
                    "note": "BVH traversal, one ray per lane (csrc/bvh_device.hpp): the counters show the VALU pipes ~80 % busy at half the "
                            "lanes and the fabric at 8-10 % of the HBM peak (hbm_view) on a BVH inside and one four times beyond the Infinity "
                            "Cache alike, so the bound named here is VALU issue, not HBM (DESIGN.md 4.2.2); frac = issue_view.valu_busy from "
                            "the committed PMC pass"})
    if out["frac"] is not None and out["frac"] > 1.0:   # a model that overshoots its peak is reported as such, not as a fraction
        out["note"] += f"; MODEL ERROR: computed fraction {out['frac']} > 1, withheld"
        out["frac"] = None
    lib = library_sha16()
    out["pmc_record"] = {"tag": None, "lib_sha16": None, "loaded_lib_sha16": lib, "matches_loaded_library": False,
                         "note": "no committed PMC record for this workload / kspp"}
    if pmc:
        out["pmc_record"] = {"tag": pmc.get("tag"), "lib_sha16": pmc.get("lib_sha16"), "loaded_lib_sha16": lib,
                             "matches_loaded_library": bool(pmc.get("lib_sha16")) and pmc.get("lib_sha16") == lib,
                             "note": "hbm_view / issue_view combine the live kernel time with counters of a committed PMC run of this workload; "
                                     "if the library has changed since, they describe the earlier build"}
    if stats is not None:
        out["per_sample"] = {k: round(v / stats["samples"], 3) for k, v in stats.items() if k != "samples"}
    if info:
        out.update({"vgprs": info["vgprs"], "lds_bytes_per_block": info["lds_bytes"], "blocks_per_cu": info["blocks_per_cu"],
                    "cu_count": info["cu_count"]})
    return out


def cpu_model():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def build_scene(pkg, scene_kind, width, height):
    hs = pkg.host_scene
    if scene_kind == "cornell":
        return hs.cornell_box(width, height)
    if scene_kind == "sphere_env":
        return hs.sphere_envmap_scene(width, height)
    if scene_kind == "c3_assets":
        s = hs.load_json(ROOT / "tests" / "golden" / "c3" / "c3_sphere_veranda.json")
        assert (s.width, s.height) == (width, height)
        return s
    count, extent = RANDOM_SCENE_TRIANGLES[scene_kind]
    return hs.random_triangle_scene(count, width=width, height=height, extent=extent)


def oracle_band(O, width, height, spp, max_depth, want_rows, threads, film_mean):
    """Rows of the Cornell frame, spread evenly, rendered by the CPU oracle (the checker, never the thing measured here):
    returns the parity record of the GPU film on those rows and the oracle's work counters."""
    import numpy as np
    oscene = O.cornell_box(width, height)
    nrows = max(1, min(want_rows, height, int(2e7 // (width * spp)) or 1))
    rows = sorted({int((i + 0.5) * height / nrows) for i in range(nrows)})
    stats = {}
    omean = np.zeros((height, width, 4), np.float32)
    om2 = np.zeros((height, width, 4), np.float32)
    for y in rows:
        _, _, st = O.render(oscene, spp, max_depth=max_depth, region=(0, y, width, y + 1), threads=threads,
                            film=(omean, om2), want_stats=True)
        for k, v in st.items():
            stats[k] = stats.get(k, 0) + v
    d = film_mean[rows, :, :3].astype(np.float64) - omean[rows, :, :3]
    parity = {"rmse_vs_cpu_rows": float(np.sqrt((d ** 2).mean(axis=2)).mean()), "tolerance": 1e-3,
              "rows": len(rows), "samples": int(stats["samples"])}
    return parity, stats


SECONDARY_WORKLOAD = "random1M_1024x1024_512spp_8bounces"


def run_secondary(pkg, dev_index, stream_ptr, steps=2, warmup=1):
    """BASELINE config 4 after the headline's timed region (N = 1): scene + BVH build + upload untimed, then `steps`
    full passes (one launch each) between stream synchronisations.  Returns the `secondary` entry."""
    import numpy as np
    name = SECONDARY_WORKLOAD
    width, height, spp, max_depth, scene_kind = WORKLOADS[name]
    t0 = time.perf_counter()
    scene = build_scene(pkg, scene_kind, width, height)
    with pkg.Renderer(dev_index) as r:
        r.set_stream(stream_ptr)
        r.upload_scene(scene)
        r.set_limits(max_depth)
        r.set_accel(1)                       # builds the BVH on the host and uploads it
        t_setup = time.perf_counter() - t0
        for _ in range(warmup):
            r.film_clear(); r.render(spp)
        r.sync()
        r.kernel_time(reset=True)
        r.sched_diag(reset=True)
        t1 = time.perf_counter()
        for _ in range(steps):
            r.film_clear(); r.render(spp)
        r.sync()
        elapsed = time.perf_counter() - t1
        kernel_ms, launches = r.kernel_time(reset=True)
        sched = r.sched_diag()
        _, m2 = r.download_film()
        counts_ok = bool((m2[..., 3] == spp).all())
        info = r.kernel_info()
        st = r.render_stats(1, sample_offset=spp)
        stats = {"samples": st["samples"], "tri_tests": st["tri_tests"], "bounces": st["bounces"], "hits": st["bounces"],
                 "node_visits": st["node_visits"], "closest_rays": st["closest_rays"], "shadow_rays": st["shadow_rays"]}
    roofline = build_roofline(name, stats, kernel_ms / launches, float(width) * height * spp, spp, load_pmc(name, spp), info, 1)
    return {"workload": name, "metric": "Msamples/s (paths x spp / s)",
            "value": round(float(width) * height * spp * steps / elapsed / 1e6, 3), "unit": "Msamples/s",
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "kernel_ms_per_launch": round(kernel_ms / launches, 4), "film_ok": counts_ok,
            "setup_s_untimed": round(t_setup, 2), "triangles": int(scene.tri_count), "accel": "bvh4",
            "roofline": roofline, "fold_handover": sched}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--kspp", type=int, default=0, help="samples per pixel per kernel launch (0 = all spp in one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BASELINE config 4 record appended to the default run")
    ap.add_argument("--cornell-bvh", action="store_true",
                    help="experiment (VERDICT r2 item 9): run a Cornell workload through the BVH kernel instead of the brute-force loop "
                         "(films are bit-identical); the roofline record then still describes the brute-force model")
    ap.add_argument("--cpu-band-rows", type=int, default=16)
    ap.add_argument("--parity-rows", type=int, default=0,
                    help="Cornell workloads without the cpu_baseline leg (N > 1, --no-cpu-baseline): check this many rows of the combined film "
                         "against the CPU oracle on rank 0 (untimed)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal knob (not used by the driver): DMT_BENCH_BACKEND=gloo lets several ranks share one GPU
    # to exercise the N > 1 code path on a 1-GPU box; the real runs are one rank per GPU over RCCL.
    backend = os.environ.get("DMT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as graft
    pkg = graft.load_package()

    width, height, spp, max_depth, scene_kind = WORKLOADS[args.workload]
    kspp = args.kspp if args.kspp > 0 else spp
    use_bvh = scene_kind != "cornell"
    scene = build_scene(pkg, scene_kind, width, height)

    r = pkg.Renderer(dev_index)
    # one explicit (non-null) stream for everything: film zeroing, kernels + their HIP timing events,
    # and the RCCL reduce are ordered on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    r.set_stream(stream.cuda_stream)
    r.upload_scene(scene)
    r.set_limits(max_depth)
    if use_bvh or args.cornell_bvh:
        r.set_accel(1)
    r.set_partition(rank, world)
    film = torch.zeros((2, height, width, 4), dtype=torch.float32, device=dev)  # one allocation: one reduce per step
    mean, m2 = film[0], film[1]
    r.film_bind(mean.data_ptr(), m2.data_ptr())

    # N > 1: stream events around every combine, so that a scaling curve can be decomposed per rank into
    # kernel time (the library's own HIP events), combine time and the wait for the slowest rank
    comb_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if world > 1 else []

    def step(timed_index=None):
        film.zero_()
        for s0 in range(0, spp, kspp):
            r.render(min(kspp, spp - s0), sample_offset=s0)
        if timed_index is not None and comb_ev:
            comb_ev[timed_index][0].record(stream)
        pkg.multigpu.combine_films(mean, m2, dst=0, film=film)  # disjoint tiles + zero frames: SUM-reduce == exact gather
        if timed_index is not None and comb_ev:
            comb_ev[timed_index][1].record(stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    r.kernel_time(reset=True)
    r.sched_diag(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    t_local = time.perf_counter() - t0      # this rank's own K steps (render + combine), before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    t_barrier = elapsed - t_local
    kernel_ms, launches = r.kernel_time(reset=True)
    sched = r.sched_diag()   # syncs; hand-over counters of the timed launches
    r.sync()   # every rank: raises DmtError if a launch did not fold every sample chunk exactly once -> non-zero exit, no JSON line
    per_rank = None
    if world > 1:
        combine_ms = sum(a.elapsed_time(b) for a, b in comb_ev)
        mine = torch.tensor([elapsed, kernel_ms / args.steps, combine_ms / args.steps, t_local / args.steps * 1e3, t_barrier * 1e3,
                             sched["handed_over"], sched["folded_for_others"], sched["slab_stalls"], sched["early_exits"],
                             sched["max_stall_ticks_10ns"]], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu().numpy()
        elapsed = float(rows[:, 0].max())   # MAX over ranks
        per_rank = {"kernel_ms_per_step": [round(float(x), 3) for x in rows[:, 1]],
                    "combine_ms_per_step": [round(float(x), 3) for x in rows[:, 2]],
                    "local_ms_per_step": [round(float(x), 3) for x in rows[:, 3]],
                    "final_barrier_wait_ms": [round(float(x), 3) for x in rows[:, 4]],
                    "note": "kernel = the library's HIP events around each megakernel launch; combine = stream events around the "
                            "reduce (on rank 0 it includes waiting for the slowest sender); local = this rank's K steps on its own "
                            "clock; final barrier wait = time spent in the closing barrier"}
        sched = {"handed_over": int(rows[:, 5].sum()), "folded_for_others": int(rows[:, 6].sum()), "slab_stalls": int(rows[:, 7].sum()),
                 "early_exits": int(rows[:, 8].sum()), "max_stall_ticks_10ns": int(rows[:, 9].max()),
                 "folds": None, "launched": None, "slabs_per_wave": sched["slabs_per_wave"]}

    total_samples = float(width) * height * spp * args.steps
    value = total_samples / elapsed / 1e6
    exit_code = 0

    if rank == 0:
        film_mean = mean.cpu().numpy()
        film_m2 = m2.cpu().numpy()
        counts_ok = bool((film_m2[..., 3] == spp).all()) and bool(np.isfinite(film_mean).all())
        info = r.kernel_info()

        cpu_baseline = None
        parity = None
        stats = None
        if use_bvh:
            # BVH path: node visits / triangle tests come from the counting build of the same kernel
            # (1 spp over the whole frame; the film is rebuilt by nothing afterwards -- timing is done)
            r.set_partition(0, 1)
            bvh_stats = r.render_stats(1, sample_offset=spp)
            stats = {"samples": bvh_stats["samples"], "tri_tests": bvh_stats["tri_tests"],
                     "bounces": bvh_stats["bounces"], "hits": bvh_stats["bounces"],
                     "node_visits": bvh_stats["node_visits"], "closest_rays": bvh_stats["closest_rays"],
                     "shadow_rays": bvh_stats["shadow_rays"]}
            if not args.no_cpu_baseline and world == 1:
                # the reference arithmetic is brute force: every triangle tested per ray.  A handful of samples only.
                O = graft.load_oracle()
                oscene = O.Scene(scene.xs, scene.ys, scene.zs, scene.mat_id, scene.bsdfs, scene.lights,
                                 scene.inf_lights, scene.camera)
                if getattr(scene, "env_rgb", None) is not None:
                    oscene.set_envmap(scene.env_rgb, scene.env_quat, scene.env_scale)
                threads = int(os.environ.get("DMT_CPU_THREADS", min(os.cpu_count() or 1, 16)))
                # bounded sample: ~6e9 triangle tests (10-30 s on 16 host threads), from the GPU's own ray counters
                rays = (stats["closest_rays"] + stats["shadow_rays"]) / max(1.0, stats["samples"])
                want = max(16.0, 6e9 / max(1.0, rays * scene.tri_count))
                ospp = int(min(spp, max(1, want // width)))
                npx = int(min(width, max(16, want // ospp)))
                nrows = int(min(height, max(1, round(want / (ospp * npx)))))
                x0, y = width // 2 - npx // 2, height // 2 - nrows // 2
                tc = time.perf_counter()
                omean, om2, ost = O.render(oscene, ospp, max_depth=max_depth, region=(x0, y, x0 + npx, y + nrows),
                                           threads=threads, want_stats=True)
                tcpu = time.perf_counter() - tc
                cpu_baseline = {
                    "value": round(ost["samples"] / tcpu / 1e6, 8), "unit": "Msamples/s", "cores": threads, "kind": "port",
                    "cpu": cpu_model(),
                    "sample": f"{npx} x {nrows} pixels x {ospp} spp around the frame centre ({ost['samples']} samples, {tcpu:.1f} s); the "
                              f"reference arithmetic is a brute-force loop over all {scene.tri_count:,} triangles per ray "
                              f"(megakernel.cu:121-133), no BVH",
                }
                if ospp == spp:
                    d = film_mean[y:y + nrows, x0:x0 + npx, :3].astype(np.float64) - omean[y:y + nrows, x0:x0 + npx, :3]
                    parity = {"rmse_vs_cpu_rows": float(np.sqrt((d ** 2).mean(axis=2)).mean()), "tolerance": 1e-3}
        elif not args.no_cpu_baseline and world == 1:
            O = graft.load_oracle()          # cpu_baseline leg: the oracle is the thing timed
            # the 1-GPU box exposes 256 logical CPUs but the job's CPU share is 16 cores
            threads = int(os.environ.get("DMT_CPU_THREADS", min(os.cpu_count() or 1, 16)))
            # (1) timed: BASELINE configs[0] / BASELINE.md "CPU-baseline plan": the whole 512 x 512 x 64 spp frame, cap 4
            cw, ch, cspp, cdepth, _ = WORKLOADS[C1_WORKLOAD]
            c1scene = O.cornell_box(cw, ch)
            tc = time.perf_counter()
            _, _, c1st = O.render(c1scene, cspp, max_depth=cdepth, threads=threads, want_stats=True)
            tcpu = time.perf_counter() - tc
            cpu_baseline = {
                "value": round(c1st["samples"] / tcpu / 1e6, 4), "unit": "Msamples/s", "cores": threads,
                "kind": "port", "cpu": cpu_model(),
                "sample": f"BASELINE configs[0] in full: cornellBox {cw}x{ch}, {cspp} spp, bounce cap {cdepth} "
                          f"({c1st['samples']} samples, {tcpu:.1f} s); CPU restatement of the reference arithmetic "
                          f"(oracle/dmt_oracle.cpp, g++ -O2 -ffp-contract=off), 32x32-tile std::thread pool",
            }
            # (2) untimed: rows of THIS workload's frame -> work counters for the models + a parity band for the GPU film
            parity, stats = oracle_band(O, width, height, spp, max_depth, args.cpu_band_rows, threads, film_mean)
        elif args.parity_rows > 0:
            # no cpu_baseline leg (N > 1 or --no-cpu-baseline), but the combined film is still checked against the CPU oracle on a few rows
            O = graft.load_oracle()
            threads = int(os.environ.get("DMT_CPU_THREADS", min(os.cpu_count() or 1, 16)))
            parity, stats = oracle_band(O, width, height, spp, max_depth, args.parity_rows, threads, film_mean)
        if stats is None:
            stats = WORKLOAD_STATS.get(args.workload)
        roofline = None
        if launches > 0:
            roofline = build_roofline(args.workload, stats, kernel_ms / launches, float(width) * height * kspp / world, kspp,
                                      load_pmc(args.workload, kspp), info, world)
            roofline["launches"] = int(launches)

        secondary = None
        if world == 1 and args.workload == DEFAULT_WORKLOAD and not args.no_secondary:
            try:
                secondary = [run_secondary(pkg, dev_index, stream.cuda_stream)]
            except Exception as e:   # the headline line is printed regardless; the failure is part of the record
                secondary = [{"workload": SECONDARY_WORKLOAD, "error": f"{type(e).__name__}: {e}"}]
        scene_text = {
            "cornell": "cornellBox() (26 triangles, spot + constant env)",
            "random1M": "1,000,000 random triangles, splitmix64 seed 0x5EED1234 (SURVEY 8d), Cornell BSDFs, spot + env",
            "random16M": "16,000,000 random triangles, same generator and density (cube of centroids 2.52x wider): BVH nodes + leaves = 0.96 GB, four times the 256 MiB Infinity Cache",
            "c3_assets": f"the reference's scenes/sphere.fbx ({scene.tri_count} triangles, GGX conductor) under "
                         "scenes/veranda_polyhaven_1k.png as importance-sampled env map (A18), spot light; scenes/fbx_example.json, "
                         "mesh in metres (tests/test_configs_gpu.py docstring)",
            "sphere_env": f"UV sphere ({scene.tri_count:,} triangles incl. ground plane, GGX conductor + Oren-Nayar), "
                          "synthetic 1024x512 HDR sky as importance-sampled env map (A18), one spot light"}[scene_kind]
        out = {
            "metric": "Msamples/s (paths x spp / s)", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_text,
                       "width": width, "height": height, "spp": spp, "max_depth": max_depth, "kspp": kspp,
                       "accel": "bvh4" if (use_bvh or args.cornell_bvh) else "brute_force", "partition": f"interleaved 8x8 tiles over {world} GPU(s)",
                       "combine": ("none" if world == 1 else
                                   "gather of owned 8x8 tiles on rank 0 (grouped send/recv; DMT_COMBINE=gather)"
                                   if os.environ.get("DMT_COMBINE", "reduce") == "gather" else
                                   "rccl reduce(sum) of zero-initialised mean/M2 frames to rank 0 (exact gather: disjoint tiles)")},
            "film_ok": counts_ok,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
            "per_rank": per_rank,
            "secondary": secondary,
            "fold_handover": sched,   # in-launch ordered fold without waiting: chunks handed to another wave, stalls, early exits
        }
        print(json.dumps(out), flush=True)
        if not counts_ok or (parity is not None and not parity["rmse_vs_cpu_rows"] < parity["tolerance"]):
            print("bench.py: film check FAILED (sample counts / finiteness / parity band)", file=sys.stderr, flush=True)
            exit_code = 3
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
