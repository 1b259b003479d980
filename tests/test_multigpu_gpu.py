"""The N > 1 path of bench.py on the HIP library, rehearsed on ONE GPU: two or three gloo ranks share the device (RCCL
refuses two ranks on one GPU, tools/diag_rccl_two_ranks_one_gpu.py), each renders its interleaved tiles with the
megakernel, the frames are combined on rank 0 and rank 0 checks the film (sample counts everywhere, a 16-row band against
the CPU oracle).  What differs from the driver's run is the transport (gloo stages through the host) -- partition, launch,
hand-over fold under device sharing, combine (reduce and gather), per-rank records and the JSON line are the same code."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(world, port, combine, workload="cornell_512x512_64spp_4bounces"):
    env = dict(os.environ, DMT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", DMT_COMBINE=combine)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--parity-rows", "6", "--workload", workload]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=420, cwd=str(ROOT))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("world,combine", [(2, "reduce"), (3, "gather")])
def test_gloo_ranks_share_the_gpu_through_bench(world, combine):
    d = _bench(world, 29540 + world, combine)
    assert d["n_gpus"] == world and d["steps"] == 2 and d["value"] > 0
    assert d["film_ok"] is True
    assert d["parity"]["rows"] == 6 and d["parity"]["rmse_vs_cpu_rows"] < 1e-4      # the COMBINED film vs the CPU oracle (observed 3e-7)
    pr = d["per_rank"]
    assert len(pr["kernel_ms_per_step"]) == world and all(x > 0 for x in pr["kernel_ms_per_step"])
    assert len(pr["combine_ms_per_step"]) == world
    fh = d["fold_handover"]
    assert fh["early_exits"] == 0                      # no wave gave up (round 2's failure mode under device sharing)
    assert combine in json.dumps(d["config"])
