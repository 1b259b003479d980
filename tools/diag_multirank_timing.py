"""Diagnostic (GPU box): where a multi-rank step spends its time when R ranks SHARE one GPU (gloo rehearsal).
Launch: python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 tools/diag_multirank_timing.py"""
import os, sys, time
from pathlib import Path
import torch, torch.distributed as dist
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
pkg = g.load_package()
scene = pkg.host_scene.cornell_box(1024, 1024)
r = pkg.Renderer(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
r.set_stream(stream.cuda_stream)
r.upload_scene(scene); r.set_limits(8); r.set_partition(rank, world)
film = torch.zeros((2, 1024, 1024, 4), device=dev)
r.film_bind(film[0].data_ptr(), film[1].data_ptr())
spp = int(os.environ.get("SPP", "256"))
for it in range(3):
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    film.zero_(); r.render(spp); torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.reduce(film, 0); torch.cuda.synchronize()
    t2 = time.perf_counter()
    dist.barrier()
    t3 = time.perf_counter()
    ms, n = r.kernel_time(reset=True)
    print(f"it {it} rank {rank}/{world}: render+sync {1e3*(t1-t0):9.1f} ms (kernel {ms/max(n,1):8.1f})  reduce {1e3*(t2-t1):9.1f}  barrier {1e3*(t3-t2):9.1f}", flush=True)
r.close()
dist.destroy_process_group()
