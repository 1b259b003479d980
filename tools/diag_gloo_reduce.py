"""Diagnostic (GPU box): cost of gloo reduce on CUDA tensors (the backend of the N > 1 REHEARSAL only; real runs use RCCL).
Launch: python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 tools/diag_gloo_reduce.py"""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda", 0)
for mib in (1, 16, 64):
    t = torch.ones(mib * 262144, device=dev)
    dist.reduce(t, 0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.reduce(t, 0); torch.cuda.synchronize()
    if rank == 0:
        print(f"world {dist.get_world_size()}  {mib:3d} MiB  reduce {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
dist.destroy_process_group()
