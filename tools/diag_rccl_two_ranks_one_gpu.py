"""Diagnostic (GPU box): can two RCCL ranks share ONE GPU on this image?  (A rehearsal of bench.py --gpus 2 over RCCL on a
1-GPU box needs it.)  Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/diag_rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=dev)
    a = torch.full((1024,), float(rank + 1), device=dev)
    dist.reduce(a, 0)
    dist.barrier()
    torch.cuda.synchronize()
    print(f"rank {rank}: ok, a[0] = {float(a[0])}")
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: refused: {type(e).__name__}: {str(e)[:300]}")
