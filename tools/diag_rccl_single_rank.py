"""Diagnostic (GPU box): RCCL initialises and reduces on this image with the stream set-up bench.py uses (one rank)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
s=torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
a=torch.ones((1024,1024,4),device=dev); dist.reduce(a,0); dist.barrier(); torch.cuda.synchronize()
print("rccl single-rank ok", float(a.sum()))
dist.destroy_process_group()
