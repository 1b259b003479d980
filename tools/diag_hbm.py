#!/usr/bin/env python3
"""Diagnostic (GPU box): achievable HBM bandwidth of this box with plain streaming kernels (torch), to put next to the
datasheet-derived 8 TB/s that bench.py's roofline.peak uses."""
import torch
dev = torch.device("cuda", 0)
n = 2 * 1024 ** 3  # floats: 8 GiB per tensor
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def timed(fn, bytes_moved, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return bytes_moved / (ms * 1e-3) / 1e12, ms
for name, fn, moved in (("copy (read + write)", lambda: b.copy_(a), 2 * 4 * n), ("read only (sum)", lambda: a.sum(), 4 * n),
                        ("write only (fill)", lambda: b.fill_(1.0), 4 * n), ("triad b = a * 2 + b", lambda: b.add_(a, alpha=2.0), 3 * 4 * n)):
    tbs, ms = timed(fn, moved)
    print(f"{name:24s} {tbs:6.2f} TB/s   ({ms:.2f} ms)")
