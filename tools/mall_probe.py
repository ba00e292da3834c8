#!/usr/bin/env python3
"""Is a freshly written / re-read buffer served by the 256 MB Infinity Cache?  Read bandwidth of torch.sum over buffers
of growing size, re-read back to back (warm) and right after a producer kernel wrote them (y.mul_)."""
import time
import torch

dev = "cuda"
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    y = torch.ones(n, dtype=torch.float32, device=dev)
    for _ in range(3):
        y.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        y.sum()
    e1.record()
    torch.cuda.synchronize()
    warm = e0.elapsed_time(e1) / reps
    # producer then consumer: time only the consumer
    tot = 0.0
    for _ in range(reps):
        y.mul_(1.0)
        e0.record()
        y.sum()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    after = tot / reps
    print(f"{mb:5d} MB: re-read {mb / 1024 / warm * 1e3:8.1f} GB/s   read after write {mb / 1024 / after * 1e3:8.1f} GB/s")
