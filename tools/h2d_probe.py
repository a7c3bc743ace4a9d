#!/usr/bin/env python3
"""Probe (GPU box): pinned-host -> device copy rate of one 16 x 608 x 608 x 3 uint8 batch, alone and under compute."""
import time
import torch
dev = torch.device("cuda:0")
h = torch.randint(0, 255, (16, 608, 608, 3), dtype=torch.uint8).pin_memory()
d = [torch.empty_like(h, device=dev) for _ in range(3)]
torch.cuda.synchronize()
for n in (1, 3):
    ss = [torch.cuda.Stream() for _ in range(n)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(60):
            with torch.cuda.stream(ss[i % n]):
                d[i % n].copy_(h, non_blocking=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    print("%d stream(s): %.3f ms per 17.7 MB copy = %.1f GB/s" % (n, dt * 1e3, h.numel() / dt / 1e9))
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1):
    for _ in range(20): a @ a
torch.cuda.synchronize(); t_mm = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1):
    for _ in range(20): a @ a
with torch.cuda.stream(s2):
    for i in range(20): d[0].copy_(h, non_blocking=True)
torch.cuda.synchronize(); t_both = time.perf_counter() - t0
print("20 matmuls %.2f ms; with 20 concurrent copies on another stream %.2f ms" % (t_mm * 1e3, t_both * 1e3))
