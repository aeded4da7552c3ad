#!/usr/bin/env python3
"""pinned host -> device copy rate of this box for a chunk-sized buffer (65 frames of 640x480: 20 MB) and pageable -> pinned staging rate"""
import time, numpy as np, torch
n = 65 * 640 * 480
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(3): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print("pinned -> device, %.1f MB: %.3f ms = %.1f GB/s" % (n / 1e6, dt * 1e3, n / dt / 1e9), flush=True)
b = torch.empty(n, dtype=torch.uint8, device="cuda"); hb = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): hb.copy_(b, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print("device -> pinned, %.1f MB: %.3f ms = %.1f GB/s" % (n / 1e6, dt * 1e3, n / dt / 1e9), flush=True)
