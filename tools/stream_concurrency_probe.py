#!/usr/bin/env python3
"""tools/stream_concurrency_probe.py -- do kernels on two HIP streams run at the same time at all on this box?
torch.cuda._sleep spins ONE workgroup for a fixed number of cycles: two of them on two streams take 1x when the streams are
mapped to different hardware queues and 2x when they are serialised."""
import time
import torch

dev = torch.device("cuda", 0)
torch.cuda._sleep(1000)
N = 4
streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
cyc = 20_000_000


def run(k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in streams[:k]:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


for k in (1, 2, 3, 4):
    print("%d streams x one spinning workgroup: %.2f ms" % (k, run(k)))
