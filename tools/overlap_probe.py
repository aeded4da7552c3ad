#!/usr/bin/env python3
"""tools/overlap_probe.py -- do two stages of DIFFERENT batches overlap on one MI355X?  Stream A extracts batch i + 1
(mo_dev_orb_detect_compute), stream B matches + poses batch i (mo_dev_match_pairs on resident descriptors): each alone,
then both enqueued per iteration.  serial = a + b; 'both' close to max(a, b) would mean a software pipeline across batches
pays, 'both' close to a + b that the kernels already fill the vector units."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd"))
import bench  # noqa: E402  (frame generator)
import vslam_amd as V  # noqa: E402

B, W, H, CAP = 256, 640, 480, 2048
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, dev, 0, B)
prm = V.orb_params(nfeatures=2000, fast_threshold=7, select_order=V.ORDER_LIBSTDCXX)


def mk(stream):
    ctx = V.Context(device=0, max_w=W, max_h=H, max_batch=B)
    ctx.set_stream(stream.cuda_stream)
    return ctx


sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ca, cb = mk(sa), mk(sb)
kps = torch.zeros((B, CAP, 7), dtype=torch.float32, device=dev)
desc = torch.zeros((B, CAP, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
desc2, cnt2, kps2 = torch.zeros_like(desc), torch.zeros_like(cnt), torch.zeros_like(kps)
qf = torch.arange(0, B - 1, dtype=torch.int32, device=dev)
tf = torch.arange(1, B, dtype=torch.int32, device=dev)
midx = torch.zeros((B - 1, CAP, 2), dtype=torch.int32, device=dev)
mdist = torch.zeros_like(midx)
mpass = torch.zeros((B - 1, CAP), dtype=torch.uint8, device=dev)


def extract(ctx, k, d, c):
    ctx._check(ctx.lib.mo_dev_orb_detect_compute(ctx.h, C.byref(prm), frames.data_ptr(), W, H, B, k.data_ptr(), d.data_ptr(), CAP,
                                                 c.data_ptr()))


def match(ctx):
    ctx._check(ctx.lib.mo_dev_match_pairs(ctx.h, desc.data_ptr(), cnt.data_ptr(), CAP, qf.data_ptr(), tf.data_ptr(), B - 1, 0.75,
                                          midx.data_ptr(), mdist.data_ptr(), mpass.data_ptr()))


def spins(streams, cyc=20_000_000):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for st in streams:
        with torch.cuda.stream(st):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3


# are the two streams concurrent at all in THIS process (stream -> hardware queue mapping depends on creation order)?
torch.cuda._sleep(1000)
print("one spinning workgroup on stream A: %.2f ms, on A and B: %.2f ms (equal = the two streams run side by side)" % (spins([sa]), spins([sa, sb])))
extract(cb, kps, desc, cnt)  # the descriptors stream B matches
torch.cuda.synchronize()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


a = timed(lambda: extract(ca, kps2, desc2, cnt2))
b = timed(lambda: match(cb))
both = timed(lambda: (extract(ca, kps2, desc2, cnt2), match(cb)))
print("aux probes: context A %s, context B %s" % (ca.aux_probe(), cb.aux_probe()))
print("extract alone %.3f ms, match alone %.3f ms, serial sum %.3f ms, both streams %.3f ms" % (a, b, a + b, both))
# the matcher started together with the extraction only meets its vector-bound head (pyramid, FAST, blur); two and three
# back-to-back match calls reach into the latency-bound tail (selection, describe)
for n in (2, 3):
    bn = timed(lambda: [match(cb) for _ in range(n)])
    both = timed(lambda: (extract(ca, kps2, desc2, cnt2), [match(cb) for _ in range(n)]))
    print("extract + %d x match: serial sum %.3f ms, both streams %.3f ms" % (n, a + bn, both))
ca.close(); cb.close()
