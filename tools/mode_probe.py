#!/usr/bin/env python3
"""tools/mode_probe.py -- mo_dev_frontend_batch in MO_MODE_INIT and MO_MODE_TRACK on the bench frames, on a first and on a second
context: per-stage times (is a slow stage a property of the mode or of the context?)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd"))
import bench  # noqa: E402
import vslam_amd as V  # noqa: E402

B, W, H, CAP = 256, 640, 480, 2048
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, dev, 0, B)
prm = V.orb_params(nfeatures=2000, fast_threshold=7, select_order=V.ORDER_LIBSTDCXX)
main = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(main)
K = [320.0, 0, 320.0, 0, 320.0, 240.0, 0, 0, 1.0]


class Run:
    def __init__(self, mode, thr):
        n = B
        self.ctx = V.Context(device=0, max_w=W, max_h=H, max_batch=n)
        self.ctx.set_stream(main.cuda_stream)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.t = [z((n, CAP, 7), torch.float32), z((n, CAP, 32), torch.uint8), z(n, torch.int32), z((n - 1, CAP, 2), torch.int32),
                  z((n - 1, CAP, 2), torch.int32), z((n - 1, CAP), torch.uint8), z((n - 1, 12), torch.float64), z((n - 1, CAP, 3), torch.float32),
                  z(n - 1, torch.int32), z((n - 1, CAP, 2), torch.int32), z(n - 1, torch.int32)]
        io = V.BatchIO()
        io.d_gray = frames.data_ptr(); io.w = W; io.h = H; io.batch = n; io.cap = CAP
        io.ratio = 0.75; io.thr_px = thr; io.n_hyp = 4096; io.seed = 4096
        for i in range(9):
            io.K[i] = K[i]
        (io.d_kps, io.d_desc, io.d_counts, io.d_match_idx, io.d_match_dist, io.d_match_pass, io.d_pose, io.d_points, io.d_n_points,
         io.d_sel_idx, io.d_sel_n) = [x.data_ptr() for x in self.t]
        io.mode = mode; io.disp_frac = 0.02
        self.io = io

    def time(self, label, steps=20):
        for _ in range(40):
            self.ctx._check(self.ctx.lib.mo_dev_frontend_batch(self.ctx.h, C.byref(prm), C.byref(self.io)))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            self.ctx._check(self.ctx.lib.mo_dev_frontend_batch(self.ctx.h, C.byref(prm), C.byref(self.io)))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        acc = {}
        for back in range(steps):
            for name, v in self.ctx.stage_times(back):
                acc[name] = acc.get(name, 0.0) + v / steps
        print(label, round(ms, 3), {k: round(v, 3) for k, v in acc.items()})


a = Run(V.MODE_INIT, 3.0); a.time("first context, init ")
b = Run(V.MODE_TRACK, 1.0); b.time("second context, track")
c = Run(V.MODE_INIT, 3.0); c.time("third context, init ")
a.time("first context again ")
b.io.mode = V.MODE_INIT; b.io.thr_px = 3.0; b.time("second context, init ")
