#!/usr/bin/env python3
"""Distribution of the two-view errors of bench.py's workload against the generator's ground truth (per pair: max |R - R_gt|,
t . t_gt, map points) - the numbers behind the bounds of tests/test_gpu_dropin.py::test_bench_workload_all_pairs_properties."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import vslam_amd as V  # noqa: E402
from vslam_amd import synth  # noqa: E402
from tests.test_gpu_dropin import _batch_io  # noqa: E402

nb, cap = 256, 2048
dev = torch.device("cuda", 0)
frames = bench.make_frames(torch, dev, 0, nb)
scene = synth.Survey8dScene(torch, torch.device("cpu"))
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
ctx.set_stream(st.cuda_stream)
prm = V.orb_params(nfeatures=2000, fast_threshold=7)
io, b, K = _batch_io(torch, V, dev, frames, nb, cap, 4096, want_mask=True)
ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
st.synchronize()
P, NP = b["pose"].cpu().numpy(), b["npts"].cpu().numpy()
eR, dt = [], []
for i in range(nb - 1):
    Rg, tg = scene.relative_pose(i, i + 1)
    eR.append(np.abs(P[i, :9].reshape(3, 3) - Rg).max()); dt.append(float(P[i, 9:] @ tg))
eR, dt = np.array(eR), np.array(dt)
q = [0, 1, 5, 25, 50, 75, 95, 99, 100]
print("err_R percentiles", dict(zip(q, np.round(np.percentile(eR, q), 5))))
print("t.t_gt percentiles", dict(zip(q, np.round(np.percentile(dt, q), 4))))
print("map points percentiles", dict(zip(q, np.percentile(NP, q))))
print("matches", float(b["mpass"].sum(dim=1).float().mean().item()))
