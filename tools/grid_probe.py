#!/usr/bin/env python3
"""tools/grid_probe.py [n_calls]: the batched grid detector (MO_DETECT_GRID) on bench.py's 256 frames, for
`rocprofv3 --kernel-trace --stats -- python3 tools/grid_probe.py` (per-kernel times of the grid stage)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import vslam_amd as V  # noqa: E402
from tests.test_gpu_dropin import _batch_io  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nb, cap = 256, 2048
dev = torch.device("cuda", 0)
cache = "/tmp/bench_frames.survey8d.0.256.npy"
frames = torch.from_numpy(np.load(cache)).to(dev) if os.path.exists(cache) else bench.make_frames(torch, dev, 0, nb)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
ctx.set_stream(st.cuda_stream)
prm = V.orb_params(nfeatures=2000, fast_threshold=7)
io, b, K = _batch_io(torch, V, dev, frames, nb, cap, 4096)
io.detector = V.DETECT_GRID
for _ in range(n_calls):
    ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
st.synchronize()
print(ctx.stage_times())
