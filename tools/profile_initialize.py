#!/usr/bin/env python3
"""cProfile of MapInitializer.initialize through the drop-in classes (one device call + the Python around it) and, beside it, the wall of the
device call alone: where the 0.6 ms of the class call go.  python tools/profile_initialize.py [--iters 300]"""
import argparse, cProfile, contextlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import vslam_amd as V
from vslam_amd import synth
from orbslam2.extractor import ORBExtractor
from orbslam2.initializer import MapInitializer
from orbslam2.matcher import DescriptorMatcher

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=300); a = ap.parse_args()
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 2, scene="survey8d").cpu().numpy()
f0, f1 = np.ascontiguousarray(fr[0]), np.ascontiguousarray(fr[1])
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
ex = ORBExtractor(n_features=2000); mt = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)
(k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)

def run():
    ini = MapInitializer(K)
    ini.set_first_frame(k0, d0, f0)
    with contextlib.redirect_stdout(io.StringIO()):
        return ini.initialize(k1, d1, mt, f1)

for _ in range(10): run()
ts = []
for _ in range(a.iters):
    t = time.perf_counter(); run(); ts.append((time.perf_counter() - t) * 1e3)
print("initialize through the classes: median %.4f ms (min %.4f)" % (float(np.median(ts)), min(ts)))
ctx = V.default_context()
ts = []
for _ in range(a.iters):
    t = time.perf_counter(); ctx.pair_frontend(k0.array, d0, k1.array, d1, V.MODE_INIT, K, 640, 480, thr_px=3.0); ts.append((time.perf_counter() - t) * 1e3)
print("device call alone (pair_frontend MODE_INIT, resident frames): median %.4f ms" % float(np.median(ts)))
pr = cProfile.Profile(); pr.enable()
for _ in range(a.iters): run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
