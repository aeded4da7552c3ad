"""one single-pair tracking call (and one 16-frame batch) with a stamp-printing variant of the library: device printf lines = phase stamps"""
import sys, os
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import vslam_amd as V
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 2, scene="survey8d").cpu().numpy()
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
prm = V.orb_params(nfeatures=2000)
(ka, da), = ctx.orb_detect_compute(fr[0], prm)
(kb, db), = ctx.orb_detect_compute(fr[1], prm)
for i in range(3):
    print("---- track call", i, flush=True)
    ctx.track_pair(ka, da, kb, db, 640, 480, K)
for i in range(2):
    print("---- init call", i, flush=True)
    ctx.pair_frontend(ka, da, kb, db, V.MODE_INIT, K)
ctx.close()
