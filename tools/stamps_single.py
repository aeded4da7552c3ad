"""one single-frame extraction with a stamp-printing variant of the library (VSLAM_AMD_LIB=visual-slam_amd/variants/lib<x>.so):
the device printf lines of the LAST of a few calls are the batch-1 phase stamps"""
import sys, os
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import vslam_amd as V
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 1, scene="survey8d").cpu().numpy()
ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
prm = V.orb_params(nfeatures=2000)
for i in range(4):
    print("---- call", i, flush=True)
    ctx.orb_detect_compute(fr[0], prm)
    ctx.sync()
ctx.close()
