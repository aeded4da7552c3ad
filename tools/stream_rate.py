import os, sys, numpy as np, cProfile, pstats, io, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, "."); sys.path.insert(0, "visual-slam_amd/examples")
import torch
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
path = "/tmp/frame_stack4k.npy"
np.save(path, np.concatenate([fr, fr[::-1]] * 8))
import run_frames
import contextlib
for extra in ([], ["--grid"], ["--batch-size-128"]):
    b = "128" if extra == ["--batch-size-128"] else "64"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        run_frames.main(["--frames", path, "--max-frames", "0", "--batch", b] + (extra if extra == ["--grid"] else []))
    print("batch", b, extra, buf.getvalue().strip().split("\n")[-1])
