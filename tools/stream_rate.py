"""the --batch driver on a real frame stack file: writes 1024 synthetic frames as .npy, runs examples/run_frames.py --batch 64 on it"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import torch
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
path = "/tmp/frame_stack.npy"
np.save(path, np.concatenate([fr, fr[::-1], fr, fr[::-1]]))
del fr
for extra in ([], ["--grid"]):
    print(subprocess.run([sys.executable, "visual-slam_amd/examples/run_frames.py", "--frames", path, "--max-frames", "0", "--batch", "64"] + extra,
                         capture_output=True, text=True).stdout.strip().split("\n")[-1], flush=True)
