#!/usr/bin/env python3
"""examples/run_frames.py WITHOUT --batch (one frame at a time through the drop-in classes, the reference's call pattern: tester_map.py:57-75)
on the frame stack tools/stream_rate.py uses, three runs per detector: the per-frame loop's rate beside the chunked mode's."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd", "examples"))
import numpy as np, torch
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
path = "/tmp/frame_stack2k.npy"
np.save(path, np.concatenate([fr, fr[::-1]] * 4))
import run_frames
for extra in ([], ["--grid"]):
    for r in range(3):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            run_frames.main(["--frames", path, "--max-frames", "2048"] + extra)
        print("per frame", extra, buf.getvalue().strip().split("\n")[-1], flush=True)
