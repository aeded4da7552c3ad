#!/usr/bin/env python3
"""cProfile of examples/run_frames.py --batch 64 over a 4096-frame stack: where the caller-side Python time per chunk goes
(tools/stream_delay.py: the stream is GPU-bound, 69 k frames/s, while the caller spends < 0.6 ms per chunk)."""
import cProfile, contextlib, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd", "examples"))
import numpy as np, torch
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
path = "/tmp/frame_stack4k.npy"
np.save(path, np.concatenate([fr, fr[::-1]] * 8))
import run_frames
argv = ["--frames", path, "--max-frames", "0", "--batch", "64"] + sys.argv[1:]
with contextlib.redirect_stdout(io.StringIO()):
    run_frames.main(argv)
pr = cProfile.Profile(); buf = io.StringIO()
pr.enable()
with contextlib.redirect_stdout(buf):
    run_frames.main(argv)
pr.disable()
print(buf.getvalue().strip().split("\n")[-1])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:7000])
