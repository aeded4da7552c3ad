import os, sys, numpy as np, cProfile, pstats, io, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, "."); sys.path.insert(0, "visual-slam_amd/examples")
import torch
from vslam_amd import synth
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
path = "/tmp/frame_stack4k.npy"
np.save(path, np.concatenate([fr, fr[::-1]] * 8))
import run_frames
import contextlib
import re
rep = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 1
for extra in ([], ["--grid"], ["--batch-size-128"]):
    b = "128" if extra == ["--batch-size-128"] else "64"
    rates = []
    for r in range(rep):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            run_frames.main(["--frames", path, "--max-frames", "0", "--batch", b] + (extra if extra == ["--grid"] else []))
        last = buf.getvalue().strip().split("\n")[-1]
        rates.append(float(re.search(r"(\d+) frames/s from the fourth chunk on", last).group(1)))
        print("batch", b, extra, last, flush=True)
    if rep > 1:
        rs = sorted(rates)
        print("batch %s %s steady-state frames/s over %d runs of 4096 frames: min %.0f  median %.0f  max %.0f  | all: %s"
              % (b, extra, rep, rs[0], rs[len(rs) // 2], rs[-1], " ".join("%.0f" % x for x in rates)), flush=True)
