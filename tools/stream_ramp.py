#!/usr/bin/env python3
"""Does a FrameStream start slow after the GPU sat idle?  Chunk 64 over a 4096-frame stack, a new stream per run, the GPU left idle for
`idle` ms before each; rate over the four quarters of the steady part of the run (chunks 4 .. 63) and over all of it.
python tools/stream_ramp.py"""
import sys, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from vslam_amd import synth
from vslam_amd.stream import FrameStream
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
stack = np.concatenate([fr, fr[::-1]] * 8)
chunk = 64
for idle in (0, 0, 300, 300, 1000, 1000, 0, 0):
    torch.cuda.synchronize(); time.sleep(idle / 1e3)
    fs = FrameStream(K, chunk=chunk, n_features=2000, cap=2112, copy=False)
    marks = []
    for k in range(0, len(stack), chunk):
        fs.submit(stack[k:k + chunk])
        if fs._in_flight == fs.lanes:
            fs.collect(); marks.append(time.perf_counter())
    while fs._in_flight: fs.collect(); marks.append(time.perf_counter())
    m = np.array(marks[3:])
    q = len(m) // 4
    rates = [chunk * (q - 1) / (m[(i + 1) * q - 1] - m[i * q]) for i in range(4)]
    print("idle %4d ms before the stream: quarters %s frames/s | chunks 4..: %.0f frames/s" % (idle, " ".join("%6.0f" % r for r in rates), chunk * (len(m) - 1) / (m[-1] - m[0])), flush=True)
    fs.close()
