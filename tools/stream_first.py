#!/usr/bin/env python3
"""The first FrameStream of a process: per chunk, how long submit took (staging into pinned memory + enqueue) and how long collect blocked
(waiting for the GPU), beside the same for a second stream - which side is slow while the first one warms up?"""
import sys, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from vslam_amd import synth
from vslam_amd.stream import FrameStream
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
stack = np.concatenate([fr, fr[::-1]] * 4)
chunk = 64
import gc
if "--nogc" in sys.argv:
    gc.disable()
stalls = []
for run in (1, 2):
    fs = FrameStream(K, chunk=chunk, n_features=2000, cap=2112, copy=False)
    ts, tc, marks = [], [], []
    for k in range(0, len(stack), chunk):
        t = time.perf_counter(); fs.submit(stack[k:k + chunk]); ts.append((time.perf_counter() - t) * 1e3)
        if fs._in_flight == fs.lanes:
            t = time.perf_counter(); fs.collect(); tc.append((time.perf_counter() - t) * 1e3); marks.append(time.perf_counter())
    while fs._in_flight: fs.collect()
    per = np.diff(np.array(marks)) * 1e3
    print("stream %d  chunk: period ms | submit ms | collect blocked ms" % run)
    for i in range(0, min(len(per), 28), 1):
        print("  %2d: %.3f | %.3f | %.3f" % (i + 3, per[i], ts[i + 3], tc[i + 1]), flush=True)
    stalls.append((run, [(i, round(x, 2)) for i, x in enumerate(ts) if x > 2.0]))
    fs.close()
print("gc %s | submits longer than 2 ms (stream, [(chunk, ms)]):" % ("disabled" if "--nogc" in sys.argv else "enabled"), stalls, "| gc counts", gc.get_count(), "collections", [g["collections"] for g in gc.get_stats()])
