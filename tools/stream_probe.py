"""where a FrameStream chunk's time goes: wall of submit / collect per chunk, device stage spans of the chunk's batched call, rate by chunk size"""
import sys, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
import vslam_amd as V
from vslam_amd import synth
from vslam_amd.stream import FrameStream
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
stack = np.concatenate([fr, fr[::-1], fr, fr[::-1]])
for chunk in (32, 64, 128, 256):
    fs = FrameStream(K, chunk=chunk, n_features=2000, cap=2048, copy=False)
    list(fs.run(stack[:2 * chunk + 1]))
    ts, tc = [], []
    t0 = time.perf_counter()
    n = 0
    for k in range(0, len(stack), chunk):
        t = time.perf_counter(); fs.submit(stack[k:k + chunk]); ts.append(time.perf_counter() - t)
        if fs._in_flight == 3:
            t = time.perf_counter(); res = fs.collect(); tc.append(time.perf_counter() - t)
            for r in res:
                p = r.pair; n += 1
    while fs._in_flight:
        t = time.perf_counter(); res = fs.collect(); tc.append(time.perf_counter() - t); n += len(res)
    el = time.perf_counter() - t0
    st = fs.ctx.stage_times()
    print("chunk %3d: %.0f frames/s  %.3f ms per chunk | submit median %.3f ms, collect median %.3f ms | device stages of one chunk: %s = %.3f ms"
          % (chunk, n / el, el / (len(stack) / chunk) * 1e3, np.median(ts) * 1e3, np.median(tc) * 1e3,
             " ".join("%s %.3f" % (a, b) for a, b in st), sum(b for _, b in st)), flush=True)
    fs.close()
