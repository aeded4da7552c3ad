#!/usr/bin/env python3
"""Is a FrameStream caller's own time per chunk hidden behind the GPU?  Chunk 64 over a 4096-frame stack; after every collect the caller
busy-waits `delay` ms (standing in for whatever it does with the results) before it submits the next chunk.  Per delay: steady-state rate
(from the fourth chunk on), how long collect blocked (> 0: the caller waits for the GPU, its own time is hidden; ~ 0: the GPU waits for the
caller) and the submit time.  python tools/stream_delay.py [--chunk 64]"""
import sys, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from vslam_amd import synth
from vslam_amd.stream import FrameStream
chunk = int(sys.argv[sys.argv.index("--chunk") + 1]) if "--chunk" in sys.argv else 64
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 256, scene="survey8d").cpu().numpy()
stack = np.concatenate([fr, fr[::-1]] * 8)
for rep in range(2):
    for delay in (0.0, 0.2, 0.4, 0.6, 0.8, 1.2):
        fs = FrameStream(K, chunk=chunk, n_features=2000, cap=2048, copy=False)
        ts, tc, n, t_steady, n_steady = [], [], 0, None, 0
        for k in range(0, len(stack), chunk):
            t = time.perf_counter(); fs.submit(stack[k:k + chunk]); ts.append(time.perf_counter() - t)
            if fs._in_flight == fs.lanes:
                t = time.perf_counter(); res = fs.collect(); tc.append(time.perf_counter() - t)
                n += len(res)
                if len(tc) == 3: t_steady, n_steady = time.perf_counter(), n
                t = time.perf_counter()
                while (time.perf_counter() - t) * 1e3 < delay: pass
        t_end = time.perf_counter()
        while fs._in_flight: n += len(fs.collect())
        rate = (n - n_steady - (fs.lanes - 1) * 0) / (t_end - t_steady) if t_steady else 0.0
        print("chunk %d delay %.1f ms: %6.0f frames/s steady  period %.3f ms | collect blocked median %.3f ms, submit median %.3f ms"
              % (chunk, delay, rate, (t_end - t_steady) / max(len(tc) - 3, 1) * 1e3, np.median(tc[3:]) * 1e3, np.median(ts[3:]) * 1e3), flush=True)
        fs.close()
