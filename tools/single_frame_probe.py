import sys, time
sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import numpy as np
import vslam_amd as V
from tests.helpers import synthetic_frame
ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
prm = V.orb_params(nfeatures=2000)
f = synthetic_frame(1)
for _ in range(5): ctx.orb_detect_compute(f, prm)
ts = []
for _ in range(50):
    t = time.perf_counter(); r = ctx.orb_detect_compute(f, prm); ts.append((time.perf_counter() - t) * 1e3)
print("wall ms median %.3f min %.3f" % (sorted(ts)[25], min(ts)))
st = ctx.stage_times()
print("gpu stages", [(n, round(ms, 4)) for n, ms in st], "sum %.3f" % sum(ms for _, ms in st))
(k, d), = r
ts = []
for _ in range(50):
    t = time.perf_counter(); ctx.match_knn2_ratio(d, d, 0.75); ts.append((time.perf_counter() - t) * 1e3)
print("match wall ms median %.3f" % sorted(ts)[25], ctx.stage_times())
