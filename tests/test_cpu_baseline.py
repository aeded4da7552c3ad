"""CPU: the C++ / OpenMP baseline library of bench.py (oracle/cpu_baseline.cpp) agrees with the parity oracle it is a port of."""
import ctypes as C
import os

import numpy as np

from oracle import geom_oracle as G
from oracle import orb_oracle as O
from tests.helpers import synthetic_frame

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libcpu_baseline.so")


def _lib():
    O.build()
    lib = C.CDLL(LIB)
    vp = C.c_void_p
    lib.orc_two_view.restype = C.c_int
    lib.orc_two_view.argtypes = [vp, vp, C.c_int, vp, C.c_double, C.c_int, C.c_uint64, vp, vp]
    lib.orc_baseline_run.restype = C.c_int
    lib.orc_baseline_run.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    return lib


def test_cpp_two_view_matches_numpy_oracle():
    lib = _lib()
    s = G.synthetic_two_view(seed=4096, n=600, outlier_frac=0.3)
    o = G.init_two_view(s["p1"], s["p2"], s["K"], thr_px=3.0, n_hyp=512, seed=4096)
    R = np.zeros(9); t = np.zeros(3)
    K = np.ascontiguousarray(s["K"].reshape(9))
    p1 = np.ascontiguousarray(s["p1"], np.float32); p2 = np.ascontiguousarray(s["p2"], np.float32)
    n = lib.orc_two_view(p1.ctypes.data, p2.ctypes.data, 600, K.ctypes.data, 3.0, 512, 4096, R.ctypes.data, t.ctypes.data)
    assert abs(n - o["n_good"]) <= 2
    assert np.linalg.norm(R.reshape(3, 3) - o["R"]) < 1e-6 and np.linalg.norm(t - o["t"].ravel()) < 1e-6
    assert np.linalg.norm(R.reshape(3, 3) - s["R"]) < 1e-4


def test_baseline_run_counts_match_the_oracle():
    lib = _lib()
    frames = np.stack([synthetic_frame(5), np.roll(synthetic_frame(5), 4, axis=1), synthetic_frame(6)])
    K = np.ascontiguousarray(np.array([320.0, 0, 320, 0, 320, 240, 0, 0, 1]))
    times = (C.c_double * 3)(); counts = (C.c_longlong * 3)()
    O.lib().orc_set_variant(0, 0)
    used = lib.orc_baseline_run(frames.ctypes.data, 3, 640, 480, 1000, 0.75, 1, K.ctypes.data, 64, 2, C.addressof(times),
                                C.addressof(counts))
    assert used == 2 and all(x >= 0 for x in times)
    feats = [O.detect_and_compute(f, O.params(nfeatures=1000)) for f in frames]
    assert counts[0] == sum(len(k) for k, _ in feats)
    nm = 0
    for a, b in zip(feats[:-1], feats[1:]):
        idx, dist = O.match_knn2(a[1], b[1])
        nm += int(O.ratio_test(idx, dist, 0.75).sum())
    assert counts[1] == nm
