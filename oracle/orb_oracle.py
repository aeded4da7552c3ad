"""ctypes loader for the CPU oracle (oracle/orb_oracle.cpp) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (visual-slam_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liborb_oracle.so")


class OrcKeyPoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


class OrcParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("fast_threshold", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "orb_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_harris.restype = C.c_float
        _lib.orc_ic_angle.restype = C.c_float
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def params(nfeatures=2000, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=7):
    return OrcParams(int(nfeatures), float(scale_factor), int(nlevels), int(edge_threshold), int(fast_threshold))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def bgr2gray(bgr):
    bgr, p = _u8(bgr)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(p, w, h, out.ctypes.data_as(C.c_void_p))
    return out


def levels(w, h, prm):
    n = prm.nlevels
    lw = (C.c_int * n)(); lh = (C.c_int * n)(); sc = (C.c_float * n)(); q = (C.c_int * n)()
    lib().orc_levels(w, h, C.byref(prm), lw, lh, sc, q)
    return list(lw), list(lh), [np.float32(s) for s in sc], list(q)


def pyramid_level(gray, prm, level, blurred=False):
    gray, p = _u8(gray)
    h, w = gray.shape
    lw, lh, _, _ = levels(w, h, prm)
    out = np.empty((lh[level], lw[level]), np.uint8)
    po = out.ctypes.data_as(C.c_void_p)
    lib().orc_pyramid_level(p, w, h, C.byref(prm), level, None if blurred else po, po if blurred else None)
    return out


def fast_level(img, threshold):
    img, p = _u8(img)
    h, w = img.shape
    cap = (w * h) // 4 + 16
    out = np.empty((cap, 3), np.int32)
    n = lib().orc_fast_level(p, w, h, int(threshold), out.ctypes.data_as(C.c_void_p), cap)
    return out[:n].copy()


def retain_best(resp, n_points):
    resp = np.ascontiguousarray(resp, np.float32)
    order = np.empty(len(resp), np.int32)
    n = lib().orc_retain_best(resp.ctypes.data_as(C.c_void_p), len(resp), int(n_points),
                              order.ctypes.data_as(C.c_void_p))
    return order[:n].copy()


def harris(img, x, y):
    img, p = _u8(img)
    h, w = img.shape
    return np.float32(lib().orc_harris(p, w, h, int(x), int(y)))


def ic_angle(img, x, y):
    img, p = _u8(img)
    h, w = img.shape
    return np.float32(lib().orc_ic_angle(p, w, h, int(x), int(y)))


def detect_and_compute(gray, prm, want_desc=True):
    """-> (structured keypoint array, (N,32) u8 descriptors or None)"""
    gray, p = _u8(gray)
    h, w = gray.shape
    cap = max(4 * prm.nfeatures + 1024, 4096)
    while True:
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = lib().orc_orb_detect_compute(p, w, h, C.byref(prm), kps.ctypes.data_as(C.c_void_p),
                                         desc.ctypes.data_as(C.c_void_p) if want_desc else None, cap)
        if n >= 0:
            break
        cap = -n
    return kps[:n].copy(), (desc[:n].copy() if want_desc and n else None)


def compute(gray, prm, kps_in):
    """kps_in: structured KP_DTYPE array -> (kept_idx, (N',32) descriptors)"""
    gray, p = _u8(gray)
    h, w = gray.shape
    kps_in = np.ascontiguousarray(kps_in, KP_DTYPE)
    n = len(kps_in)
    kept = np.empty(max(n, 1), np.int32)
    desc = np.zeros((max(n, 1), 32), np.uint8)
    m = lib().orc_orb_compute(p, w, h, C.byref(prm), kps_in.ctypes.data_as(C.c_void_p), n,
                              kept.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p))
    if m < 0:
        raise ValueError("negative octave")
    return kept[:m].copy(), desc[:m].copy()


def match_knn2(q, t):
    q, pq = _u8(q)
    t, pt = _u8(t)
    nq, nt = len(q), len(t)
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.int32)
    lib().orc_match_knn2(pq, nq, pt, nt, idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
    return idx, dist


def ratio_test(idx, dist, ratio, enabled=True):
    nq = len(idx)
    out = np.empty(nq, np.uint8)
    idx = np.ascontiguousarray(idx, np.int32); dist = np.ascontiguousarray(dist, np.int32)
    lib().orc_ratio_test(idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p), nq,
                         C.c_double(float(ratio)), int(bool(enabled)), out.ctypes.data_as(C.c_void_p))
    return out.astype(bool)


def min_eigen(gray):
    gray, p = _u8(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    lib().orc_min_eigen(p, w, h, out.ctypes.data_as(C.c_void_p))
    return out


def grid_good_features(gray, n_features=2000):
    """corner stage of ORBExtractor.distribute_keypoints -> (N,2) float32 (x, y), cell-major order"""
    gray, p = _u8(gray)
    h, w = gray.shape
    out = np.empty((64 * max(n_features // 64, 1), 2), np.float32)
    n = lib().orc_grid_good_features(p, w, h, int(n_features), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()
