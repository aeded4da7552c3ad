"""vslam_amd -- ctypes binding of libvslam_amd.so (include/vslam_amd.h), the MI355X-native ORB front-end.

This module is plumbing only: it loads the HIP library that sits next to this package, declares the C-ABI
signatures and turns status codes into exceptions.  There is NO CPU implementation behind it: without the
built library, or without a HIP device, every call fails loudly (NativeUnavailable).
"""
import ctypes as C
import os
import weakref

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# VSLAM_AMD_LIB: another build of the same library (A/B timing of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("VSLAM_AMD_LIB") or os.path.join(_PKG_ROOT, "libvslam_amd.so")

MO_OK, MO_ERR_ARG, MO_ERR_HIP, MO_ERR_CAPACITY, MO_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
ABI_VERSION = 5  # MO_ABI_VERSION of include/vslam_amd.h: the struct layouts mirrored below
ORDER_LIBSTDCXX, ORDER_MSVC = 0, 1

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class NativeUnavailable(RuntimeError):
    pass


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvslam_amd error %d: %s" % (code, msg))
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("edge_threshold", C.c_int32), ("first_level", C.c_int32), ("wta_k", C.c_int32),
                ("score_type", C.c_int32), ("patch_size", C.c_int32), ("fast_threshold", C.c_int32),
                ("select_order", C.c_int32)]


class BatchIO(C.Structure):
    _fields_ = [("d_gray", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("batch", C.c_int32), ("cap", C.c_int32),
                ("ratio", C.c_double), ("K", C.c_double * 9), ("thr_px", C.c_double), ("n_hyp", C.c_int32),
                ("seed", C.c_uint64),
                ("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("d_counts", C.c_void_p), ("d_match_idx", C.c_void_p),
                ("d_match_dist", C.c_void_p), ("d_match_pass", C.c_void_p), ("d_pose", C.c_void_p),
                ("d_points", C.c_void_p), ("d_n_points", C.c_void_p),
                ("mode", C.c_int32), ("disp_frac", C.c_double), ("d_sel_idx", C.c_void_p), ("d_sel_dist", C.c_void_p),
                ("d_sel_n", C.c_void_p), ("d_pose_mask", C.c_void_p), ("pair_index_base", C.c_uint64),
                ("detector", C.c_int32), ("d_grid_xy", C.c_void_p), ("d_grid_n", C.c_void_p), ("d_grid_kept", C.c_void_p),
                ("n_kf_pairs", C.c_int32), ("d_kf_query", C.c_void_p), ("d_kf_train", C.c_void_p), ("d_kf_P1", C.c_void_p),
                ("d_kf_P2", C.c_void_p), ("d_kf_F", C.c_void_p)]


class FrameRef(C.Structure):
    _fields_ = [("token", C.c_uint64), ("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32)]


class PairParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("ratio", C.c_double), ("disp_frac", C.c_double),
                ("K", C.c_double * 9), ("thr_px", C.c_double), ("n_hyp", C.c_int32), ("seed", C.c_uint64), ("pair_index", C.c_uint64)]


class PairOut(C.Structure):
    _fields_ = [("match_idx", C.c_void_p), ("match_dist", C.c_void_p), ("match_pass", C.c_void_p), ("sel_idx", C.c_void_p),
                ("sel_dist", C.c_void_p), ("inlier", C.c_void_p), ("ransac", C.c_void_p), ("X", C.c_void_p),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("E", C.c_double * 9), ("n_sel", C.c_int32), ("n_good", C.c_int32),
                ("n1", C.c_int32), ("n2", C.c_int32), ("token1", C.c_uint64), ("token2", C.c_uint64)]


class StreamParams(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("ch", C.c_int32), ("chunk", C.c_int32), ("cap", C.c_int32), ("detector", C.c_int32),
                ("mode", C.c_int32), ("ratio", C.c_double), ("disp_frac", C.c_double), ("K", C.c_double * 9), ("thr_px", C.c_double),
                ("n_hyp", C.c_int32), ("seed", C.c_uint64), ("pair_index_base", C.c_uint64), ("want_matches", C.c_int32),
                ("want_points", C.c_int32)]


class StreamResult(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_pairs", C.c_int32), ("first_frame", C.c_uint64), ("first_pair", C.c_uint64), ("cap", C.c_int32),
                ("flags", C.c_int32), ("prev_count", C.c_int32), ("counts", C.c_void_p), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("sel_idx", C.c_void_p), ("sel_dist", C.c_void_p), ("sel_n", C.c_void_p), ("pose", C.c_void_p), ("pose_mask", C.c_void_p),
                ("n_points", C.c_void_p), ("match_idx", C.c_void_p), ("match_dist", C.c_void_p), ("match_pass", C.c_void_p),
                ("points", C.c_void_p)]


MODE_INIT, MODE_TRACK, MODE_KEYFRAME = 0, 1, 2
DETECT_ORB, DETECT_GRID = 0, 1
TIMING_SLOTS = 64  # MO_TIMING_SLOTS of the library: event sets kept for Context.stage_times(back)


# every symbol include/vslam_amd.h declares: name -> (restype, argtypes)
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
SIGNATURES = {
    "mo_create": (_vp, [_i, _i, _i, _i]),
    "mo_destroy": (None, [_vp]),
    "mo_last_error": (C.c_char_p, [_vp]),
    "mo_set_stream": (_i, [_vp, _vp]),
    "mo_set_stream_null": (_i, [_vp]),
    "mo_sync": (_i, [_vp]),
    "mo_device_count": (_i, []),
    "mo_abi_version": (_i, []),
    "mo_orb_detect_compute": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mo_orb_compute": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "mo_orb_grid_good_features": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mo_orb_grid_detect_compute": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mo_dbg_min_eigen": (_i, [_vp, _vp, _i, _i, _vp]),
    "mo_dbg_set_poison": (_i, [_vp, _i]),
    "mo_undistort": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mo_dev_undistort": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mo_match_knn2_ratio": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "mo_init_two_view": (_i, [_vp, _vp, _vp, _i, _vp, _d, _d, _i, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mo_recover_pose": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mo_find_fundamental": (_i, [_vp, _vp, _vp, _i, _d, _d, _i, C.c_uint64, _vp, _vp, _vp]),
    "mo_track_pair": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _d, _d, _vp, _d, _i, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp]),
    "mo_triangulate_points": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mo_dev_frontend_batch": (_i, [_vp, _vp, _vp]),
    "mo_dev_orb_detect_compute": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mo_dev_match_pairs": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    "mo_dev_status": (_i, [_vp, _vp]),
    "mo_stream_create": (_vp, [_vp, _vp, _vp]),
    "mo_stream_destroy": (None, [_vp]),
    "mo_stream_submit": (_i, [_vp, _vp, _i, _i, C.c_size_t]),
    "mo_stream_collect": (_i, [_vp, _vp]),
    "mo_stream_last_error": (C.c_char_p, [_vp]),
    "mo_stream_lanes": (_i, []),
    "mo_comm_unique_id": (_i, [_vp]),
    "mo_comm_init": (_i, [_vp, _vp, _i, _i]),
    "mo_comm_destroy": (_i, [_vp]),
    "mo_gather_map_points": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mo_host_times": (_i, [_vp, _vp]),
    "mo_set_host_timing": (_i, [_vp, _i]),
    "mo_last_token": (_i, [_vp, _vp]),
    "mo_pair_frontend": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mo_stage_times": (_i, [_vp, _vp, _vp, _i]),
    "mo_stage_times_back": (_i, [_vp, _i, _vp, _vp, _i]),
    "mo_dbg_pyramid_level": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mo_dbg_fast_level": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "mo_dbg_retain_best": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
}

_lib = None


def _preload_shared_hip_runtime():
    """One process must use ONE HIP runtime.  A PyTorch-ROCm wheel bundles its own libamdhip64; if this library pulled
    in the system runtime first, a later `import torch` would find "No HIP GPUs".  So when torch is installed (the
    benchmark and the tests use it for device memory and torch.distributed) and not yet loaded, its runtime is loaded
    first - by path, without importing torch - and libvslam_amd.so binds to that same copy."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen libvslam_amd.so and declare all signatures (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeUnavailable(
            "%s is missing: build it with `make -C visual-slam_amd/csrc` (or __graft_entry__.build()). "
            "There is no CPU fallback." % LIB_PATH)
    _preload_shared_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        except AttributeError:
            if os.environ.get("VSLAM_AMD_LIB"):  # an older build under A/B timing (tools/ab.py) may predate a symbol
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if not os.environ.get("VSLAM_AMD_LIB") and lib.mo_abi_version() != ABI_VERSION:
        raise NativeUnavailable("libvslam_amd.so has ABI version %d, this binding was written for %d: rebuild (make -C visual-slam_amd/csrc)"
                                % (lib.mo_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def device_count():
    return int(load_library().mo_device_count())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# Resident single-frame results: descriptor array (by identity) -> (token, context, the keypoint record array it came with).  A frame's
# keypoints and descriptors stay on the device after detect_and_compute (the library keeps the last four); when the SAME arrays come
# back for a pair step (matcher.match, track_from_last_frame, MapInitializer.initialize) nothing is uploaded.  The arrays are handed
# out read-only so that the device copy cannot go stale behind an in-place write (a copy of them is an ordinary, writable array).
_resident = {}


def _bind_resident(ctx, token, kps_arr, desc):
    if not token or desc is None:
        return
    desc.flags.writeable = False
    kps_arr.flags.writeable = False
    key = id(desc)
    _resident[key] = (token, weakref.ref(ctx), weakref.ref(desc, lambda _, key=key: _resident.pop(key, None)), weakref.ref(kps_arr))


def resident_token(ctx, desc, kps_arr=None):
    """token under which (kps_arr, desc) are resident in ctx, 0 if they are not these very arrays (or were never resident)"""
    e = _resident.get(id(desc))
    if e is None or e[1]() is not ctx or e[2]() is not desc:
        return 0
    if kps_arr is not None and e[3]() is not kps_arr:
        return 0
    return e[0]


def _resident_kps(desc):
    e = _resident.get(id(desc))
    return e[3]() if e is not None and e[2]() is desc else None


def orb_params(nfeatures=2000, scale_factor=1.2, nlevels=8, edge_threshold=31, fast_threshold=7,
               select_order=ORDER_LIBSTDCXX):
    return OrbParams(int(nfeatures), float(scale_factor), int(nlevels), int(edge_threshold), 0, 2, 0, 31,
                     int(fast_threshold), int(select_order))


class Context:
    """Owns one mo_ctx (one GPU, one stream).  Thread-compatible, like the reference's single-threaded use."""

    def __init__(self, device=0, max_w=2048, max_h=2048, max_batch=1):
        self.lib = load_library()
        if self.lib.mo_device_count() <= 0:
            raise NativeUnavailable("no HIP device visible: the MI355X front-end has no CPU fallback")
        self.h = self.lib.mo_create(int(device), int(max_w), int(max_h), int(max_batch))
        if not self.h:
            raise NativeUnavailable(self.lib.mo_last_error(None).decode())
        self.device = device
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "h", None):
            self.lib.mo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != MO_OK:
            raise NativeError(rc, self.lib.mo_last_error(self.h).decode())

    def set_stream(self, stream_handle):
        """None: the context's own stream; 0: the HIP null stream (torch.cuda.current_stream().cuda_stream of torch's default
        stream) - the library's launches are then ordered with the work torch has on it; otherwise a hipStream_t handle."""
        if stream_handle is None:
            self._check(self.lib.mo_set_stream(self.h, None))
        elif int(stream_handle) == 0:
            self._check(self.lib.mo_set_stream_null(self.h))
        else:
            self._check(self.lib.mo_set_stream(self.h, C.c_void_p(int(stream_handle))))

    def sync(self):
        self._check(self.lib.mo_sync(self.h))

    def dev_status(self):
        """flag word of the mo_dev_* calls since the last query (0 = nothing overflowed); synchronises the stream"""
        f = (C.c_int32 * 4)()
        rc = self.lib.mo_dev_status(self.h, f)
        if rc not in (MO_OK, MO_ERR_CAPACITY):
            self._check(rc)
        return int(f[0])

    # ---- multi-GPU gather through the C-ABI (RCCL); the id travels between ranks by the caller's own channel ------------
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.mo_comm_unique_id(buf)
        if rc != MO_OK:
            raise NativeUnavailable("RCCL could not be loaded (mo_comm_unique_id returned %d)" % rc)
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.mo_comm_init(self.h, buf, int(rank), int(world)))

    def gather_map_points(self, d_local_ptr, rows_local, rows_max, cap, root, d_all_ptr, d_rows_all_ptr):
        self._check(self.lib.mo_gather_map_points(self.h, C.c_void_p(d_local_ptr), int(rows_local), int(rows_max), int(cap), int(root),
                                                  C.c_void_p(d_all_ptr) if d_all_ptr else None, C.c_void_p(d_rows_all_ptr)))

    def stage_times(self, back=0):
        """(name, ms) per stage of the call `back` calls ago (0 = the last; the library keeps TIMING_SLOTS event sets)"""
        names = C.POINTER(C.c_char_p)()
        ms = (C.c_float * 32)()
        n = self.lib.mo_stage_times_back(self.h, back, C.byref(names), ms, 32)
        if n < 0:
            self._check(n)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def set_host_timing(self, on):
        """stage events inside the single-call host entry points (off by default: they idle the GPU between kernels)"""
        self._check(self.lib.mo_set_host_timing(self.h, int(bool(on))))

    def host_times(self):
        """host-side clock of the last single-call host entry point: dict(enqueue_us, wait_us, unpack_us, total_us)"""
        us = (C.c_double * 4)()
        self._check(self.lib.mo_host_times(self.h, us))
        return dict(enqueue_us=us[0], wait_us=us[1], unpack_us=us[2], total_us=us[3])

    # ---- host API ---------------------------------------------------------------------------------------
    def orb_detect_compute(self, images, prm, want_desc=True, cap=None):
        """images: (H,W) / (H,W,3) / (B,H,W) / (B,H,W,3) uint8 -> list of (kps structured array, desc or None)"""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim == 2:
            a = a[None]
        elif a.ndim == 3 and a.shape[-1] == 3 and images.ndim == 3:
            a = a[None]
        ch = 3 if a.ndim == 4 else 1
        b, h, w = a.shape[0], a.shape[1], a.shape[2]
        cap = int(cap or (prm.nfeatures + 1024))
        while True:
            kps = np.empty((b, cap), KP_DTYPE)   # (only the rows the library fills are handed out)
            desc = np.empty((b, cap, 32), np.uint8) if want_desc else None
            counts = np.zeros(b, np.int32)
            rc = self.lib.mo_orb_detect_compute(self.h, C.byref(prm), _ptr(a), w, h, w * ch, ch, b, _ptr(kps),
                                                _ptr(desc), cap, _ptr(counts))
            if rc == MO_ERR_CAPACITY and counts.max() > cap:
                cap = int(counts.max())
                continue
            self._check(rc)
            break
        self.last_token = 0
        if b == 1 and want_desc:  # the result stays resident on the device under this token (pair_frontend takes it back)
            tok = C.c_uint64(0)
            self.lib.mo_last_token(self.h, C.byref(tok))
            self.last_token = int(tok.value)
        out = []
        for f in range(b):
            n = int(counts[f])
            # views of this call's own buffers (fresh per call, nobody else holds them): no second copy of 120 KB per frame
            out.append((kps[f, :n], desc[f, :n] if want_desc and n else None))
        if self.last_token:
            _bind_resident(self, self.last_token, out[0][0], out[0][1])
        return out

    def orb_compute(self, image, prm, kps_in):
        a = np.ascontiguousarray(image, dtype=np.uint8)
        ch = 3 if a.ndim == 3 else 1
        h, w = a.shape[0], a.shape[1]
        k = np.ascontiguousarray(kps_in, KP_DTYPE)
        n = len(k)
        kept = np.zeros(max(n, 1), np.int32)
        desc = np.zeros((max(n, 1), 32), np.uint8)
        n_out = C.c_int(0)
        self._check(self.lib.mo_orb_compute(self.h, C.byref(prm), _ptr(a), w, h, w * ch, ch, _ptr(k), n, _ptr(kept),
                                            _ptr(desc), C.byref(n_out)))
        return kept[:n_out.value].copy(), desc[:n_out.value].copy()

    def grid_good_features(self, image, n_features):
        """corner stage of ORBExtractor.distribute_keypoints -> (N,2) float32 (x, y), cell-major order"""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        ch = 3 if a.ndim == 3 else 1
        h, w = a.shape[0], a.shape[1]
        xy = np.zeros((64 * max(int(n_features) // 64, 1), 2), np.float32)
        n = C.c_int(0)
        self._check(self.lib.mo_orb_grid_good_features(self.h, _ptr(a), w, h, w * ch, ch, int(n_features), _ptr(xy), C.byref(n)))
        return xy[:n.value].copy()

    def grid_detect_compute(self, image, prm, n_features, records=False):
        """ORBExtractor.distribute_keypoints in one device call -> (xy (N,2) float32 of ALL corners, kept (M,) int32 indices into xy
        of the corners orb.compute keeps, desc (M,32) uint8).  records=True: a fourth value, the KeyPoint(x, y, 31) records of the kept
        corners (KP_DTYPE, aligned with desc) - they and desc stay resident on the device like a detect_and_compute result (pair steps
        on these very arrays upload nothing)."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        ch = 3 if a.ndim == 3 else 1
        h, w = a.shape[0], a.shape[1]
        slots = 64 * max(int(n_features) // 64, 1)
        xy = np.empty((slots, 2), np.float32)
        kept = np.empty(slots, np.int32)
        desc = np.empty((slots, 32), np.uint8)
        n, nk = C.c_int(0), C.c_int(0)
        self._check(self.lib.mo_orb_grid_detect_compute(self.h, C.byref(prm), _ptr(a), w, h, w * ch, ch, int(n_features), _ptr(xy),
                                                        C.byref(n), _ptr(kept), _ptr(desc), C.byref(nk)))
        xy, kept, desc = xy[:n.value], kept[:nk.value], desc[:nk.value]
        if not records:
            return xy, kept, desc
        rec = np.zeros(nk.value, KP_DTYPE)
        rec["x"], rec["y"] = xy[kept, 0], xy[kept, 1]
        rec["size"], rec["angle"], rec["class_id"] = 31.0, -1.0, -1
        tok = C.c_uint64(0)
        self.lib.mo_last_token(self.h, C.byref(tok))
        self.last_token = int(tok.value)
        if self.last_token and nk.value:
            _bind_resident(self, self.last_token, rec, desc)
        return xy, kept, desc, rec

    def undistort(self, image, K, dist):
        """cv2.undistort(image, K, dist): (H, W) or (H, W, 3) uint8 -> same shape"""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        ch = 3 if a.ndim == 3 else 1
        h, w = a.shape[0], a.shape[1]
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        d5 = np.zeros(5); dd = np.asarray(dist, np.float64).ravel(); d5[:min(5, len(dd))] = dd[:5]
        out = np.empty_like(a)
        self._check(self.lib.mo_undistort(self.h, _ptr(a), w, h, w * ch, ch, _ptr(Kc), _ptr(d5), _ptr(out)))
        return out

    def dbg_min_eigen(self, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        out = np.zeros((h, w), np.float32)
        self._check(self.lib.mo_dbg_min_eigen(self.h, _ptr(g), w, h, _ptr(out)))
        return out

    def match_knn2_ratio(self, q, t, ratio=None):
        """q (nq,32) or (B,nq,32), t likewise -> idx (..,nq,2) i32, dist (..,nq,2) i32, pass (..,nq) bool"""
        if getattr(q, "ndim", 0) == 2 and _resident:
            tq, tt = resident_token(self, q), resident_token(self, t)
            if tq and tt:  # both frames are still on the device: the matcher alone through the pair call, nothing uploaded
                r = self.pair_frontend(_resident_kps(q), q, _resident_kps(t), t, MODE_INIT, np.eye(3), ratio=ratio, n_hyp=0, token1=tq, token2=tt)
                return r["idx"], r["dist"], r["keep"]
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        single = q.ndim == 2
        if single:
            q, t = q[None], t[None]
        b, nq, nt = q.shape[0], q.shape[1], t.shape[1]
        idx = np.full((b, nq, 2), -1, np.int32)
        dist = np.full((b, nq, 2), np.iinfo(np.int32).max, np.int32)
        ps = np.zeros((b, nq), np.uint8)
        r = C.c_double(float(ratio)) if ratio is not None else None
        self._check(self.lib.mo_match_knn2_ratio(self.h, _ptr(q), nq, _ptr(t), nt, C.byref(r) if r is not None else None,
                                                 b, _ptr(idx), _ptr(dist), _ptr(ps)))
        if single:
            return idx[0], dist[0], ps[0].astype(bool)
        return idx, dist, ps.astype(bool)

    def init_two_view(self, p1, p2, K, thr_px=3.0, prob=0.999, n_hyp=4096, seed=4096):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        m = len(p1)
        R = np.zeros(9); t = np.zeros(3); E = np.zeros(9)
        ran = np.zeros(max(m, 1), np.uint8); inl = np.zeros(max(m, 1), np.uint8)
        X = np.zeros((max(m, 1), 3), np.float32)
        ng = C.c_int(0)
        self._check(self.lib.mo_init_two_view(self.h, _ptr(p1), _ptr(p2), m, _ptr(Kc), float(thr_px), float(prob),
                                              int(n_hyp), C.c_uint64(int(seed)), _ptr(R), _ptr(t), _ptr(E), _ptr(ran),
                                              _ptr(inl), _ptr(X), C.byref(ng)))
        return dict(R=R.reshape(3, 3), t=t.reshape(3, 1), E=E.reshape(3, 3), ransac_mask=ran[:m].astype(bool),
                    pose_mask=inl[:m].astype(bool), X=X[:m], n_good=ng.value)

    def recover_pose(self, E, p1, p2, K, mask=None):
        """cv2.recoverPose(E, p1, p2, K, mask) for any E -> dict(n_good, R, t, mask (m,) bool, X (m, 3) float32)"""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        Ec = np.ascontiguousarray(E, np.float64).reshape(9); Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        m = len(p1)
        mi = None if mask is None else np.ascontiguousarray(np.asarray(mask).ravel() != 0, np.uint8)
        R = np.zeros(9); t = np.zeros(3); mo = np.zeros(max(m, 1), np.uint8); X = np.zeros((max(m, 1), 3), np.float32); ng = C.c_int(0)
        self._check(self.lib.mo_recover_pose(self.h, _ptr(Ec), _ptr(p1), _ptr(p2), m, _ptr(Kc), _ptr(mi), _ptr(R), _ptr(t), _ptr(mo),
                                             _ptr(X), C.byref(ng)))
        return dict(n_good=ng.value, R=R.reshape(3, 3), t=t.reshape(3, 1), mask=mo[:m].astype(bool), X=X[:m])

    def find_fundamental(self, p1, p2, thr_px=3.0, prob=0.99, n_hyp=4096, seed=4096):
        """cv2.findFundamentalMat(p1, p2, FM_RANSAC, thr_px, prob) -> (F 3x3 float64 or None, mask (m,) bool)"""
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        m = len(p1)
        F = np.zeros(9); mask = np.zeros(max(m, 1), np.uint8); ni = C.c_int(0)
        self._check(self.lib.mo_find_fundamental(self.h, _ptr(p1), _ptr(p2), m, float(thr_px), float(prob), int(n_hyp),
                                                 C.c_uint64(int(seed)), _ptr(F), _ptr(mask), C.byref(ni)))
        if not np.isfinite(F).all():
            return None, np.zeros(m, bool)
        return F.reshape(3, 3), mask[:m].astype(bool)

    def pair_frontend(self, kps1, desc1, kps2, desc2, mode, K, width=0, height=0, ratio=0.75, disp_frac=0.02, thr_px=None, n_hyp=4096,
                      seed=4096, pair_index=0, token1=0, token2=0, want_matches=None):
        """matcher -> (tracking filters) -> two-view stage on one frame pair in ONE device call (mo_pair_frontend).  kps*: structured
        KP_DTYPE arrays, desc*: (N, 32) uint8; token*: tokens of resident single-frame results (0: upload the arrays).
        MODE_INIT  -> dict(idx (n1, 2), dist (n1, 2), keep (n1,) bool, R, t, E, ransac_mask (n1,), pose_mask (n1,), X (n1, 3), n_good)
        MODE_TRACK -> dict(sel (n, 2) [queryIdx, trainIdx] in the reference's order, sel_dist, inlier (n,) bool, R, t, E, n_inliers)
        both carry token1 / token2: the names under which the two frames are resident now"""
        token1 = token1 or resident_token(self, desc1, kps1)   # (identity of the arrays detect_and_compute handed out)
        token2 = token2 or resident_token(self, desc2, kps2)
        k1 = np.ascontiguousarray(kps1, KP_DTYPE).reshape(-1); k2 = np.ascontiguousarray(kps2, KP_DTYPE).reshape(-1)
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        n1, n2 = len(k1), len(k2)
        if len(d1) != n1 or len(d2) != n2:
            raise ValueError("keypoints and descriptor rows differ in number (%d / %d, %d / %d)" % (n1, len(d1), n2, len(d2)))
        f1 = FrameRef(int(token1), _ptr(k1), _ptr(d1), n1); f2 = FrameRef(int(token2), _ptr(k2), _ptr(d2), n2)
        pp = PairParams()
        pp.mode = int(mode); pp.w = int(width); pp.h = int(height); pp.ratio = float(ratio if ratio is not None else -1.0)
        pp.disp_frac = float(disp_frac); pp.thr_px = float(thr_px if thr_px is not None else (1.0 if mode == MODE_TRACK else 3.0))
        pp.n_hyp = int(n_hyp); pp.seed = int(seed); pp.pair_index = int(pair_index)
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        for i in range(9):
            pp.K[i] = Kc[i]
        o = PairOut()
        m1 = max(n1, 1)
        init = mode == MODE_INIT
        if want_matches is None:
            want_matches = init
        if want_matches:
            idx = np.empty((m1, 2), np.int32); dist = np.empty((m1, 2), np.int32); keep = np.empty(m1, np.uint8)
            o.match_idx, o.match_dist, o.match_pass = _ptr(idx), _ptr(dist), _ptr(keep)
        inl = np.zeros(m1, np.uint8)
        o.inlier = _ptr(inl)
        if init:
            ran = np.zeros(m1, np.uint8); X = np.empty((m1, 3), np.float32)
            o.ransac, o.X = _ptr(ran), _ptr(X)
        else:
            sel = np.empty((m1, 2), np.int32); sd = np.empty(m1, np.int32)
            o.sel_idx, o.sel_dist = _ptr(sel), _ptr(sd)
        self._check(self.lib.mo_pair_frontend(self.h, C.byref(f1), C.byref(f2), C.byref(pp), C.byref(o)))
        r = dict(R=np.array(o.R[:]).reshape(3, 3), t=np.array(o.t[:]).reshape(3, 1), E=np.array(o.E[:]).reshape(3, 3),
                 token1=int(o.token1), token2=int(o.token2))
        if want_matches:
            r.update(idx=idx[:n1], dist=dist[:n1], keep=keep[:n1].view(bool))
        if init:
            r.update(ransac_mask=ran[:n1].view(bool), pose_mask=inl[:n1].view(bool), X=X[:n1], n_good=int(o.n_good))
        else:
            n = int(o.n_sel)
            r.update(sel=sel[:n], sel_dist=sd[:n], inlier=inl[:n].view(bool), n_inliers=int(o.n_good))
        return r

    def track_pair(self, kps1, desc1, kps2, desc2, width, height, K, ratio=0.75, disp_frac=0.02, thr_px=1.0, n_hyp=4096,
                   seed=4096, pair_index=0, token1=0, token2=0):
        """One tracking step (reference tracker.py:214-254) on the device: match -> displacement filter -> 2 x median distance
        filter -> essential matrix at thr_px -> pose.  kps*: structured KP_DTYPE arrays, desc*: (N, 32) uint8.
        -> dict(sel (n, 2) int32 [queryIdx, trainIdx] in the reference's order, sel_dist, inlier (n,) bool, R, t, E, n_inliers)"""
        return self.pair_frontend(kps1, desc1, kps2, desc2, MODE_TRACK, K, width, height, ratio, disp_frac, thr_px, n_hyp, seed, pair_index,
                                  token1, token2, want_matches=False)

    def triangulate_points(self, P1, P2, p1, p2):
        P1 = np.ascontiguousarray(P1, np.float64).reshape(12)
        P2 = np.ascontiguousarray(P2, np.float64).reshape(12)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        X4 = np.zeros((max(n, 1), 4), np.float32)
        self._check(self.lib.mo_triangulate_points(self.h, _ptr(P1), _ptr(P2), _ptr(p1), _ptr(p2), n, _ptr(X4)))
        return X4[:n]

    # ---- probes --------------------------------------------------------------------------------------------
    def dbg_pyramid_level(self, gray, prm, level, blurred=False):
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        out = np.zeros(h * w, np.uint8)
        lw, lh = C.c_int(0), C.c_int(0)
        self._check(self.lib.mo_dbg_pyramid_level(self.h, C.byref(prm), _ptr(g), w, h, level, int(blurred), _ptr(out),
                                                  C.byref(lw), C.byref(lh)))
        return out[:lw.value * lh.value].reshape(lh.value, lw.value).copy()

    def dbg_fast_level(self, gray, prm, level):
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        cap = w * h // 4 + 16
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        self._check(self.lib.mo_dbg_fast_level(self.h, C.byref(prm), _ptr(g), w, h, level, _ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def dbg_retain_best(self, resp, n_points, order):
        r = np.ascontiguousarray(resp, np.float32)
        out = np.zeros(max(len(r), 1), np.int32)
        n = C.c_int(0)
        self._check(self.lib.mo_dbg_retain_best(self.h, _ptr(r), len(r), int(n_points), int(order), _ptr(out), C.byref(n)))
        return out[:n.value].copy()


_default_ctx = None


def default_context():
    """Process-wide context used by the drop-in orbslam2 classes (device from VSLAM_AMD_DEVICE / LOCAL_RANK)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("VSLAM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _default_ctx = Context(device=dev, max_w=4095, max_h=4095, max_batch=1)
    return _default_ctx
