"""FrameStream -- a frame iterator through the BATCHED mode (mo_stream, csrc/stream.hip): host frames in, per-frame results out.

The reference's driver calls the classes once per frame (src/tests/tester_map.py:57-75 -> Tracker.process_frame, tracker.py:73-146),
one launch + synchronisation round trip each.  A caller that can look ahead (a video file, a recorded sequence) feeds an iterator
instead: frames are gathered into chunks, every chunk is ONE mo_dev_frontend_batch call, the upload of the next chunk and the handling
of the previous chunk's results overlap the compute of the current one.  Every consecutive pair of the sequence is processed exactly
once (a chunk re-extracts the previous chunk's last frame), with the sampling stream of its GLOBAL pair index: the results equal those
of the per-frame loop (ORBExtractor.detect_and_compute + utils.track_from_last_frame(..., pair_index=i)) - keypoints, descriptors and
kept matches bit for bit, poses bit for bit.

    stream = FrameStream(K, chunk=64, n_features=2000)                 # MODE_TRACK on the ORB detector by default
    for r in stream.run(frames):                                       # frames: iterable of (H, W) / (H, W, 3) uint8 arrays
        r.index, r.keypoints (KP_DTYPE records), r.descriptors (n, 32)
        r.pair  -> None for frame 0, else dict(sel, sel_dist, inlier, R, t, n_inliers, ok) against frame index - 1
"""
import ctypes as C

import numpy as np

from . import (DETECT_GRID, DETECT_ORB, KP_DTYPE, MODE_INIT, MODE_TRACK, MO_ERR_CAPACITY, MO_OK, Context, NativeError, StreamParams,
               StreamResult, orb_params)


# (chunks in flight = mo_stream_lanes() of the library a stream was created on: FrameStream.lanes)


class _Chunk:
    """the arrays of one collected chunk (views of the lane's pinned result buffer) and what is needed to cut a frame's part out of them"""
    __slots__ = ("counts", "kps", "desc", "pose", "npts", "mask", "sel", "seld", "seln", "midx", "mdist", "mpass", "pts", "off", "prev_count",
                 "first_frame", "first_pair", "cap", "track", "copy", "stream", "serial", "ok")

    def check(self):
        """copy=False: the arrays are views of a lane's pinned buffer, which the second submit after this chunk's collect overwrites
        (and close() frees): reading them later fails loudly instead of returning another chunk's data"""
        if self.copy:
            return
        st = self.stream
        if st is None or st.h_stream is None or st._submitted - self.serial >= st.lanes:
            raise RuntimeError("FrameResult of a FrameStream(copy=False) read after its pinned buffer was reused or the stream closed: "
                               "read the arrays while iterating, or create the stream with copy=True")


class FrameResult:
    """One frame of the sequence.  .index; .keypoints (KP_DTYPE records), .descriptors (n, 32) uint8; .pair: None for frame 0, else the
    PairResult of the pair step against frame index - 1.  The arrays are cut out of the chunk when they are first asked for (with
    copy=True, the default, the chunk's arrays are the caller's own copies; with copy=False views of the pinned result buffer)."""
    __slots__ = ("index", "_c", "_f", "_kp", "_de", "_pr")

    def __init__(self, index, chunk, f):
        self.index, self._c, self._f = index, chunk, f
        self._kp = self._de = self._pr = None

    def _own(self, x):
        return x   # (copy=True: the chunk's arrays already are the caller's own copies, made in bulk at collect)

    @property
    def keypoints(self):
        if self._kp is None:
            self._c.check()
            self._kp = self._own(self._c.kps[self._f, :int(self._c.counts[self._f])])
        return self._kp

    @property
    def descriptors(self):
        if self._de is None:
            self._c.check()
            self._de = self._own(self._c.desc[self._f, :int(self._c.counts[self._f])])
        return self._de

    @property
    def pair(self):
        c = self._c
        j = self._f - c.off                            # pair row whose TRAIN frame is this frame
        if j < 0:
            return None
        if self._pr is None:
            c.check()
            self._pr = PairResult(c, j, self._f)
        return self._pr


class PairResult:
    """The pair step of one frame against its predecessor; fields are cut out of the chunk when asked for (attribute or ["key"] access).
    MODE_TRACK: ok (>= 8 kept matches and a model, tracker.py:234), R (3, 3), t (3, 1), n_inliers, pair_index, sel (m, 2) [queryIdx, trainIdx]
    in the reference's order, sel_dist (m,), inlier (m,) bool.  MODE_INIT: ok, R, t, n_inliers, pair_index, idx / dist (n_prev, 2), keep,
    pose_mask (n_prev,) bool[, X (n_prev, 3)]."""
    __slots__ = ("_c", "_j", "_f", "_inl")

    def __init__(self, chunk, j, f):
        self._c, self._j, self._f, self._inl = chunk, j, f, None

    def __getitem__(self, key):
        try:
            return getattr(self, key)
        except AttributeError:
            raise KeyError(key)

    def __contains__(self, key):
        return hasattr(self, key)

    @property
    def _nq(self):
        c = self._c
        return min(int(c.prev_count) if (self._j == 0 and c.off == 0) else int(c.counts[self._f - 1]), c.cap)

    ok = property(lambda self: bool(self._c.ok[self._j]))
    R = property(lambda self: self._c.pose[self._j, :9].reshape(3, 3))
    t = property(lambda self: self._c.pose[self._j, 9:].reshape(3, 1))
    n_inliers = property(lambda self: int(self._c.npts[self._j]) if self._c.ok[self._j] else 0)
    pair_index = property(lambda self: self._c.first_pair + self._j)

    @property
    def sel(self):
        c = self._c
        if not c.track:
            raise AttributeError("sel")
        return c.sel[self._j, :int(c.seln[self._j])]

    @property
    def sel_dist(self):
        c = self._c
        if not c.track:
            raise AttributeError("sel_dist")
        return c.seld[self._j, :int(c.seln[self._j])]

    @property
    def inlier(self):
        c = self._c
        if not c.track:
            raise AttributeError("inlier")
        if self._inl is None:
            s = self.sel
            self._inl = (c.mask[self._j][s[:, 0]] != 0) if c.ok[self._j] else np.zeros(len(s), bool)
        return self._inl

    @property
    def pose_mask(self):
        c = self._c
        if c.track:
            raise AttributeError("pose_mask")
        return c.mask[self._j, :self._nq].view(bool)

    def _knn(self, name):
        a = getattr(self._c, name)
        if a is None:
            raise AttributeError(name)
        return a[self._j, :self._nq]

    idx = property(lambda self: self._knn("midx"))
    dist = property(lambda self: self._knn("mdist"))
    keep = property(lambda self: self._knn("mpass").view(bool))
    X = property(lambda self: self._knn("pts"))


def _view(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n == 0:
        return np.zeros(shape, dtype)
    return np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape)


class FrameStream:
    def __init__(self, K, width=640, height=480, channels=1, chunk=64, n_features=2000, cap=None, detector=DETECT_ORB, mode=MODE_TRACK,
                 ratio=0.75, disp_frac=0.02, thr_px=None, n_hyp=4096, seed=4096, prm=None, device=0, want_matches=False, want_points=False,
                 copy=True):
        """copy=True: every yielded array is the caller's own (a copy out of the pinned result buffer); copy=False: views that stay valid
        until two more chunks have been submitted (the benchmark's rate without the per-frame copies)"""
        self.prm = prm if prm is not None else orb_params(nfeatures=n_features)
        self.cap = int(cap or ((self.prm.nfeatures + 48 + 63) // 64 * 64))   # (retainBest keeps ties: a few rows beyond nfeatures)
        self.chunk, self.w, self.h, self.ch, self.mode, self.copy = int(chunk), int(width), int(height), int(channels), int(mode), bool(copy)
        self.ctx = Context(device=device, max_w=self.w, max_h=self.h, max_batch=self.chunk + 1)
        sp = StreamParams()
        sp.w, sp.h, sp.ch, sp.chunk, sp.cap, sp.detector, sp.mode = self.w, self.h, self.ch, self.chunk, self.cap, int(detector), int(mode)
        sp.ratio = float(ratio if ratio is not None else -1.0); sp.disp_frac = float(disp_frac)
        sp.thr_px = float(thr_px if thr_px is not None else (1.0 if mode == MODE_TRACK else 3.0))
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        for i in range(9):
            sp.K[i] = Kc[i]
        sp.n_hyp, sp.seed, sp.pair_index_base = int(n_hyp), int(seed), 0
        sp.want_matches, sp.want_points = int(bool(want_matches) or mode == MODE_INIT), int(bool(want_points))
        self.sp = sp
        self.h_stream = self.ctx.lib.mo_stream_create(self.ctx.h, C.byref(self.prm), C.byref(sp))
        if not self.h_stream:
            raise NativeError(-1, self.ctx.lib.mo_last_error(self.ctx.h).decode())
        lib = self.ctx.lib
        self.lanes = int(lib.mo_stream_lanes()) if hasattr(lib, "mo_stream_lanes") else 3   # (else: an older build under VSLAM_AMD_LIB)
        self._in_flight = 0
        self._submitted = 0

    def close(self):
        if getattr(self, "h_stream", None):
            self.ctx.lib.mo_stream_destroy(self.h_stream)
            self.h_stream = None
            self.ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- chunk level -----------------------------------------------------------------------------------------------------
    def submit(self, block):
        """block: (n, H, W[, 3]) uint8, n <= chunk; returns without waiting for the GPU"""
        a = np.ascontiguousarray(block, np.uint8)
        if a.shape[1] != self.h or a.shape[2] != self.w or (a.ndim == 4) != (self.ch == 3):
            raise ValueError("frames of shape %s do not fit the stream (%d x %d x %d)" % (a.shape[1:], self.h, self.w, self.ch))
        rc = self.ctx.lib.mo_stream_submit(self.h_stream, a.ctypes.data_as(C.c_void_p), int(a.shape[0]), 0, 0)
        if rc != MO_OK:
            raise NativeError(rc, self.ctx.lib.mo_stream_last_error(self.h_stream).decode())
        self._in_flight += 1
        self._submitted += 1

    def collect(self):
        """results of the oldest chunk in flight -> list of FrameResult"""
        r = StreamResult()
        rc = self.ctx.lib.mo_stream_collect(self.h_stream, C.byref(r))
        if rc not in (MO_OK, MO_ERR_CAPACITY):
            raise NativeError(rc, self.ctx.lib.mo_stream_last_error(self.h_stream).decode())
        self._in_flight -= 1
        if rc == MO_ERR_CAPACITY:
            raise NativeError(rc, "capacity flag %d raised inside a streamed chunk (cap = %d rows per frame)" % (r.flags, r.cap))
        nf, npair, cap = r.n_frames, r.n_pairs, r.cap
        c = _Chunk()
        c.cap, c.copy, c.track = cap, self.copy, self.mode == MODE_TRACK
        c.counts = np.minimum(_view(r.counts, (nf,), np.int32), cap)
        c.kps = _view(r.kps, (nf, cap), KP_DTYPE)
        c.desc = _view(r.desc, (nf, cap, 32), np.uint8)
        c.pose = _view(r.pose, (npair, 12), np.float64)
        c.npts = _view(r.n_points, (npair,), np.int32)
        c.mask = _view(r.pose_mask, (npair, cap), np.uint8)
        c.sel = c.seld = c.seln = c.midx = c.mdist = c.mpass = c.pts = None
        if c.track:
            c.sel = _view(r.sel_idx, (npair, cap, 2), np.int32)
            c.seld = _view(r.sel_dist, (npair, cap), np.int32)
            c.seln = np.minimum(_view(r.sel_n, (npair,), np.int32), cap)
        if r.match_idx:
            c.midx = _view(r.match_idx, (npair, cap, 2), np.int32); c.mdist = _view(r.match_dist, (npair, cap, 2), np.int32)
            c.mpass = _view(r.match_pass, (npair, cap), np.uint8)
        if r.points:
            c.pts = _view(r.points, (npair, cap, 3), np.float32)
        if self.copy:   # the caller's own arrays: one bulk copy per array out of the pinned buffer (a FrameResult then slices these)
            for name in ("kps", "desc", "pose", "npts", "mask", "sel", "seld", "seln", "midx", "mdist", "mpass", "pts"):
                a = getattr(c, name)
                if a is not None:
                    setattr(c, name, a.copy())
        finite = np.isfinite(c.pose).all(axis=1)
        c.ok = (finite & (c.seln >= 8)) if c.track else finite   # tracker.py:234: fewer than 8 kept matches -> no pose
        c.stream, c.serial = self, self._submitted - self._in_flight   # (index of this chunk + 1 among the submitted ones)
        c.off = 1 if r.first_pair == r.first_frame else 0   # first chunk: frame row 0 has no pair in front of it
        c.prev_count, c.first_frame, c.first_pair = int(r.prev_count), int(r.first_frame), int(r.first_pair)
        first = c.first_frame
        return [FrameResult(first + f, c, f) for f in range(nf)]

    # ---- iterator level --------------------------------------------------------------------------------------------------
    def run(self, frames):
        """frames: an array (N, H, W[, 3]) uint8 - a frame stack: chunks are handed over as slices, no per-frame copy in Python - or any
        iterable of (H, W[, 3]) uint8 arrays (gathered into chunks frame by frame) -> generator of FrameResult in frame order.  Up to
        `self.lanes` chunks are in flight: while the GPU works on chunk i and uploads chunk i + 1, the caller consumes the results of chunk i - 1.
        With copy=False a FrameResult's arrays must be read before two more chunks have been submitted (i.e. while iterating)."""
        if isinstance(frames, np.ndarray) and frames.ndim == (4 if self.ch == 3 else 3):
            for k in range(0, len(frames), self.chunk):
                if self._in_flight == self.lanes:
                    yield from self.collect()
                self.submit(frames[k:k + self.chunk])
                if self._in_flight == self.lanes:        # (the oldest chunk is read while the two younger ones upload / compute)
                    yield from self.collect()
            while self._in_flight:
                yield from self.collect()
            return
        shape = (self.chunk, self.h, self.w) + ((3,) if self.ch == 3 else ())
        block = np.empty(shape, np.uint8)
        fill = 0
        for fr in frames:
            block[fill] = fr
            fill += 1
            if fill == self.chunk:
                if self._in_flight == self.lanes:
                    yield from self.collect()
                self.submit(block)   # (staged into pinned memory inside the call: the block is refilled at once)
                fill = 0
                if self._in_flight == self.lanes:
                    yield from self.collect()
        if fill:
            if self._in_flight == self.lanes:
                yield from self.collect()
            self.submit(block[:fill])
        while self._in_flight:
            yield from self.collect()
