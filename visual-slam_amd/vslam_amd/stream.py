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


class FrameResult:
    __slots__ = ("index", "keypoints", "descriptors", "pair")

    def __init__(self, index, keypoints, descriptors, pair):
        self.index, self.keypoints, self.descriptors, self.pair = index, keypoints, descriptors, pair


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
        self.cap = int(cap or ((self.prm.nfeatures + 63) // 64 * 64 + 48))
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
        self._in_flight = 0
        self._prev = None  # (keypoint records, count) of the frame in front of the next chunk

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
        counts = np.minimum(_view(r.counts, (nf,), np.int32), cap)
        kps = _view(r.kps, (nf, cap), KP_DTYPE)
        desc = _view(r.desc, (nf, cap, 32), np.uint8)
        pose = _view(r.pose, (npair, 12), np.float64)
        npts = _view(r.n_points, (npair,), np.int32)
        mask = _view(r.pose_mask, (npair, cap), np.uint8)
        track = self.mode == MODE_TRACK
        if track:
            sel = _view(r.sel_idx, (npair, cap, 2), np.int32)
            seld = _view(r.sel_dist, (npair, cap), np.int32)
            seln = np.minimum(_view(r.sel_n, (npair,), np.int32), cap)
        if r.match_idx:
            midx = _view(r.match_idx, (npair, cap, 2), np.int32); mdist = _view(r.match_dist, (npair, cap, 2), np.int32)
            mpass = _view(r.match_pass, (npair, cap), np.uint8)
        pts = _view(r.points, (npair, cap, 3), np.float32) if r.points else None
        own = (lambda x: x.copy()) if self.copy else (lambda x: x)
        off = 1 if r.first_pair == r.first_frame else 0   # first chunk: frame row 0 has no pair in front of it
        out = []
        for f in range(nf):
            n = int(counts[f])
            pr = None
            j = f - off                                    # pair row whose TRAIN frame is this frame
            if j >= 0:
                nq = int(r.prev_count) if (j == 0 and off == 0) else int(counts[f - 1])
                nq = min(nq, cap)
                P = pose[j]
                pr = dict(R=P[:9].reshape(3, 3).copy(), t=P[9:].reshape(3, 1).copy(), n_inliers=int(npts[j]), pair_index=int(r.first_pair) + j)
                if track:
                    m = int(seln[j])
                    s = sel[j, :m]
                    ok = m >= 8 and bool(np.isfinite(P).all())      # tracker.py:234
                    pr.update(sel=own(s), sel_dist=own(seld[j, :m]), inlier=(mask[j][s[:, 0]] != 0) if ok else np.zeros(m, bool), ok=ok)
                    if not ok:
                        pr["n_inliers"] = 0
                else:
                    pr.update(pose_mask=own(mask[j, :nq]).view(bool), ok=bool(np.isfinite(P).all()))
                if r.match_idx:
                    pr.update(idx=own(midx[j, :nq]), dist=own(mdist[j, :nq]), keep=own(mpass[j, :nq]).view(bool))
                if pts is not None:
                    pr["X"] = own(pts[j, :nq])
            out.append(FrameResult(int(r.first_frame) + f, own(kps[f, :n]), own(desc[f, :n]), pr))
        return out

    # ---- iterator level --------------------------------------------------------------------------------------------------
    def run(self, frames):
        """frames: iterable of uint8 arrays (H, W) or (H, W, 3) -> generator of FrameResult in frame order.  Two chunks are kept in
        flight: while the GPU works on chunk i and uploads chunk i + 1, the caller consumes the results of chunk i - 1."""
        shape = (self.chunk, self.h, self.w) + ((3,) if self.ch == 3 else ())
        blocks = [np.empty(shape, np.uint8), np.empty(shape, np.uint8), np.empty(shape, np.uint8)]
        k, fill = 0, 0
        for fr in frames:
            blocks[k % 3][fill] = fr
            fill += 1
            if fill == self.chunk:
                if self._in_flight == 2:
                    yield from self.collect()
                self.submit(blocks[k % 3])   # (staged into pinned memory inside the call: the block may be refilled at once)
                k += 1
                fill = 0
        if fill:
            if self._in_flight == 2:
                yield from self.collect()
            self.submit(blocks[k % 3][:fill])
        while self._in_flight:
            yield from self.collect()
