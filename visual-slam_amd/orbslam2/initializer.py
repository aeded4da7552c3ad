"""MapInitializer -- drop-in for the reference class of the same name (src/orbslam2/initializer.py:12-181).

Public surface kept: MapInitializer(camera_matrix, min_matches=10, min_inliers_ratio=0.9), set_first_frame(kp, des, image),
initialize(kp, des, matcher, image) -> (success, R, t, map_points, matches), draw_initialization(img1, img2, matches) and the
attributes initialization_done / first_frame_* / current_frame_keypoints that Tracker reads (tracker.py:43,106,162,168-170).

What initialize() computes, stage by stage, and where it runs - ONE device call (mo_pair_frontend, MO_MODE_INIT) with the drop-in
matcher, two (matcher.match, then mo_init_two_view) with any other matcher object:
  1. matcher.match(first descriptors, current descriptors)            -> HIP k_match           (initializer.py:67)
  2. essential matrix, RANSAC threshold 3 px + relative pose           -> HIP two-view kernels  (initializer.py:79-83)
  3. P1 = K[I|0], P2 = K[R|t]; triangulation of the pose-mask matches  -> the same launch       (initializer.py:86-95)
  4. keep points in front of both cameras, sample colours, build the map-point dictionaries (host bookkeeping)
Return conventions follow the reference: (False, None, None, None, None) without a first frame, (False, None, None, None,
matches) when matching or geometry is insufficient, R 3x3 / t 3x1 float64 on success.
"""
import numpy as np

from . import utils as _geom
from .types import points_of


def _pixels(keypoints, indices):
    """(N, 2) float32 pixel coordinates of the selected keypoints."""
    return points_of(keypoints, indices)


class MapInitializer:
    def __init__(self, camera_matrix, min_matches=10, min_inliers_ratio=0.9):
        self.camera_matrix = camera_matrix
        self.min_matches = min_matches
        self.min_inliers_ratio = min_inliers_ratio  # kept for API parity; the reference never reads it either
        self.initialization_done = False
        self.first_frame_keypoints = None
        self.first_frame_descriptors = None
        self.first_frame_image = None
        self.current_frame_keypoints = None

    # ------------------------------------------------------------------ state
    def set_first_frame(self, keypoints, descriptors, image):
        self.first_frame_keypoints, self.first_frame_descriptors = keypoints, descriptors
        self.first_frame_image = np.array(image, copy=True)  # the caller reuses its frame buffer
        self.initialization_done = False

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _failure(matches):
        return False, None, None, None, matches

    def _colours(self, keypoints, query_indices):
        """Colour of the first frame under each reference keypoint (BGR triple; grey replicated; red when outside), initializer.py:128-138;
        the integer pixel is Python's int(kp.pt): truncation towards zero"""
        img = self.first_frame_image
        h, w = img.shape[:2]
        xy = np.trunc(points_of(keypoints, query_indices).astype(np.float64)).astype(np.int64)
        px, py = xy[:, 0], xy[:, 1]
        inside = (px >= 0) & (px < w) & (py >= 0) & (py < h)
        if inside.all():   # (always, for keypoints the extractor produced: they keep 31 px from the border) one gather instead of a loop
            rows = img[py, px] if img.ndim == 3 else np.repeat(img[py, px][:, None], 3, axis=1)
            return list(rows)
        colours = []
        for x, y, ok in zip(px.tolist(), py.tolist(), inside.tolist()):
            if not ok:
                colours.append(np.array([0, 0, 255]))
            elif img.ndim == 3:
                colours.append(img[y, x, :])
            else:
                colours.append(np.array([img[y, x]] * 3))
        return colours

    def _device_pair(self, ref_kps, ref_desc, cur_kps, cur_desc, matcher):
        """matcher.match + essential matrix + recoverPose + triangulation in ONE device call (mo_pair_frontend, MO_MODE_INIT) when the
        matcher is the drop-in brute-force matcher and both frames are record arrays; None otherwise (custom matcher objects, keypoint
        objects a caller wrote to: the per-stage path below)."""
        from .matcher import DescriptorMatcher, _NativeBFMatcher
        from .types import KeyPointSeq
        if type(matcher) is not DescriptorMatcher or not isinstance(matcher.matcher, _NativeBFMatcher):
            return None
        if not (isinstance(ref_kps, KeyPointSeq) and ref_kps.pristine and isinstance(cur_kps, KeyPointSeq) and cur_kps.pristine):
            return None
        if ref_desc.dtype != np.uint8 or cur_desc.dtype != np.uint8 or len(ref_desc) != len(ref_kps) or len(cur_desc) != len(cur_kps):
            return None
        import vslam_amd
        return vslam_amd.default_context().pair_frontend(ref_kps.array, ref_desc, cur_kps.array, cur_desc, vslam_amd.MODE_INIT,
                                                         self.camera_matrix, ratio=float(matcher.ratio_threshold), thr_px=3.0,
                                                         n_hyp=_geom.N_HYPOTHESES, seed=_geom.SEED)

    # ------------------------------------------------------------------ the hot path
    def initialize(self, current_keypoints, current_descriptors, matcher, current_image):
        ref_kps, ref_desc = self.first_frame_keypoints, self.first_frame_descriptors
        if ref_kps is None or ref_desc is None:
            return self._failure(None)
        from .types import dmatches_from_arrays
        fused = None
        if current_descriptors is not None and len(ref_desc) and len(current_descriptors):
            fused = self._device_pair(ref_kps, ref_desc, current_keypoints, current_descriptors, matcher)
        if fused is not None:
            # one call, one synchronisation: knn + ratio list, E at 3 px, pose, pose mask and map points by query keypoint
            q = np.flatnonzero(fused["keep"])
            tr, dist = fused["idx"][q, 0], fused["dist"][q, 0]
            if len(q) < self.min_matches:
                print(f"Not enough matches for initialization: {len(q)} < {self.min_matches}")
                return self._failure(dmatches_from_arrays(q, tr, dist))
            if not np.isfinite(fused["E"]).all():
                return self._failure(dmatches_from_arrays(q, tr, dist))
            R, t = fused["R"], fused["t"]
            ok = fused["pose_mask"][q]
            q, tr, dist = q[ok], tr[ok], dist[ok]
            if not len(q):
                return self._failure([])
            X = fused["X"][q]
        else:
            putative = matcher.match(ref_desc, current_descriptors)
            if len(putative) < self.min_matches:
                print(f"Not enough matches for initialization: {len(putative)} < {self.min_matches}")
                return self._failure(putative)
            K = self.camera_matrix
            from .types import match_arrays
            q, tr, dist = match_arrays(putative)   # (the arrays of an untouched match list, else read off the DMatch objects)
            q, tr = q.astype(np.intp), tr.astype(np.intp)
            xy_ref = _pixels(ref_kps, q)
            xy_cur = _pixels(current_keypoints, tr)
            # essential matrix at 3 px, recoverPose and the triangulation of the pose-mask survivors with P = K [I | 0], K [R | t]
            # (initializer.py:79-95) are ONE native call: it returns what the three cv2 calls of the reference return
            import vslam_amd
            g = vslam_amd.default_context().init_two_view(xy_ref, xy_cur, K, thr_px=3.0, n_hyp=_geom.N_HYPOTHESES, seed=_geom.SEED)
            if not np.isfinite(g["E"]).all():
                return self._failure(putative)
            R, t = g["R"], np.asarray(g["t"], np.float64).reshape(3, 1)
            ok = g["pose_mask"]
            q, tr, dist = q[ok], tr[ok], dist[ok]
            if not len(q):  # (the reference would raise on its debug prints here)
                return self._failure([])
            X = g["X"][ok]

        # cheirality as the reference re-checks it on the float32 points: positive depth in the first camera and in the second one
        depth_cur = (X.astype(np.float64) @ R.T + t.ravel())[:, 2]
        front = np.flatnonzero((X[:, 2] > 0) & (depth_cur > 0))
        X, q, tr, dist = X[front], q[front], tr[front], dist[front]
        survivors = dmatches_from_arrays(q, tr, dist)
        if len(X) < self.min_matches // 2:
            print(f"Insufficient valid 3D points after filtering: {len(X)}")
            return self._failure(survivors)

        colours = self._colours(ref_kps, q)
        map_points = [{"position": position, "color": colour, "keypoint_references": {0: a, 1: b}, "observed_frames": [0, 1]}
                      for position, colour, a, b in zip(X, colours, q.tolist(), tr.tolist())]

        self.current_frame_keypoints = current_keypoints
        self.initialization_done = True
        return True, R, t, map_points, survivors

    # ------------------------------------------------------------------ visualisation (needs cv2; not on the HIP path)
    def draw_initialization(self, first_image, current_image, matches):
        from .types import HAVE_CV2
        if not HAVE_CV2:
            raise RuntimeError("draw_initialization is visualisation and needs cv2 (out of scope of the HIP path)")
        import cv2

        def as_bgr(img):
            return cv2.cvtColor(img, cv2.COLOR_GRAY2BGR) if img.ndim == 2 else img.copy()

        return cv2.drawMatches(as_bgr(first_image), self.first_frame_keypoints, as_bgr(current_image),
                               self.current_frame_keypoints, matches, None, matchColor=(0, 255, 0),
                               singlePointColor=(255, 0, 0), flags=cv2.DrawMatchesFlags_NOT_DRAW_SINGLE_POINTS)
