"""MapInitializer -- drop-in for the reference class of the same name (src/orbslam2/initializer.py:12-181).

Public surface kept: MapInitializer(camera_matrix, min_matches=10, min_inliers_ratio=0.9), set_first_frame(kp, des, image),
initialize(kp, des, matcher, image) -> (success, R, t, map_points, matches), draw_initialization(img1, img2, matches) and the
attributes initialization_done / first_frame_* / current_frame_keypoints that Tracker reads (tracker.py:43,106,162,168-170).

What initialize() computes, stage by stage, and where it runs:
  1. matcher.match(first descriptors, current descriptors)            -> HIP k_match           (initializer.py:67)
  2. essential matrix, RANSAC threshold 3 px + relative pose           -> HIP two-view kernels  (initializer.py:79-83)
  3. P1 = K[I|0], P2 = K[R|t]; triangulation of the pose-mask matches  -> HIP DLT kernel        (initializer.py:86-95)
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
        """Colour of the first frame under each reference keypoint (BGR triple; grey replicated; red when outside)."""
        img = self.first_frame_image
        h, w = img.shape[:2]
        colours = []
        for qi in query_indices:
            px, py = (int(v) for v in keypoints[qi].pt)
            if not (0 <= px < w and 0 <= py < h):
                colours.append(np.array([0, 0, 255]))
            elif img.ndim == 3:
                colours.append(img[py, px, :])
            else:
                colours.append(np.array([img[py, px]] * 3))
        return colours

    # ------------------------------------------------------------------ the hot path
    def initialize(self, current_keypoints, current_descriptors, matcher, current_image):
        ref_kps, ref_desc = self.first_frame_keypoints, self.first_frame_descriptors
        if ref_kps is None or ref_desc is None:
            return self._failure(None)

        putative = matcher.match(ref_desc, current_descriptors)
        if len(putative) < self.min_matches:
            print(f"Not enough matches for initialization: {len(putative)} < {self.min_matches}")
            return self._failure(putative)

        K = self.camera_matrix
        xy_ref = _pixels(ref_kps, [m.queryIdx for m in putative])
        xy_cur = _pixels(current_keypoints, [m.trainIdx for m in putative])
        E, ransac_mask = _geom.calculate_essential_matrix(xy_ref, xy_cur, K, threshold=3.0)
        if E is None:
            return self._failure(putative)
        _, R, t, pose_mask = _geom.recover_pose(E, xy_ref, xy_cur, K, ransac_mask)
        t = np.asarray(t, np.float64).reshape(3, 1)

        keep = np.flatnonzero(np.asarray(pose_mask).ravel())
        survivors = [putative[i] for i in keep]
        if not survivors:  # (the reference would raise on its debug prints here)
            return self._failure(survivors)

        P_ref = _geom.compute_projection_matrix(np.eye(3), np.zeros((3, 1)), K)
        P_cur = _geom.compute_projection_matrix(R, t, K)
        X = _geom.convert_to_3d_points(_geom.triangulate_points(xy_ref[keep], xy_cur[keep], P_ref, P_cur))

        # cheirality: positive depth in the first camera and in the second one (X' = R X + t)
        depth_cur = (X.astype(np.float64) @ R.T + t.ravel())[:, 2]
        front = np.flatnonzero((X[:, 2] > 0) & (depth_cur > 0))
        X = X[front]
        survivors = [survivors[i] for i in front]
        if len(X) < self.min_matches // 2:
            print(f"Insufficient valid 3D points after filtering: {len(X)}")
            return self._failure(survivors)

        colours = self._colours(ref_kps, [m.queryIdx for m in survivors])
        map_points = []
        for position, colour, m in zip(X, colours, survivors):
            map_points.append({"position": position, "color": colour,
                               "keypoint_references": {0: m.queryIdx, 1: m.trainIdx}, "observed_frames": [0, 1]})

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
