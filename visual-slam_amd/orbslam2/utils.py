"""Geometry helpers with the reference's names and signatures (src/orbslam2/utils.py:56-70,120-160), on MI355X.

cv2.findEssentialMat / recoverPose / triangulatePoints are replaced by the HIP two-view kernels.  The essential-matrix
estimate and the pose recovery are ONE fused native call (mo_init_two_view); calculate_essential_matrix runs it and
recover_pose returns the pose of that same call (looked up by the point arrays), so the reference's call sequence
    E, mask = calculate_essential_matrix(p1, p2, K, threshold=3.0); _, R, t, mask_pose = recover_pose(E, p1, p2, K, mask)
keeps working unchanged.  The I/O helpers of the reference's utils.py (YAML, PLY, undistort) are out of scope."""
import numpy as np

import vslam_amd

RANSAC = 8  # numeric value of cv2.RANSAC, accepted for signature compatibility
N_HYPOTHESES = 4096
SEED = 4096

_last = {}


def _key(p1, p2, K):
    return (np.asarray(p1, np.float32).tobytes(), np.asarray(p2, np.float32).tobytes(), np.asarray(K, np.float64).tobytes())


def _two_view(points1, points2, camera_matrix, threshold, prob):
    r = vslam_amd.default_context().init_two_view(points1, points2, camera_matrix, thr_px=threshold, prob=prob,
                                                  n_hyp=N_HYPOTHESES, seed=SEED)
    _last.clear()
    _last[_key(points1, points2, camera_matrix)] = (r, float(threshold))
    return r


def calculate_essential_matrix(points1, points2, camera_matrix, method=RANSAC, prob=0.999, threshold=1.0):
    r = _two_view(points1, points2, camera_matrix, threshold, prob)
    if r["n_good"] == 0 and not np.isfinite(r["E"]).all():
        return None, None
    return r["E"], r["ransac_mask"].astype(np.uint8).reshape(-1, 1)


def recover_pose(E, points1, points2, camera_matrix, mask=None):
    hit = _last.get(_key(points1, points2, camera_matrix))
    if hit is None or E is None or not np.allclose(hit[0]["E"], E):
        raise RuntimeError("recover_pose must follow calculate_essential_matrix on the same points (fused native call)")
    r = hit[0]
    pm = r["pose_mask"]
    if mask is not None:
        pm = pm & (np.asarray(mask).ravel() != 0)
    return int(pm.sum()), r["R"].copy(), r["t"].copy(), (pm.astype(np.uint8) * 255).reshape(-1, 1)


def compute_projection_matrix(R, t, camera_matrix):
    t = np.array(t)
    if t.ndim == 1 or (t.ndim == 2 and t.shape == (1, 3)):
        t = t.reshape(3, 1)
    return camera_matrix @ np.hstack((R, t))


def triangulate_points(points1, points2, P1, P2):
    X4 = vslam_amd.default_context().triangulate_points(P1, P2, points1, points2)
    return np.ascontiguousarray(X4.T)  # (4, N) float32, like cv2.triangulatePoints


def convert_to_3d_points(points_4d):
    if points_4d.shape[0] != 4:
        points_4d = points_4d.T
    return (points_4d[:3] / points_4d[3]).T


def track_from_last_frame(last_keypoints, last_descriptors, keypoints, descriptors, camera_matrix, image_shape,
                          ratio_threshold=0.75, threshold_percent=0.02, ransac_threshold=1.0):
    """The body of the reference's Tracker._track_from_last_frame (tracker.py:198-266) as ONE device call: matcher.match ->
    filter_matches_by_geometric_distance (threshold_percent of (w + h) / 2) -> filter_matches_by_distance (2 x median) ->
    cv2.findEssentialMat(RANSAC, 0.999, 1.0) -> cv2.recoverPose.  Tracker calls cv2 directly for the last two, so the caller
    needs this one-line replacement.  -> (success, T 4x4 float64, inlier matches as DMatch list); like the
    reference it fails with fewer than 8 filtered matches or without a valid essential matrix."""
    from .types import DMatch, keypoints_to_array
    if last_keypoints is None or last_descriptors is None or descriptors is None:
        return False, None, []
    h, w = image_shape[:2]
    r = vslam_amd.default_context().track_pair(keypoints_to_array(last_keypoints), last_descriptors, keypoints_to_array(keypoints),
                                               descriptors, w, h, camera_matrix, ratio=ratio_threshold,
                                               disp_frac=threshold_percent, thr_px=ransac_threshold, n_hyp=N_HYPOTHESES, seed=SEED)
    if len(r["sel"]) < 8 or not np.isfinite(r["R"]).all():
        return False, None, []
    T = np.eye(4)
    T[:3, :3] = r["R"]
    T[:3, 3] = r["t"].reshape(3)
    inliers = [DMatch(int(q), int(t), 0, float(d)) for (q, t), d, ok in zip(r["sel"], r["sel_dist"], r["inlier"]) if ok]
    return True, T, inliers
