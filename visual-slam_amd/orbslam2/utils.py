"""Geometry helpers with the reference's names and signatures (src/orbslam2/utils.py:56-70,120-160), on MI355X.

cv2.findEssentialMat / recoverPose / triangulatePoints / findFundamentalMat are replaced by the HIP two-view kernels; every
function is an independent native call (recover_pose accepts any E, not only the one calculate_essential_matrix returned).
The YAML / PLY helpers of the reference's utils.py are out of scope; undistort_image is built (mo_undistort)."""
import numpy as np

import vslam_amd

RANSAC = 8  # numeric value of cv2.RANSAC, accepted for signature compatibility
N_HYPOTHESES = 4096
SEED = 4096

def undistort_image(image, camera_matrix, distortion):
    """utils.py:40-52: cv2.undistort(image, camera_matrix, distortion) on the device (grayscale or BGR uint8).  Coefficients beyond
    (k1, k2, p1, p2, k3) are not supported; the reference's configs hold exactly these five."""
    d = np.asarray(distortion, np.float64).ravel()
    if len(d) > 5 and np.any(d[5:] != 0):
        raise NotImplementedError("only the 5-coefficient model (k1, k2, p1, p2, k3) of the reference's configs is built")
    return vslam_amd.default_context().undistort(image, camera_matrix, d[:5])


def calculate_essential_matrix(points1, points2, camera_matrix, method=RANSAC, prob=0.999, threshold=1.0):
    """utils.py:120-126.  `method` is accepted for signature compatibility (the device path is the parallel 8-point RANSAC of
    north_star); `prob` likewise: all N_HYPOTHESES hypotheses are always scored."""
    r = vslam_amd.default_context().init_two_view(points1, points2, camera_matrix, thr_px=threshold, prob=prob,
                                                  n_hyp=N_HYPOTHESES, seed=SEED)
    if r["n_good"] == 0 and not np.isfinite(r["E"]).all():
        return None, None
    return r["E"], r["ransac_mask"].astype(np.uint8).reshape(-1, 1)


def recover_pose(E, points1, points2, camera_matrix, mask=None):
    """utils.py:129-134: cv2.recoverPose for ANY essential matrix (mo_recover_pose: decomposition, cheirality vote on the masked
    points, winner's R, t and mask) -> (n_good, R, t, mask (N, 1) uint8 with 255 for survivors, like cv2)."""
    r = vslam_amd.default_context().recover_pose(E, points1, points2, camera_matrix, mask)
    return r["n_good"], r["R"], r["t"], (r["mask"].astype(np.uint8) * 255).reshape(-1, 1)


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
                          ratio_threshold=0.75, threshold_percent=0.02, ransac_threshold=1.0, pair_index=0):
    """The body of the reference's Tracker._track_from_last_frame (tracker.py:198-266) as ONE device call: matcher.match ->
    filter_matches_by_geometric_distance (threshold_percent of (w + h) / 2) -> filter_matches_by_distance (2 x median) ->
    cv2.findEssentialMat(RANSAC, 0.999, 1.0) -> cv2.recoverPose.  Tracker calls cv2 directly for the last two, so the caller
    needs this one-line replacement.  -> (success, T 4x4 float64, inlier matches as DMatch list); like the
    reference it fails with fewer than 8 filtered matches or without a valid essential matrix.
    When the arguments are the very arrays detect_and_compute returned for the last two frames, both are still resident on the device
    and nothing is uploaded.  pair_index (an extension): position of this pair in the sequence - a loop that counts its pairs draws
    the sampling streams of the batched mode (vslam_amd.stream) and gets its poses bit for bit."""
    from .types import dmatches_from_arrays, keypoints_to_array
    if last_keypoints is None or last_descriptors is None or descriptors is None:
        return False, None, []
    h, w = image_shape[:2]
    r = vslam_amd.default_context().track_pair(keypoints_to_array(last_keypoints), last_descriptors, keypoints_to_array(keypoints),
                                               descriptors, w, h, camera_matrix, ratio=ratio_threshold,
                                               disp_frac=threshold_percent, thr_px=ransac_threshold, n_hyp=N_HYPOTHESES, seed=SEED,
                                               pair_index=pair_index)
    if len(r["sel"]) < 8 or not np.isfinite(r["R"]).all():
        return False, None, []
    T = np.eye(4)
    T[:3, :3] = r["R"]
    T[:3, 3] = r["t"].reshape(3)
    ok = r["inlier"]
    sel = r["sel"][ok]
    inliers = dmatches_from_arrays(sel[:, 0], sel[:, 1], r["sel_dist"][ok])
    return True, T, inliers


def calculate_fundamental_matrix(points1, points2, threshold=3.0, prob=0.99):
    """cv2.findFundamentalMat(points1, points2, cv2.FM_RANSAC, threshold, prob) as the reference calls it (matcher.py:191,
    local_mapper.py:136) -> (F 3x3 or None, mask (N, 1) uint8 or None)"""
    F, mask = vslam_amd.default_context().find_fundamental(points1, points2, thr_px=threshold, prob=prob)
    if F is None:
        return None, None
    return F, mask.astype(np.uint8).reshape(-1, 1)


def triangulate_new_map_points(prev_keypoints, prev_descriptors, cur_keypoints, cur_descriptors, pose1, pose2, camera_matrix,
                               ratio=0.8, threshold=3.0):
    """The geometric core of the reference's LocalMapper._process_new_keyframe (local_mapper.py:116-149): knnMatch(k=2) with the
    0.8 ratio test -> cv2.findFundamentalMat(FM_RANSAC, 3.0) -> keep inliers -> triangulate with the two keyframe poses.
    pose1 / pose2: 4x4 keyframe poses.  -> (points_3d (N, 3) float32, inlier matches as DMatch list); ([], []) when fewer than
    8 ratio-test matches or no fundamental matrix is found (the reference returns early in both cases)."""
    from .types import dmatches_from_arrays
    idx, dist, keep = vslam_amd.default_context().match_knn2_ratio(np.uint8(prev_descriptors), np.uint8(cur_descriptors), ratio)
    two = idx[:, 1] >= 0  # local_mapper.py:123: only pairs with two neighbours take part
    q = np.flatnonzero(keep & two)
    if len(q) < 8:
        return np.zeros((0, 3), np.float32), []
    from .types import points_of
    points1 = points_of(prev_keypoints, q)
    points2 = points_of(cur_keypoints, idx[q, 0])
    F, mask = calculate_fundamental_matrix(points1, points2, threshold)
    if F is None:
        return np.zeros((0, 3), np.float32), []
    m = mask.ravel() > 0
    P1 = compute_projection_matrix(pose1[:3, :3], pose1[:3, 3], camera_matrix)
    P2 = compute_projection_matrix(pose2[:3, :3], pose2[:3, 3], camera_matrix)
    pts = convert_to_3d_points(triangulate_points(points1[m], points2[m], P1, P2))
    matches = dmatches_from_arrays(q[m], idx[q[m], 0], dist[q[m], 0])
    return pts, matches
