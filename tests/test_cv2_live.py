"""Live parity against cv2 itself -- runs only where `import cv2` succeeds (it does not in the build container nor on the
GPU box today: these tests SKIP there; nothing is installed or fetched to change that).

What they would pin the day a cv2 wheel is present: everything the reference's own
fixtures leave unpinned -- IC angle / fastAtan2, the 7x7 Gaussian, the rBRIEF bits, Hamming distances and knn indices --
by calling cv2 DIRECTLY with the arguments the reference passes (src/orbslam2/extractor.py:38-48,65; matcher.py:29,70).
The reference's .py files are never imported.

The order of equal-response keypoints depends on the STL cv2 was linked against: keypoints are compared as
a set first and in order for the select_order that matches the platform.
"""
import sys

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from tests.helpers import synthetic_frame  # noqa: E402


def _cv2_orb(nfeatures=2000, fast=7):
    # extractor.py:38-48
    return cv2.ORB_create(nfeatures=nfeatures, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2,
                          scoreType=cv2.ORB_HARRIS_SCORE, patchSize=31, fastThreshold=fast)


def _as_rows(kps):
    return np.array([(k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave) for k in kps], np.float32)


def _rows_of(arr):
    return np.stack([arr["x"], arr["y"], arr["size"], arr["angle"], arr["response"], arr["octave"].astype(np.float32)], 1)


def _platform_order():
    import vslam_amd as V
    return V.ORDER_MSVC if sys.platform.startswith("win") else V.ORDER_LIBSTDCXX


def _compare(kps_cv, desc_cv, arr, desc):
    a, b = _as_rows(kps_cv), _rows_of(arr)
    assert len(a) == len(b)
    ka = np.lexsort(a.T[::-1]); kb = np.lexsort(b.T[::-1])
    assert np.array_equal(a[ka], b[kb]), "keypoint SET differs from cv2 (pt, size, angle, response, octave)"
    assert np.array_equal(desc_cv[ka], desc[kb]), "descriptors differ from cv2"
    assert np.array_equal(a, b) and np.array_equal(desc_cv, desc), "keypoint ORDER differs from cv2 (select_order / STL)"


def test_oracle_equals_cv2_detect_and_compute():
    from oracle import orb_oracle as O
    img = synthetic_frame(20250523)
    kps, desc = _cv2_orb().detectAndCompute(img, None)
    O.lib().orc_set_variant(1 if sys.platform.startswith("win") else 0, 0)  # (stl, nth variant)
    arr, d = O.detect_and_compute(img, O.params(nfeatures=2000))
    _compare(kps, desc, arr, d)


def test_oracle_equals_cv2_knn():
    from oracle import orb_oracle as O
    a, b = synthetic_frame(1), synthetic_frame(2)
    _, d1 = _cv2_orb().detectAndCompute(a, None)
    _, d2 = _cv2_orb().detectAndCompute(b, None)
    knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(d1, d2, k=2)  # matcher.py:29,70
    idx, dist = O.match_knn2(d1, d2)
    assert np.array_equal(idx, np.array([[m.trainIdx for m in row] for row in knn], np.int32))
    assert np.array_equal(dist, np.array([[int(m.distance) for m in row] for row in knn], np.int32))


@pytest.mark.gpu
def test_hip_equals_cv2_detect_and_compute():
    import vslam_amd as V
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    for seed, nf, fast in ((20250523, 2000, 7), (5, 500, 20)):
        img = synthetic_frame(seed)
        kps, desc = _cv2_orb(nf, fast).detectAndCompute(img, None)
        (arr, d), = ctx.orb_detect_compute(img, V.orb_params(nfeatures=nf, fast_threshold=fast, select_order=_platform_order()))
        _compare(kps, desc, arr, d)
    ctx.close()


@pytest.mark.gpu
def test_hip_equals_cv2_knn_and_ratio():
    import vslam_amd as V
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    a, b = synthetic_frame(1), synthetic_frame(2)
    _, d1 = _cv2_orb().detectAndCompute(a, None)
    _, d2 = _cv2_orb().detectAndCompute(b, None)
    knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(d1, d2, k=2)
    idx, dist, keep = ctx.match_knn2_ratio(d1, d2, 0.75)
    assert np.array_equal(idx, np.array([[m.trainIdx for m in row] for row in knn], np.int32))
    assert np.array_equal(dist, np.array([[int(m.distance) for m in row] for row in knn], np.int32))
    good = [row[0].queryIdx for row in knn if len(row) >= 2 and row[0].distance < 0.75 * row[1].distance]  # matcher.py:73-81
    assert np.array_equal(np.nonzero(keep)[0], np.array(good))
    ctx.close()


@pytest.mark.gpu
def test_hip_two_view_agrees_with_cv2_on_clean_data():
    """cv2's 5-point RANSAC and the 8-point RANSAC prescribed by north_star are different estimators: on outlier-free,
    noise-free correspondences both must recover the same R, t (1e-4) -- the only regime where they are comparable."""
    import vslam_amd as V
    from oracle import geom_oracle as G
    s = G.synthetic_two_view(seed=9, n=400, outlier_frac=0.0)
    E, mask = cv2.findEssentialMat(s["p1"], s["p2"], s["K"], method=cv2.RANSAC, prob=0.999, threshold=3.0)  # utils.py:120-126
    _, R, t, _ = cv2.recoverPose(E, s["p1"], s["p2"], s["K"], mask=mask)  # utils.py:129-134
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    g = ctx.init_two_view(s["p1"], s["p2"], s["K"], thr_px=3.0)
    assert np.linalg.norm(g["R"] - R) < 1e-3 and np.linalg.norm(g["t"] - t) < 1e-3  # cv2 does not refit: its own error floor
    ctx.close()
