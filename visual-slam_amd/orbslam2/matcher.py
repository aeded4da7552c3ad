"""DescriptorMatcher -- same public surface as the reference's src/orbslam2/matcher.py:12-217.

  __init__   matcher.py:17-42   'bruteforce-hamming' -> the HIP 2-NN matcher; unknown type -> ValueError (as the reference);
                                'flann' (LSH, never selected by any config, non-deterministic) is out of scope
  match      matcher.py:44-83   None / empty -> []; cast to uint8; knnMatch(k=2) + Lowe ratio in query order
  filter_matches_by_geometric_distance  matcher.py:109-142   host list logic
  filter_matches_by_distance            matcher.py:144-169   host list logic
  filter_matches_by_fundamental         matcher.py:171-200   HIP F-matrix RANSAC (mo_find_fundamental)
"""
import math

import numpy as np

import vslam_amd
from .types import DMatch, dmatches_from_arrays, match_arrays, points_of


class _NativeBFMatcher:
    """What the reference keeps in `self.matcher` (cv2.BFMatcher(NORM_HAMMING)): knnMatch(k<=2)."""

    def knnMatch(self, descriptors1, descriptors2, k=2):
        if k not in (1, 2):
            raise NotImplementedError("only k <= 2 is built (the reference calls knnMatch(k=2), matcher.py:70)")
        idx, dist, _ = vslam_amd.default_context().match_knn2_ratio(descriptors1, descriptors2, None)
        out = []
        for q in range(len(idx)):
            row = []
            for j in range(k):
                if idx[q, j] >= 0:
                    row.append(DMatch(q, int(idx[q, j]), 0, float(dist[q, j])))
            out.append(tuple(row))
        return tuple(out)


class DescriptorMatcher:
    def __init__(self, matcher_type='bruteforce-hamming', ratio_threshold=0.75):
        self.ratio_threshold = ratio_threshold
        if matcher_type == 'bruteforce-hamming':
            self.matcher = _NativeBFMatcher()
        elif matcher_type == 'flann':
            raise NotImplementedError("FLANN-LSH matching is out of scope (unused by every config of the reference)")
        else:
            raise ValueError(f"Unknown matcher type: {matcher_type}")

    def match(self, descriptors1, descriptors2, ratio_test=True):
        if descriptors1 is None or descriptors2 is None:
            return []
        if descriptors1.shape[0] == 0 or descriptors2.shape[0] == 0:
            return []
        if descriptors1.dtype != np.uint8:
            descriptors1 = np.uint8(descriptors1)
        if descriptors2.dtype != np.uint8:
            descriptors2 = np.uint8(descriptors2)
        ratio = float(self.ratio_threshold) if ratio_test else None
        idx, dist, keep = vslam_amd.default_context().match_knn2_ratio(descriptors1, descriptors2, ratio)
        # survivors as Python scalars in three bulk conversions (numpy scalar reads per match were a third of this call)
        q = np.nonzero(keep)[0]
        return dmatches_from_arrays(q, idx[q, 0], dist[q, 0])

    def match_with_mask(self, descriptors1, descriptors2, mask):
        raise NotImplementedError("match_with_mask has no caller in the reference (dead code, SURVEY.md 2.1)")

    def filter_matches_by_geometric_distance(self, keypoints1, keypoints2, matches, threshold_percent, image_shape):
        height, width = image_shape[:2]
        limit = ((width + height) / 2.0) * threshold_percent
        if not matches:
            return []
        # the reference's per-match math.hypot(dx, dy) <= limit on Python floats (float64 of the float32 coordinates): vectorised,
        # with the matches within rounding distance of the limit decided by math.hypot itself
        q, t, _ = match_arrays(matches)  # (the arrays of an untouched match list; no DMatch object is created for the rejected ones)
        a = points_of(keypoints1, q).astype(np.float64)
        b = points_of(keypoints2, t).astype(np.float64)
        dx, dy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
        dist = np.hypot(dx, dy)
        near = dist <= limit
        for i in np.flatnonzero(np.abs(dist - limit) <= 1e-9 * max(limit, 1.0)).tolist():
            near[i] = math.hypot(dx[i], dy[i]) <= limit
        return [matches[i] for i in np.flatnonzero(near).tolist()]

    def filter_matches_by_distance(self, matches, distance_threshold=None):
        if not matches:
            return []
        _, _, d = match_arrays(matches)
        order = np.argsort(d, kind="stable")  # sorted(matches, key=lambda m: m.distance): stable, ties keep their order
        if distance_threshold is None:
            distance_threshold = np.median(d) * 2.0
        return [matches[i] for i in order[d[order] < distance_threshold].tolist()]

    def filter_matches_by_fundamental(self, keypoints1, keypoints2, matches, threshold=3.0):
        """matcher.py:171-200: cv2.findFundamentalMat(points1, points2, FM_RANSAC, threshold, 0.99) -> (inlier matches, bool mask);
        fewer than 8 matches are returned unfiltered with an all-true mask, like the reference."""
        if len(matches) < 8:
            return matches, np.ones(len(matches), dtype=bool)
        q, t, _ = match_arrays(matches)
        points1 = points_of(keypoints1, q)
        points2 = points_of(keypoints2, t)
        F, mask = vslam_amd.default_context().find_fundamental(points1, points2, thr_px=threshold, prob=0.99)
        if F is None:  # cv2 returns mask None here and the reference would raise on mask.ravel(); no model -> no inliers
            return [], np.zeros(len(matches), dtype=bool)
        return [matches[i] for i in np.flatnonzero(np.asarray(mask).ravel()).tolist()], mask

    def draw_matches(self, img1, keypoints1, img2, keypoints2, matches, flags=0):
        from .types import HAVE_CV2
        if not HAVE_CV2:
            raise RuntimeError("draw_matches is visualisation and needs cv2 (out of scope of the HIP path)")
        import cv2
        return cv2.drawMatches(img1, keypoints1, img2, keypoints2, matches, None, flags=flags)
