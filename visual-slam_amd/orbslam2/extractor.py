"""ORBExtractor -- same public surface as the reference's src/orbslam2/extractor.py:12-174, computed on MI355X.

Reference behaviour mirrored per method (file:line of the reference):
  __init__            extractor.py:19-48   stores the five parameters; builds the ORB handle with edgeThreshold=31,
                                           firstLevel=0, WTA_K=2, HARRIS score, patchSize=31, fastThreshold=min_threshold
                                           (ini_threshold is stored and, as in the reference, never used)
  detect_and_compute  extractor.py:50-67   BGR->gray if needed, then orb.detectAndCompute(image, None)
  compute             extractor.py:68-83   orb.compute(image, keypoints); returns the ORIGINAL keypoint list next to the
                                           descriptors of the keypoints cv2 kept (the reference's index quirk is kept)
  extract_features    extractor.py:146-160 dispatch on `distributed`
  distribute_keypoints extractor.py:85-144 grid Shi-Tomasi (min-eigenvalue map once, one workgroup per cell) + compute
"""
import os

import numpy as np

import vslam_amd
from .types import keypoints_at_lazy, keypoints_from_array, keypoints_to_array


def _default_order():
    v = os.environ.get("VSLAM_AMD_SELECT_ORDER", "libstdc++").lower()
    return vslam_amd.ORDER_MSVC if v in ("msvc", "windows", "1") else vslam_amd.ORDER_LIBSTDCXX


class _NativeORB:
    """What the reference keeps in `self.orb` (a cv2.ORB): exposes detectAndCompute / compute / detect."""

    def __init__(self, prm):
        self.prm = prm

    def detectAndCompute(self, image, mask=None):
        if mask is not None:
            raise NotImplementedError("masks are not used by the reference (extractor.py:65 passes None)")
        (kps, desc), = vslam_amd.default_context().orb_detect_compute(image, self.prm)
        return keypoints_from_array(kps), desc

    def detect(self, image, mask=None):
        if mask is not None:
            raise NotImplementedError("masks are not used by the reference")
        (kps, _), = vslam_amd.default_context().orb_detect_compute(image, self.prm, want_desc=False)
        return keypoints_from_array(kps)

    def compute(self, image, keypoints):
        arr = keypoints_to_array(keypoints)
        kept, desc = vslam_amd.default_context().orb_compute(image, self.prm, arr)
        from .types import KeyPointSeq
        if isinstance(keypoints, KeyPointSeq) and keypoints.pristine:
            kept_kps = KeyPointSeq(arr[kept])                 # records, no objects: the input sequence stays pristine
        else:
            kept_kps = tuple(keypoints[i] for i in kept)
        return kept_kps, (desc if len(kept) else None)


class ORBExtractor:
    """ORB feature extractor (drop-in for the reference class of the same name)."""

    def __init__(self, n_features=2000, scale_factor=1.2, n_levels=8, ini_threshold=20, min_threshold=7):
        self.n_features = n_features
        self.scale_factor = scale_factor
        self.n_levels = n_levels
        self.ini_threshold = ini_threshold
        self.min_threshold = min_threshold
        self.orb = _NativeORB(vslam_amd.orb_params(nfeatures=n_features, scale_factor=scale_factor, nlevels=n_levels,
                                                   edge_threshold=31, fast_threshold=min_threshold,
                                                   select_order=_default_order()))

    def detect_and_compute(self, image):
        image = np.asarray(image)
        return self.orb.detectAndCompute(image, None)  # 3-channel input is converted to gray on the device

    def compute(self, image, keypoints):
        image = np.asarray(image)
        return keypoints, self.orb.compute(image, keypoints)[1]

    def distribute_keypoints(self, image, n_features=None, aligned=False):
        """Grid-based detection (reference extractor.py:85-144): 8x8 cells, Shi-Tomasi corners per cell
        (maxCorners = n_features // 64, qualityLevel 0.01, minDistance 10), KeyPoint(x, y, 31) each, then
        orb.compute on all of them.  Like the reference, the returned list holds ALL corners while the descriptor
        rows are those cv2 keeps (corners within 31 px of the border are dropped by compute).  The list is a KeyPointList: records
        that answer like the reference's list and create a KeyPoint object when one is asked for (VSLAM_AMD_KEYPOINTS=tuple: the plain list).
        aligned=True (an extension, not in the reference): only the kept corners are returned - as a lazy KeyPointSeq, like
        detect_and_compute - so that keypoint i belongs to descriptor row i: what a caller needs to index keypoints with match indices."""
        if n_features is None:
            n_features = self.n_features
        image = np.asarray(image)
        # corners, KeyPoint(x, y, 31) records of the ones orb.compute keeps, and their descriptors in ONE device call
        # (mo_orb_grid_detect_compute: one upload, one synchronisation; the records never exist as Python objects)
        if aligned:  # the kept corners as a record sequence, resident on the device with their descriptors (like detect_and_compute)
            from .types import KeyPointSeq
            _, kept, descriptors, rec = vslam_amd.default_context().grid_detect_compute(image, self.orb.prm, n_features, records=True)
            return KeyPointSeq(rec), (descriptors if len(kept) else None)
        xy, kept, descriptors = vslam_amd.default_context().grid_detect_compute(image, self.orb.prm, n_features)
        all_keypoints = keypoints_at_lazy(xy, 31)  # the reference's list of ALL corners, as records: an object exists once it is asked for
        if not len(kept):
            descriptors = None
        return all_keypoints, descriptors

    def extract_features(self, image, distributed=True):
        if distributed:
            return self.distribute_keypoints(image)
        return self.detect_and_compute(image)

    def draw_keypoints(self, image, keypoints):
        from .types import HAVE_CV2
        if not HAVE_CV2:
            raise RuntimeError("draw_keypoints is visualisation and needs cv2 (out of scope of the HIP path)")
        import cv2
        return cv2.drawKeypoints(image, keypoints, None, flags=cv2.DRAW_MATCHES_FLAGS_DRAW_RICH_KEYPOINTS)
