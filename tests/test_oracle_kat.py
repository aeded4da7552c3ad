"""Pins the CPU oracle against the reference's only known-answer vectors.

data/groundtruth_matches/pairNN/gt.yaml keypoints1/2 = filter_close_keypoints(cv2.ORB_create(nfeatures=200)
.detect(frame_bgr), 10 px) (reference src/utils/gt_match_annotator.py:45,64-72,121-124,346-350).  The list is
order-sensitive, so 40/40 exact equality pins gray conversion, the INTER_LINEAR_EXACT pyramid, FAST-9 + NMS,
the border filter, both retainBest passes INCLUDING the std::nth_element permutation (MSVC STL - the fixtures
were generated on Windows), the Harris ranking, quotas and the level scaling of kp.pt.
"""
import os

import numpy as np
import pytest

from oracle import orb_oracle as O
from tests.helpers import greedy_min_dist, gt_pair, load_png_bgr


@pytest.mark.parametrize("pair", range(1, 21))
def test_detector_kat(pair):
    O.lib().orc_set_variant(1, 0)  # MSVC STL ordering, nth_element at n_points-1
    prm = O.params(nfeatures=200, fast_threshold=20)  # cv2.ORB_create(nfeatures=200) defaults
    d, gt = gt_pair(pair)
    for k in (1, 2):
        bgr = load_png_bgr(os.path.join(d, "img%d.png" % k))
        gray = O.bgr2gray(bgr)
        kps, _ = O.detect_and_compute(gray, prm, want_desc=False)
        got = greedy_min_dist([(float(a["x"]), float(a["y"])) for a in kps])
        exp = [tuple(p) for p in gt["keypoints%d" % k]]
        assert got == exp


def test_level_geometry_640x480():
    lw, lh, sc, q = O.levels(640, 480, O.params())
    assert lw == [640, 533, 444, 370, 309, 257, 214, 179]
    assert lh == [480, 400, 333, 278, 231, 193, 161, 134]
    assert q == [434, 362, 302, 251, 209, 175, 145, 122]
    assert sum(a * b for a, b in zip(lw, lh)) == 950532


def test_libstdcxx_order_same_set():
    """The two STL orderings keep the same keypoint SET; only the order differs."""
    from tests.helpers import synthetic_frame
    img = synthetic_frame(20250523)
    prm = O.params(nfeatures=500)
    O.lib().orc_set_variant(1, 0)
    a, da = O.detect_and_compute(img, prm)
    O.lib().orc_set_variant(0, 0)
    b, db = O.detect_and_compute(img, prm)
    O.lib().orc_set_variant(1, 0)
    key = lambda k: sorted(zip(k["octave"].tolist(), k["y"].tolist(), k["x"].tolist()))
    assert key(a) == key(b)
    ra = {(o, x, y): bytes(d) for o, x, y, d in zip(a["octave"], a["x"], a["y"], da)}
    rb = {(o, x, y): bytes(d) for o, x, y, d in zip(b["octave"], b["x"], b["y"], db)}
    assert ra == rb


def test_grid_good_features_properties():
    """Oracle restatement of the distribute_keypoints corner stage (extractor.py:104-136): structural properties."""
    from tests.helpers import synthetic_frame
    img = synthetic_frame(5)
    xy = O.grid_good_features(img, 2000)
    assert 500 < len(xy) <= 64 * 31
    cells = (xy[:, 1].astype(int) // 60) * 8 + xy[:, 0].astype(int) // 80
    assert (np.diff(cells) >= 0).all()  # cell-major order
    eig = O.min_eigen(img)
    for c in np.unique(cells):
        p = xy[cells == c]
        assert len(p) <= 31
        q = eig[p[:, 1].astype(int), p[:, 0].astype(int)]
        assert (np.diff(q) <= 0).all()  # best first inside a cell
        d = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(len(p)) * 100
        assert d.min() >= 10  # minDistance
