#!/usr/bin/env python3
"""Frame loop in the style of the reference's src/tests/tester_map.py:32-110, driven by the drop-in classes only.

The reference's Tracker/LocalMapper are not part of this repo; this small state machine reproduces the calls Tracker makes
on the three classes (tracker.py:87,162,168-170,214,221,230,242-249) on a synthetic sequence (a camera translating past a
two-depth scene), so the drop-in can be exercised end to end on an MI355X without cv2 or a video file:
    NOT_INITIALIZED: frame 0 -> set_first_frame, frame 1.. -> initialize (two-view map)
    TRACKING: match against the previous frame, the two match filters, essential matrix at threshold 1.0 + pose - as one
              fused device call (orbslam2.utils.track_from_last_frame, default) or with the filters as Python loops over
              DMatch objects like the reference (--python-filters)
Usage: python visual-slam_amd/examples/run_frames.py [--frames 30] [--grid] [--python-filters]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from orbslam2 import utils as geom                      # noqa: E402
from orbslam2.extractor import ORBExtractor             # noqa: E402
from orbslam2.initializer import MapInitializer         # noqa: E402
from orbslam2.matcher import DescriptorMatcher          # noqa: E402

W, H = 640, 480


def scene(seed, w, h, rects):
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.full((h, w), 128.0) + 25 * np.sin(np.arange(w) / 47.0)[None, :] + 25 * np.cos(np.arange(h) / 31.0)[:, None]
    for _ in range(rects):
        rw, rh = rng.integers(5, 22, size=2)
        x, y = rng.integers(0, w - 1), rng.integers(0, h - 1)
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    return np.clip(img + rng.normal(0, 1, img.shape), 0, 255)


def make_sequence(n, seed=7):
    span = 16 * n + W
    bg, fg, mk = scene(seed, span, H, 800 * span // W), scene(seed + 1, span, H, 800 * span // W), scene(seed + 2, span, H, 40 * span // W)
    rng = np.random.default_rng(seed)
    for i in range(n):
        fr = np.where(mk[:, 16 * i:16 * i + W] > 130, fg[:, 16 * i:16 * i + W], bg[:, 8 * i:8 * i + W])
        yield np.clip(np.rint(fr + rng.normal(0, 1, fr.shape)), 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--grid", action="store_true", help="use extract_features(distributed=True) like Tracker.process_frame")
    ap.add_argument("--python-filters", action="store_true", help="tracking step through the per-method API (Python filter loops)")
    args = ap.parse_args()
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])    # configs/monocular.yaml:3
    extractor = ORBExtractor(n_features=2000, scale_factor=1.2, n_levels=8, ini_threshold=20, min_threshold=7)
    matcher = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)
    initializer = MapInitializer(K)
    state, last, poses, n_map = "NOT_INITIALIZED", None, [], 0
    t0 = time.perf_counter()
    for i, frame in enumerate(make_sequence(args.frames)):
        kps, desc = extractor.extract_features(frame, distributed=args.grid)
        if args.grid and desc is not None and len(desc) != len(kps):
            kps, desc = extractor.detect_and_compute(frame)  # the reference's index quirk makes grid keypoints unusable here
        if state == "NOT_INITIALIZED":
            if initializer.first_frame_keypoints is None:
                initializer.set_first_frame(kps, desc, frame)
            else:
                ok, R, t, pts, matches = initializer.initialize(kps, desc, matcher, frame)
                if ok:
                    state, n_map = "TRACKING", len(pts)
                    poses.append((R, t))
                    print("frame %d: initialised, %d map points, t = %s" % (i, n_map, np.round(t.ravel(), 3)))
        elif not args.python_filters:
            ok, T, inl = geom.track_from_last_frame(last[0], last[1], kps, desc, K, frame.shape, ratio_threshold=0.75,
                                                    threshold_percent=0.02 * 2.5)
            if ok:
                poses.append((T[:3, :3], T[:3, 3:4]))
                if i % 5 == 0:
                    print("frame %d: %d pose inliers, t = %s" % (i, len(inl), np.round(T[:3, 3], 3)))
        else:
            m = matcher.match(last[1], desc)
            m = matcher.filter_matches_by_geometric_distance(last[0], kps, m, 0.02 * 2.5, frame.shape)
            m = matcher.filter_matches_by_distance(m)
            if len(m) >= 8:
                p1 = np.float32([last[0][x.queryIdx].pt for x in m])
                p2 = np.float32([kps[x.trainIdx].pt for x in m])
                E, mask = geom.calculate_essential_matrix(p1, p2, K, prob=0.999, threshold=1.0)   # tracker.py:242
                if E is not None:
                    n_in, R, t, _ = geom.recover_pose(E, p1, p2, K, mask)                          # tracker.py:249
                    poses.append((R, t))
                    if i % 5 == 0:
                        print("frame %d: %d matches, %d pose inliers, t = %s" % (i, len(m), n_in, np.round(t.ravel(), 3)))
        last = (kps, desc)
    dt = time.perf_counter() - t0
    print("%d frames in %.2f s (%.1f frames/s through the Python drop-in classes), state %s, %d poses"
          % (args.frames, dt, args.frames / dt, state, len(poses)))
    return state, poses, n_map


if __name__ == "__main__":
    main()
