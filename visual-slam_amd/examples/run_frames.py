#!/usr/bin/env python3
"""Frame loop in the style of the reference's src/tests/tester_map.py:32-110 / src/run_video.py:64-71,140-152, driven by the
drop-in classes only.

The reference's Tracker / LocalMapper are not part of this repo; this small state machine reproduces the calls Tracker makes on
the three classes (tracker.py:87,162,168-170,214,221,230,242-249):
    NOT_INITIALIZED: frame 0 -> set_first_frame, frame 1.. -> initialize (two-view map)
    TRACKING: match against the previous frame, the two match filters (threshold_percent = 0.02, tracker.py:219), essential matrix
              at threshold 1.0 + pose - as one fused device call (orbslam2.utils.track_from_last_frame, default) or with the filters
              as Python loops over DMatch objects like the reference (--python-filters)

Inputs:
    --config <yaml>   the reference's configuration file (configs/monocular.yaml): camera.camera_matrix (9 floats, row-major),
                      camera.distortion_coeffs (k1 k2 p1 p2 k3), orb.n_features / scale_factor / n_levels / ini_threshold /
                      min_threshold, matcher.matcher_type / ratio_threshold, skip_frames, max_frames.  Without it: the values of
                      configs/monocular.yaml:3,8-12 with the Tracker's ratio 0.75.
    --frames <path>   a directory of image files (sorted by name; anything PIL opens), or a .npy / .npz array [N, H, W] or
                      [N, H, W, 3] (BGR like cv2).  Frames are undistorted on the device when a distortion coefficient is non-zero
                      (run_video.py:145-149) and converted to gray on the device.  Without it: a synthetic sequence (a camera
                      translating past a two-depth scene), so the drop-in can be exercised end to end without cv2 or a video file.
    --batch N         the same sequence through vslam_amd.stream.FrameStream: frames are gathered into chunks of N and every chunk is
                      ONE batched device call (upload of the next chunk and the handling of the previous one overlapped with the
                      compute); the first two frames still initialise the map through MapInitializer, every later frame takes its
                      tracking result (against the previous frame) from the stream.  Same keypoints, descriptors, kept matches and
                      poses as the per-frame loop - at the batched mode's rate.
Usage: python visual-slam_amd/examples/run_frames.py [--config cfg.yaml] [--frames dir|file] [--max-frames 30] [--grid] [--python-filters] [--batch 64]
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
DEFAULTS = {"camera": {"camera_matrix": [320.0, 0.0, 320.0, 0.0, 320.0, 240.0, 0.0, 0.0, 1.0], "distortion_coeffs": [0.0] * 5},
            "orb": {"n_features": 2000, "scale_factor": 1.2, "n_levels": 8, "ini_threshold": 20, "min_threshold": 7},
            "matcher": {"matcher_type": "bruteforce-hamming", "ratio_threshold": 0.75}, "skip_frames": 0, "max_frames": 0}


def load_config(path):
    """the reference's utils.load_config + load_camera_intrinsics (utils.py:6-37) -> (config dict with defaults filled in, K 3x3, D)"""
    import yaml
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULTS.items()}
    if path:
        with open(path) as f:
            user = yaml.safe_load(f) or {}
        ucam = user.get("camera", {}) or {}
        if "camera_matrix" not in ucam or "distortion_coeffs" not in ucam:   # utils.py:31-32 of the reference
            raise KeyError("'camera_matrix' or 'distortion_coeffs' missing in 'camera' section of config")
        for k, v in user.items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k].update(v)
            else:
                cfg[k] = v
    cam = cfg["camera"]
    K = np.array(cam["camera_matrix"], dtype=float).reshape(3, 3)
    D = np.array(cam["distortion_coeffs"], dtype=float)
    return cfg, K, D


def scene(seed, w, h, rects):
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.full((h, w), 128.0) + 25 * np.sin(np.arange(w) / 47.0)[None, :] + 25 * np.cos(np.arange(h) / 31.0)[:, None]
    for _ in range(rects):
        rw, rh = rng.integers(5, 22, size=2)
        x, y = rng.integers(0, w - 1), rng.integers(0, h - 1)
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    return np.clip(img + rng.normal(0, 1, img.shape), 0, 255)


def synthetic_sequence(n, seed=7):
    """a camera translating along x past two depth layers: the background pans 4 px / frame, the foreground patches 8 px / frame
    (both inside the tracker's displacement gate of 2 % of (w + h) / 2 = 11.2 px, tracker.py:219)"""
    span = 8 * n + W
    bg, fg, mk = scene(seed, span, H, 800 * span // W), scene(seed + 1, span, H, 800 * span // W), scene(seed + 2, span, H, 40 * span // W)
    rng = np.random.default_rng(seed)
    for i in range(n):
        fr = np.where(mk[:, 8 * i:8 * i + W] > 130, fg[:, 8 * i:8 * i + W], bg[:, 4 * i:4 * i + W])
        yield np.clip(np.rint(fr + rng.normal(0, 1, fr.shape)), 0, 255).astype(np.uint8)


def frames_from(path):
    """a directory of images (BGR arrays like cv2.imread, in name order) or a .npy / .npz stack"""
    if os.path.isdir(path):
        from PIL import Image
        for name in sorted(os.listdir(path)):
            try:
                im = Image.open(os.path.join(path, name))
            except Exception:
                continue  # not an image (e.g. gt.yaml beside the frames)
            a = np.array(im.convert("RGB") if im.mode not in ("L", "I;16") else im.convert("L"))
            yield np.ascontiguousarray(a[:, :, ::-1]) if a.ndim == 3 else a
        return
    data = np.load(path)
    arr = data[sorted(data.files)[0]] if hasattr(data, "files") else data
    for a in arr:
        yield np.ascontiguousarray(a, dtype=np.uint8)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="", help="YAML file with the reference's keys (configs/monocular.yaml)")
    ap.add_argument("--frames", default="", help="directory of images, or .npy / .npz frame stack; default: synthetic sequence")
    ap.add_argument("--max-frames", type=int, default=30, help="frames of the synthetic sequence / cap on the input (0 = config's max_frames)")
    ap.add_argument("--grid", action="store_true", help="extract_features(distributed=True), the path Tracker.process_frame takes")
    ap.add_argument("--python-filters", action="store_true", help="tracking step through the per-method API (Python filter loops)")
    ap.add_argument("--batch", type=int, default=0, help="N > 0: the sequence through FrameStream in chunks of N frames (one batched device call each)")
    args = ap.parse_args(argv)
    cfg, K, D = load_config(args.config)
    orb, mt = cfg["orb"], cfg["matcher"]
    extractor = ORBExtractor(n_features=orb["n_features"], scale_factor=orb["scale_factor"], n_levels=orb["n_levels"],
                             ini_threshold=orb["ini_threshold"], min_threshold=orb["min_threshold"])
    matcher = DescriptorMatcher(mt["matcher_type"], ratio_threshold=mt["ratio_threshold"])
    initializer = MapInitializer(K)
    limit = args.max_frames or cfg.get("max_frames", 0) or 10 ** 9
    skip = int(cfg.get("skip_frames", 0) or 0)
    source = frames_from(args.frames) if args.frames else synthetic_sequence(limit if limit < 10 ** 9 else 30)
    state, last, poses, n_map, n_seen = "NOT_INITIALIZED", None, [], 0, 0
    t0 = time.perf_counter()
    if args.batch > 0:
        return run_batched(args, cfg, K, D, orb, mt, initializer, source, limit, skip, t0)
    for idx, frame in enumerate(source):
        if n_seen >= limit:
            break
        if skip and idx % (skip + 1) != 0:   # tester_map.py:60-63
            continue
        n_seen += 1
        if np.any(D):                        # run_video.py:145-149
            frame = geom.undistort_image(frame, K, D)
        if args.grid:                        # Tracker's default path; keypoint i belongs to descriptor row i (see distribute_keypoints)
            kps, desc = extractor.distribute_keypoints(frame, aligned=True)
        else:
            kps, desc = extractor.extract_features(frame, distributed=False)
        if state == "NOT_INITIALIZED":
            if initializer.first_frame_keypoints is None:
                initializer.set_first_frame(kps, desc, frame)
            else:
                ok, R, t, pts, matches = initializer.initialize(kps, desc, matcher, frame)
                if ok:
                    state, n_map = "TRACKING", len(pts)
                    poses.append((R, t))
                    print("frame %d: initialised, %d map points, t = %s" % (idx, n_map, np.round(t.ravel(), 3)))
        elif not args.python_filters:
            ok, T, inl = geom.track_from_last_frame(last[0], last[1], kps, desc, K, frame.shape, ratio_threshold=mt["ratio_threshold"],
                                                    threshold_percent=0.02)   # tracker.py:219
            if ok:
                poses.append((T[:3, :3], T[:3, 3:4]))
                if idx % 5 == 0:
                    print("frame %d: %d pose inliers, t = %s" % (idx, len(inl), np.round(T[:3, 3], 3)))
        else:
            m = matcher.match(last[1], desc)
            m = matcher.filter_matches_by_geometric_distance(last[0], kps, m, 0.02, frame.shape)   # tracker.py:219-221
            m = matcher.filter_matches_by_distance(m)
            if len(m) >= 8:
                p1 = np.float32([last[0][x.queryIdx].pt for x in m])
                p2 = np.float32([kps[x.trainIdx].pt for x in m])
                E, mask = geom.calculate_essential_matrix(p1, p2, K, prob=0.999, threshold=1.0)   # tracker.py:242
                if E is not None:
                    n_in, R, t, _ = geom.recover_pose(E, p1, p2, K, mask)                          # tracker.py:249
                    poses.append((R, t))
                    if idx % 5 == 0:
                        print("frame %d: %d matches, %d pose inliers, t = %s" % (idx, len(m), n_in, np.round(t.ravel(), 3)))
        last = (kps, desc)
    dt = time.perf_counter() - t0
    print("%d frames in %.2f s (%.1f frames/s through the Python drop-in classes), state %s, %d poses"
          % (n_seen, dt, n_seen / max(dt, 1e-9), state, len(poses)))
    return state, poses, n_map


def run_batched(args, cfg, K, D, orb, mt, initializer, source, limit, skip, t0):
    """--batch N: the frame loop above with the extraction and the tracking step of EVERY frame taken from FrameStream (one batched
    device call per N frames).  Initialisation (the first frame pair that yields a map) goes through MapInitializer on the streamed
    keypoints / descriptors, exactly as Tracker does (tracker.py:162,168-170)."""
    import vslam_amd as V
    from orbslam2.types import KeyPointSeq
    from vslam_amd.stream import FrameStream

    def selected():
        seen = 0
        for idx, frame in enumerate(source):
            if seen >= limit:
                return
            if skip and idx % (skip + 1) != 0:
                continue
            seen += 1
            yield geom.undistort_image(frame, K, D) if np.any(D) else frame
    stack = None
    if args.frames and not os.path.isdir(args.frames) and not np.any(D) and not skip:   # a frame stack on disk: handed over as slices
        data = np.load(args.frames)   # (read into memory: a memory-mapped stack would page-fault inside the staging copies)
        arr = data[sorted(data.files)[0]] if hasattr(data, "files") else data
        stack = np.ascontiguousarray(arr[:limit] if limit < 10 ** 9 else arr, dtype=np.uint8)
    frames = iter(stack) if stack is not None else selected()
    first = next(frames, None)
    if first is None:
        return "NOT_INITIALIZED", [], 0
    h, w = first.shape[:2]
    prm = V.orb_params(nfeatures=orb["n_features"], scale_factor=orb["scale_factor"], nlevels=orb["n_levels"], fast_threshold=orb["min_threshold"])
    fs = FrameStream(K, width=w, height=h, channels=3 if first.ndim == 3 else 1, chunk=args.batch, prm=prm,
                     detector=V.DETECT_GRID if args.grid else V.DETECT_ORB, ratio=mt["ratio_threshold"], disp_frac=0.02, thr_px=1.0, copy=False)
    # (copy=False: the results are views of the stream's pinned buffers, valid while iterating; what this loop keeps - the keypoints and
    #  descriptors the initialiser holds on to, the poses - is copied explicitly.  A tracking frame only reads its pose: 12 doubles.)
    matcher = DescriptorMatcher(mt["matcher_type"], ratio_threshold=mt["ratio_threshold"])
    kept = {0: first}   # (frames the initialiser may still want: the first one, and the one being initialised against)

    def chain():
        yield first
        for i, f in enumerate(frames, 1):
            if state[0] == "NOT_INITIALIZED":
                kept[i] = f
            yield f
    feed = stack if stack is not None else chain()
    frame_at = (lambda i: stack[i]) if stack is not None else (lambda i: kept.get(i, first))
    state, poses, n_map, n_seen = ["NOT_INITIALIZED"], [], 0, 0
    try:
        t_steady, n_steady = None, 0
        for r in fs.run(feed):
            n_seen += 1
            if n_seen == 3 * args.batch + 1:     # (steady state: context, plan and the three lanes warmed up by the first chunks)
                t_steady, n_steady = time.perf_counter(), n_seen - 1
            if state[0] == "NOT_INITIALIZED":
                kps, desc = KeyPointSeq(r.keypoints.copy()), r.descriptors.copy()
                if initializer.first_frame_keypoints is None:
                    initializer.set_first_frame(kps, desc, frame_at(r.index))
                else:
                    ok, R, t, pts, matches = initializer.initialize(kps, desc, matcher, frame_at(r.index))
                    if ok:
                        state[0], n_map = "TRACKING", len(pts)
                        kept.clear()
                        poses.append((R, t))
                        print("frame %d: initialised, %d map points, t = %s" % (r.index, n_map, np.round(t.ravel(), 3)))
            else:
                p = r.pair
                if p is not None and p.ok:
                    poses.append((p.R.copy(), p.t.copy()))
                    if r.index % 100 == 0:
                        print("frame %d: %d pose inliers, t = %s" % (r.index, int(p.inlier.sum()), np.round(p.t.ravel(), 3)))
    finally:
        fs.close()
    t_end = time.perf_counter()
    dt = t_end - t0
    steady = "" if t_steady is None else "; %.0f frames/s from the fourth chunk on" % ((n_seen - n_steady) / max(t_end - t_steady, 1e-9))
    print("%d frames in %.2f s (%.1f frames/s through FrameStream incl. start-up, chunks of %d%s), state %s, %d poses"
          % (n_seen, dt, n_seen / max(dt, 1e-9), args.batch, steady, state[0], len(poses)))
    return state[0], poses, n_map


if __name__ == "__main__":
    main()
