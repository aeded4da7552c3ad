#!/usr/bin/env python3
"""Batch-1 latency breakdown of the calls the reference's Tracker makes per frame (BASELINE config 2; reference
src/tests/tester_map.py:57-67, src/orbslam2/tracker.py:87,198-266): one frame in, host arrays / Python objects out.

For each call: median wall ms through the drop-in class, through the ctypes binding (numpy arrays), inside the C call
(mo_host_times: enqueue / wait / unpack) and on the device (mo_stage_times: h2d, every kernel stage, d2h).
    python tools/single_frame_probe.py [--json out.json] [--iters 50]
Run under `rocprofv3 --kernel-trace --stats` for the kernel timeline (profiles/summarize_timeline.py reads the trace)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import vslam_amd as V  # noqa: E402


def med(fn, n, after=None):
    fn(); fn()
    ts, extra = [], []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
        if after:
            extra.append(after())
    return float(np.median(ts)), extra


def measure(ctx, fn, n):
    """wall and host clocks with the stage events OFF (the product's default: an event between two kernels idles the GPU ~ 4.5 us), then
    the device spans of the same call with the events ON"""
    ctx.set_host_timing(False)
    wall_ms, extra = med(fn, n, lambda: ctx.host_times())
    host = {k: round(float(np.median([e[k] for e in extra])) / 1e3, 4) for k in extra[0]}
    ctx.set_host_timing(True)
    wall_t, ev = med(fn, max(10, n // 2), lambda: ctx.stage_times())
    ctx.set_host_timing(False)
    names = [nm for nm, _ in ev[0]]
    dev = {nm: round(float(np.median([dict(e)[nm] for e in ev])), 4) for nm in names}
    dev_total = round(sum(dev.values()), 4)
    return {"wall_ms": round(wall_ms, 4), "binding_python_ms": round(wall_ms - host["total_us"], 4),
            "c_call_ms": host["total_us"], "c_enqueue_ms": host["enqueue_us"], "c_wait_ms": host["wait_us"], "c_unpack_ms": host["unpack_us"],
            "device_ms_with_events": dev, "device_total_ms_with_events": dev_total, "wall_ms_with_events": round(wall_t, 4),
            "note": "wall / c_* : stage events off (default); device_ms_with_events: hipEvent spans of the same call with mo_set_host_timing(1), "
                    "which adds ~ 4.5 us of idle GPU per event to the call"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--scene", default="survey8d")
    args = ap.parse_args()
    from vslam_amd import synth
    import torch
    fr = synth.make_frames(torch, torch.device("cuda", 0), 0, 2, scene=args.scene).cpu().numpy()
    f0, f1 = np.ascontiguousarray(fr[0]), np.ascontiguousarray(fr[1])
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
    from orbslam2 import utils as geom
    from orbslam2.extractor import ORBExtractor
    from orbslam2.initializer import MapInitializer
    from orbslam2.matcher import DescriptorMatcher
    ex = ORBExtractor(n_features=2000)
    mt = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)
    ctx = V.default_context()
    prm = ex.orb.prm
    n = args.iters
    out = {}
    (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)

    out["detect_and_compute"] = measure(ctx, lambda: ctx.orb_detect_compute(f0, prm), n)
    w = out["detect_and_compute"]["wall_ms"]
    out["detect_and_compute"]["class_wall_ms"] = round(med(lambda: ex.detect_and_compute(f0), n)[0], 4)
    out["detect_and_compute"]["class_python_ms"] = round(out["detect_and_compute"]["class_wall_ms"] - w, 4)

    (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)   # (the last two extractions: both resident)
    out["match_resident"] = measure(ctx, lambda: ctx.match_knn2_ratio(d0, d1, 0.75), n)
    d0c, d1c = d0.copy(), d1.copy()   # (copies are not the resident arrays: the upload path)
    out["match"] = measure(ctx, lambda: ctx.match_knn2_ratio(d0c, d1c, 0.75), n)
    w = out["match_resident"]["wall_ms"]
    out["match"]["class_wall_ms"] = round(med(lambda: mt.match(d0, d1), n)[0], 4)
    out["match"]["class_python_ms"] = round(out["match"]["class_wall_ms"] - w, 4)

    ka0, ka1 = k0.array, k1.array
    out["track_pair_resident"] = measure(ctx, lambda: ctx.track_pair(ka0, d0, ka1, d1, 640, 480, K), n)
    kc0, kc1 = ka0.copy(), ka1.copy()
    out["track_pair"] = measure(ctx, lambda: ctx.track_pair(kc0, d0c, kc1, d1c, 640, 480, K), n)
    w = out["track_pair_resident"]["wall_ms"]
    out["track_pair"]["class_wall_ms"] = round(med(lambda: geom.track_from_last_frame(k0, d0, k1, d1, K, f1.shape), n)[0], 4)
    out["track_pair"]["class_python_ms"] = round(out["track_pair"]["class_wall_ms"] - w, 4)

    # a Tracker in TRACKING state: every frame is extracted once and tracked against the previous one (both resident by then)
    state = {"last": ex.detect_and_compute(f0), "i": 0}

    def tracker_frame():
        state["i"] ^= 1
        cur = ex.detect_and_compute(f1 if state["i"] else f0)
        r = geom.track_from_last_frame(state["last"][0], state["last"][1], cur[0], cur[1], K, f1.shape)
        state["last"] = cur
        return r
    out["tracker_frame_class_wall_ms"] = round(med(tracker_frame, n)[0], 4)
    (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)
    out["extract_features_distributed_class_wall_ms"] = round(med(lambda: ex.extract_features(f0, distributed=True), n)[0], 4)
    # the detector Tracker.process_frame takes by default (reference tracker.py:87: extract_features() with distributed=True): the same
    # breakdown at the binding, and a tracked frame on it (grid extraction + track_from_last_frame on the resident records)
    out["grid_detect_compute"] = measure(ctx, lambda: ctx.grid_detect_compute(f0, prm, 2000, records=True), n)
    out["grid_detect_compute"]["class_wall_ms"] = out["extract_features_distributed_class_wall_ms"]
    out["grid_detect_compute"]["class_python_ms"] = round(out["extract_features_distributed_class_wall_ms"] - out["grid_detect_compute"]["wall_ms"], 4)
    gstate = {"last": ex.distribute_keypoints(f0, aligned=True), "i": 0}  # (aligned: keypoint i belongs to descriptor row i, records resident)

    def tracker_frame_grid():
        gstate["i"] ^= 1
        cur = ex.distribute_keypoints(f1 if gstate["i"] else f0, aligned=True)
        r = geom.track_from_last_frame(gstate["last"][0], gstate["last"][1], cur[0], cur[1], K, f1.shape)
        gstate["last"] = cur
        return r
    out["tracker_frame_grid_class_wall_ms"] = round(med(tracker_frame_grid, n)[0], 4)
    (k0, d0), (k1, d1) = ex.detect_and_compute(f0), ex.detect_and_compute(f1)

    init = MapInitializer(K)
    init.set_first_frame(k0, d0, f0)
    import contextlib, io
    def run_init():
        with contextlib.redirect_stdout(io.StringIO()):
            return init.initialize(k1, d1, mt, f1)
    out["initialize_class_wall_ms"] = round(med(run_init, max(10, n // 5))[0], 4)
    out["keypoints"] = [len(k0), len(k1)]
    print(json.dumps(out, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
