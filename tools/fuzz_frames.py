#!/usr/bin/env python3
"""Randomised sweep of the one-frame-at-a-time and streamed forms (round 4) against each other and the batched kernels they share:
random frame sizes, feature counts, detectors, chunk sizes and sequence lengths;
  * FrameStream (mo_stream) == the per-frame loop (single-frame extraction + mo_pair_frontend with pair_index = i): keypoints, descriptors,
    kept tracking matches, poses, inlier masks - bit for bit;
  * the pair step by tokens (resident frames) == by host arrays (uploaded) == after the tokens went stale;
  * MODE_INIT of the pair step == matcher call + explicit-point two-view call.
Every mismatch is printed with its configuration; exit code 1 if any.
Usage (GPU box, repo root): python tools/fuzz_frames.py [--n 60] [--seed 1] [--budget-s 300]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import vslam_amd as V                          # noqa: E402
from vslam_amd.stream import FrameStream       # noqa: E402
from tests.helpers import parallax_frames      # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=300.0)
    args = ap.parse_args(argv)
    rng = np.random.Generator(np.random.PCG64(args.seed))
    bad, done, t0 = 0, 0, time.time()
    for it in range(args.n):
        if time.time() - t0 > args.budget_s:
            break
        w, h = int(rng.integers(20, 120)) * 8, int(rng.integers(16, 90)) * 8
        nfeat = int(rng.choice([64, 300, 1000, 2000, 3000]))
        grid = bool(rng.integers(0, 2)) and w >= 160 and h >= 160
        chunk, nfr = int(rng.integers(1, 10)), int(rng.integers(2, 15))
        n_hyp = int(rng.choice([64, 256, 512, 1024]))
        copy = bool(rng.integers(0, 2))
        cfg = dict(it=it, w=w, h=h, nfeat=nfeat, grid=grid, chunk=chunk, nfr=nfr, n_hyp=n_hyp, copy=copy)
        frames = parallax_frames(nfr, seed=int(rng.integers(1, 10 ** 6)), w=w, h=h, bg_step=int(rng.integers(1, 4)), fg_step=int(rng.integers(3, 8)))
        frames = np.clip(frames.astype(np.float32) + rng.normal(0, float(rng.choice([0.0, 1.0, 3.0])), frames.shape), 0, 255).round().astype(np.uint8)
        K = np.array([[0.5 * w, 0, 0.5 * w], [0, 0.5 * w, 0.5 * h], [0, 0, 1.0]])
        prm = V.orb_params(nfeatures=nfeat)
        ok = True
        try:
            fs = FrameStream(K, width=w, height=h, chunk=chunk, prm=prm, detector=V.DETECT_GRID if grid else V.DETECT_ORB, n_hyp=n_hyp, copy=copy)
            try:
                got = []
                for r in fs.run(iter(frames) if rng.integers(0, 2) else frames):
                    p = r.pair
                    got.append((np.array(r.keypoints), np.array(r.descriptors),
                                None if p is None else dict(ok=p.ok, sel=np.array(p.sel), sel_dist=np.array(p.sel_dist), inlier=np.array(p.inlier),
                                                            R=np.array(p.R), t=np.array(p.t), n_inliers=p.n_inliers, pair_index=p.pair_index)))
            finally:
                fs.close()
        except V.NativeError as e:
            print("refused", cfg, str(e)[:120], flush=True)
            continue
        ctx = V.Context(device=0, max_w=w, max_h=h, max_batch=1)
        last = None
        for i in range(nfr):
            if grid:
                _, kept, d, k = ctx.grid_detect_compute(frames[i], prm, nfeat, records=True)
            else:
                (k, d), = ctx.orb_detect_compute(frames[i], prm)
                if d is None:
                    d = np.zeros((0, 32), np.uint8)
            gk, gd, gp = got[i]
            if not (np.array_equal(gk, k) and np.array_equal(gd, d)):
                ok = False
                print("MISMATCH features frame %d: %d vs %d" % (i, len(gk), len(k)), cfg, flush=True)
            if last is not None:
                route = int(rng.integers(0, 3))   # 0 tokens (resident), 1 copies (uploaded), 2 stale tokens
                a = (last[0], last[1], k, d) if route != 1 else (last[0].copy(), last[1].copy(), k.copy(), d.copy())
                if route == 2:
                    for _ in range(4):
                        ctx.orb_detect_compute(frames[0], V.orb_params(nfeatures=64))
                if len(a[0]) and len(a[2]):
                    r = ctx.track_pair(a[0], a[1], a[2], a[3], w, h, K, n_hyp=n_hyp, pair_index=i - 1)
                    rok = len(r["sel"]) >= 8 and bool(np.isfinite(r["R"]).all())
                    same = gp["pair_index"] == i - 1 and np.array_equal(gp["sel"], r["sel"]) and np.array_equal(gp["sel_dist"], r["sel_dist"]) and gp["ok"] == rok
                    if same and rok:
                        same = np.array_equal(gp["R"], r["R"]) and np.array_equal(gp["t"], r["t"]) and np.array_equal(gp["inlier"], r["inlier"]) and gp["n_inliers"] == r["n_inliers"]
                    if not same:
                        ok = False
                        print("MISMATCH track pair %d (route %d): kept %d vs %d" % (i - 1, route, len(gp["sel"]), len(r["sel"])), cfg, flush=True)
                    if rng.integers(0, 3) == 0:   # initialisation step: fused == matcher + explicit two-view call
                        f = ctx.pair_frontend(a[0], a[1], a[2], a[3], V.MODE_INIT, K, ratio=0.75, thr_px=3.0, n_hyp=n_hyp)
                        idx, dist, keep = ctx.match_knn2_ratio(np.array(a[1]), np.array(a[3]), 0.75)
                        q = np.flatnonzero(keep)
                        e = ctx.init_two_view(np.stack([a[0]["x"], a[0]["y"]], 1)[q], np.stack([a[2]["x"], a[2]["y"]], 1)[idx[q, 0]], K, thr_px=3.0, n_hyp=n_hyp)
                        eq = np.array_equal(f["idx"], idx) and np.array_equal(f["keep"], keep) and f["n_good"] == e["n_good"]
                        if eq and len(q) >= 8 and np.isfinite(e["R"]).all():
                            eq = np.allclose(f["R"], e["R"], atol=1e-12) and np.array_equal(f["pose_mask"][q], e["pose_mask"])
                        if not eq:
                            ok = False
                            print("MISMATCH init pair %d: n_good %d vs %d" % (i - 1, f["n_good"], e["n_good"]), cfg, flush=True)
            last = (k, d)
        ctx.close()
        bad += 0 if ok else 1
        done += 1
        if it % 5 == 0:
            print("... %d configurations, %d bad, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("fuzz_frames: %d configurations checked, %d with a mismatch, %.0f s" % (done, bad, time.time() - t0), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
