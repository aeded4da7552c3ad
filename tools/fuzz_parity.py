#!/usr/bin/env python3
"""Randomised parity sweep of the extractor against the CPU oracle (a search for latent defects, not part of the test suite):
random image sizes, scenes, ORB parameters and both STL orders through the host API (detect_and_compute, batch of 1 - 5 frames),
the grid detector, compute() with caller keypoints, and the matcher on the descriptors found.  Every mismatch is printed with the
configuration that produced it; exit code 1 if any.
Usage (GPU box, repo root): python tools/fuzz_parity.py [--n 150] [--seed 1] [--budget-s 400]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import vslam_amd as V                      # noqa: E402
from oracle import orb_oracle as O         # noqa: E402
from tests.helpers import synthetic_frame  # noqa: E402


def scene(rng, w, h):
    kind = rng.integers(0, 6)
    if kind == 0:
        return "texture", synthetic_frame(int(rng.integers(1, 10 ** 6)), w, h)
    if kind == 1:
        return "noise", rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == 2:  # few blobs on a flat ground: sparse keypoints, empty levels
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(0, 12))):
            x, y = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
            img[y:y + int(rng.integers(2, 30)), x:x + int(rng.integers(2, 30))] = int(rng.integers(0, 256))
        return "blobs", img
    if kind == 3:  # saturated checkerboard: ties in every score
        c = int(rng.integers(3, 17))
        yy, xx = np.mgrid[0:h, 0:w]
        return "checker%d" % c, (((yy // c + xx // c) & 1) * 255).astype(np.uint8)
    if kind == 4:  # low-contrast texture: few corners above the threshold
        t = synthetic_frame(int(rng.integers(1, 10 ** 6)), w, h).astype(np.float32)
        return "lowcontrast", np.clip(110 + (t - 128) * 0.12, 0, 255).astype(np.uint8)
    g = np.linspace(0, 255, w)[None, :] * np.ones((h, 1))
    return "gradient+noise", np.clip(g + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)


def same_kps(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in a.dtype.names)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=400.0)
    ap.add_argument("--max-batch", type=int, default=5, help="frames the context is created for; <= 2: the plan of single-frame contexts "
                    "(2-row FAST strips) and batches of 1 .. max-batch frames only")
    args = ap.parse_args(argv)
    rng = np.random.Generator(np.random.PCG64(args.seed))
    ctx = V.Context(device=0, max_w=1000, max_h=800, max_batch=args.max_batch)
    bad, done, t0 = 0, 0, time.time()
    tally = {}
    for it in range(args.n):
        if time.time() - t0 > args.budget_s:
            break
        w, h = int(rng.integers(64, 1000)), int(rng.integers(64, 800))
        kw = dict(nfeatures=int(rng.choice([1, 7, 50, 200, 500, 1000, 2000, 3000, 6000])),
                  scale_factor=float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.33, 1.5, 2.0])),
                  nlevels=int(rng.integers(1, 9)), fast_threshold=int(rng.choice([0, 1, 5, 7, 7, 20, 40, 100])),
                  edge_threshold=int(rng.choice([31, 31, 31, 19, 25, 40])))
        order = int(rng.integers(0, 2))
        nb = min(int(rng.choice([1, 1, 1, 2, 3, 5])), args.max_batch)
        names, imgs = zip(*[scene(rng, w, h) for _ in range(nb)])
        cfg = dict(it=it, w=w, h=h, order=order, scenes=names, **kw)
        O.lib().orc_set_variant(order, 0)
        try:
            prm = V.orb_params(select_order=order, **kw)
            oprm = O.params(**kw)
            res = ctx.orb_detect_compute(np.stack(imgs), prm) if nb > 1 else ctx.orb_detect_compute(imgs[0], prm)
        except V.NativeError as e:   # a refusal is fine (too small a level, ...); a wrong answer is not
            tally["refused"] = tally.get("refused", 0) + 1
            print("refused", cfg, str(e)[:100], flush=True)
            continue
        ok = True
        descs = []
        for f in range(nb):
            ek, ed = O.detect_and_compute(imgs[f], oprm)
            kps, desc = res[f]
            if not same_kps(kps, ek.astype(kps.dtype)) or (len(ek) and not np.array_equal(desc, ed)) or (not len(ek) and desc is not None):
                ok = False
                print("MISMATCH detect_and_compute frame %d: %d vs %d keypoints" % (f, len(kps), len(ek)), cfg, flush=True)
            descs.append(ed if len(ek) else None)
            tally[names[f]] = tally.get(names[f], 0) + 1
        # compute() on a random subset of the oracle's keypoints plus a few arbitrary ones (some near the border)
        ek, _ = O.detect_and_compute(imgs[0], oprm, want_desc=False)
        kin = ek[rng.permutation(len(ek))[:int(rng.integers(0, 200))]].copy() if len(ek) else np.zeros(0, V.KP_DTYPE)
        extra = np.zeros(int(rng.integers(0, 20)), V.KP_DTYPE)
        extra["x"] = rng.uniform(0, w, len(extra)); extra["y"] = rng.uniform(0, h, len(extra)); extra["size"] = 31
        extra["angle"] = rng.choice([-1.0, 0.0, 123.4], len(extra)); extra["octave"] = rng.integers(0, kw["nlevels"], len(extra)); extra["class_id"] = -1
        kin = np.concatenate([kin.astype(V.KP_DTYPE), extra])
        if len(kin):
            kept, d = ctx.orb_compute(imgs[0], prm, kin)
            ekept, edc = O.compute(imgs[0], oprm, kin)
            if not np.array_equal(kept, ekept) or (len(ekept) and not np.array_equal(d, edc)):
                ok = False
                print("MISMATCH compute: kept %d vs %d" % (len(kept), len(ekept)), cfg, flush=True)
        # grid detector (640 x 480-like and odd sizes alike; the fused call picks its own path)
        if w >= 128 and h >= 128 and rng.integers(0, 2):
            nf = int(rng.choice([64, 500, 2000, 4000]))
            O.lib().orc_set_variant(0, 0)
            exy = O.grid_good_features(imgs[0], nf)
            try:
                xy, kept, d = ctx.grid_detect_compute(imgs[0], V.orb_params(nfeatures=nf), nf)
            except V.NativeError as e:   # documented limit: > 2048 local maxima in one cell (plateaus of a synthetic pattern)
                tally["grid refused"] = tally.get("grid refused", 0) + 1
                print("grid refused", cfg, str(e)[:100], flush=True)
                xy = None
            if xy is None:
                pass
            elif not np.array_equal(xy, exy):
                ok = False
                print("MISMATCH grid corners: %d vs %d" % (len(xy), len(exy)), cfg, "nf", nf, flush=True)
            else:
                k = np.zeros(len(exy), V.KP_DTYPE)
                k["x"], k["y"], k["size"], k["angle"], k["class_id"] = exy[:, 0], exy[:, 1], 31, -1, -1
                ekept, edc = O.compute(imgs[0], O.params(nfeatures=nf), k)
                if not np.array_equal(kept, ekept) or (len(ekept) and not np.array_equal(d, edc)):
                    ok = False
                    print("MISMATCH grid compute: kept %d vs %d" % (len(kept), len(ekept)), cfg, "nf", nf, flush=True)
        # matcher on what was found (ragged sizes, ties on the checkerboards)
        have = [d for d in descs if d is not None and len(d)]
        if len(have) >= 1:
            q, t = have[0], have[-1]
            ratio = float(rng.choice([0.5, 0.75, 0.9]))
            idx, dist, ps = ctx.match_knn2_ratio(q, t, ratio)
            eidx, edist = O.match_knn2(q, t)
            if not (np.array_equal(idx, eidx) and np.array_equal(dist, edist) and np.array_equal(ps, O.ratio_test(eidx, edist, ratio))):
                ok = False
                print("MISMATCH match %d x %d" % (len(q), len(t)), cfg, flush=True)
        bad += 0 if ok else 1
        done += 1
        if it % 10 == 0:
            print("... %d configurations, %d bad, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("fuzz: %d configurations checked, %d with a mismatch, %.0f s; scenes %s" % (done, bad, time.time() - t0, tally), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
