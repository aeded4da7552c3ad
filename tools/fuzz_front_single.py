#!/usr/bin/env python3
"""Randomised sweep of the single-frame pyramid + blur kernel (csrc/front_single.hip) against the batched path's kernels (k_resize2 /
k_resize / k_blur, which tests/test_gpu_orb.py and tools/fuzz_parity.py hold against the CPU oracle): random frame sizes, scale factors
and level counts, every raw and blurred level byte for byte through mo_dbg_pyramid_level.  Random sizes give tile layouts, pad-column
cases and fall-back decisions the fixed geometries of tests/test_gpu_front_single.py do not.  Prints every mismatch with its
configuration, the number of geometries the kernel does not cover (they keep the batched kernels), exit code 1 on any mismatch.
Usage (GPU box, repo root): python tools/fuzz_front_single.py [--n 400] [--seed 1] [--budget-s 200]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import vslam_amd as V  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=200.0)
    a = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(a.seed))
    ctx = V.Context(device=0, max_w=2048, max_h=2048, max_batch=2)
    t0 = time.time()
    done = bad = uncovered = refused = 0
    why = {}
    near = [0, 0]  # parameters near the reference's (scale <= 1.3, <= 8 levels): [taken by the kernel, not taken]
    levels_checked = 0
    for it in range(a.n):
        if time.time() - t0 > a.budget_s:
            break
        if it % 4 == 0:    # sizes around the common ones, odd widths included
            w, h = int(rng.choice([320, 640, 752, 848, 1024, 1280, 1920])) + int(rng.integers(-3, 4)), int(rng.choice([240, 480, 600, 720, 1080])) + int(rng.integers(-3, 4))
        else:
            w, h = int(rng.integers(64, 2049)), int(rng.integers(64, 1300))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.25, 1.3, 1.41, 1.5, 1.7, 2.0]))
        nl = int(rng.integers(2, 13))
        cfg = dict(it=it, w=w, h=h, scale_factor=sf, nlevels=nl)
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        if rng.integers(0, 3) == 0:  # smooth content: errors of one grey level would hide in noise less often
            img = (np.add.outer(np.arange(h) * 3, np.arange(w) * 5) % 256).astype(np.uint8)
        p = V.orb_params(select_order=V.ORDER_LIBSTDCXX, scale_factor=sf, nlevels=nl)
        try:
            ref = [(ctx.dbg_pyramid_level(img, p, L, blurred=0), ctx.dbg_pyramid_level(img, p, L, blurred=1)) for L in range(nl)]
        except V.NativeError as e:  # e.g. a level that collapses to zero size: the plan itself is refused
            refused += 1
            continue
        try:
            got = [(ctx.dbg_pyramid_level(img, p, L, blurred=2), ctx.dbg_pyramid_level(img, p, L, blurred=3)) for L in range(nl)]
        except V.NativeError as e:
            if "does not cover" not in str(e):
                print("error", cfg, e, flush=True)
                bad += 1
            uncovered += 1
            if sf <= 1.3 and nl <= 8:
                near[1] += 1
                print("not covered near the reference's parameters:", cfg, str(e).split("geometry: ")[-1], flush=True)
            r = str(e).split("geometry: ")[-1]
            why[r] = why.get(r, 0) + 1
            continue
        ok = True
        for L in range(nl):
            if L > 0 and not np.array_equal(got[L][0], ref[L][0]):
                print("MISMATCH raw level", L, int((got[L][0] != ref[L][0]).sum()), "pixels", cfg, flush=True)
                ok = False
            if not np.array_equal(got[L][1], ref[L][1]):
                print("MISMATCH blurred level", L, int((got[L][1] != ref[L][1]).sum()), "pixels", cfg, flush=True)
                ok = False
            levels_checked += 2
        done += 1
        if sf <= 1.3 and nl <= 8:
            near[0] += 1
        bad += 0 if ok else 1
    print("fuzz_front_single: %d configurations compared (%d levels), %d with a mismatch, %d not covered by the kernel (batched kernels kept), %d plans refused, %.0f s"
          % (done, levels_checked, bad, uncovered, refused, time.time() - t0), flush=True)
    print("scale <= 1.3 and <= 8 levels (the reference runs 1.2 x 8): %d taken by the kernel, %d not" % tuple(near), flush=True)
    print("not covered, by reason:", dict(sorted(why.items(), key=lambda kv: -kv[1])), flush=True)
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
