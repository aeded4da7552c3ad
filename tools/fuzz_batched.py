#!/usr/bin/env python3
"""Randomised sweep of the device-resident batched call (mo_dev_frontend_batch) against the one-frame host API (itself checked
against the oracle by the test suite and tools/fuzz_parity.py): random frame sizes, batch sizes 2 - 13, both detectors, the three
pose modes, random nfeatures / cap (cap overflow must raise the flag, not fault), frames that are flat or nearly flat inside the
batch.  Keypoints, descriptors and match lists must be identical; poses must be sound (NaN or a rotation; point counts within the
number of correspondences).  Exit code 1 on any mismatch.
Usage (GPU box, repo root): python tools/fuzz_batched.py [--n 60] [--seed 1] [--budget-s 400]"""
import argparse
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, "visual-slam_amd"); sys.path.insert(0, ".")
import torch                               # noqa: E402
import vslam_amd as V                      # noqa: E402
from oracle import geom_oracle as G        # noqa: E402
from tests.helpers import synthetic_frame  # noqa: E402


def frames_of(rng, nb, w, h):
    step = int(rng.choice([0, 2, 8, 13]))
    wide = synthetic_frame(int(rng.integers(1, 10 ** 6)), w + step * nb + 4, h)
    out = np.stack([wide[:, step * i:step * i + w] for i in range(nb)]).copy()
    out = np.clip(out.astype(np.float32) + rng.normal(0, 2.0, out.shape), 0, 255).round().astype(np.uint8)
    for f in range(nb):   # a flat or nearly flat frame now and then
        r = rng.random()
        if r < 0.08:
            out[f] = int(rng.integers(0, 256))
        elif r < 0.14:
            keep = out[f, h // 2 - 20:h // 2 + 20, w // 2 - 20:w // 2 + 20].copy()
            out[f] = 90
            out[f, h // 2 - 20:h // 2 + 20, w // 2 - 20:w // 2 + 20] = keep
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=400.0)
    args = ap.parse_args(argv)
    rng = np.random.Generator(np.random.PCG64(args.seed))
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    bad, done, t0, tally = 0, 0, time.time(), {}
    host = V.Context(device=0, max_w=1100, max_h=800, max_batch=1)
    for it in range(args.n):
        if time.time() - t0 > args.budget_s:
            break
        w, h = int(rng.integers(128, 1100)), int(rng.integers(128, 800))
        if rng.random() < 0.3:
            w, h = 640, 480
        nb = int(rng.integers(2, 14))
        nfeat = int(rng.choice([64, 300, 1000, 2000, 3000]))
        detector = int(rng.integers(0, 2))
        mode = int(rng.integers(0, 3))
        cap = int(rng.choice([nfeat + 64, nfeat + 64, max(nfeat // 2, 32)]))   # the last one overflows on textured frames
        cfg = dict(it=it, w=w, h=h, nb=nb, nfeat=nfeat, detector=detector, mode=mode, cap=cap)
        frames = frames_of(rng, nb, w, h)
        K = np.array([[0.5 * w, 0, 0.5 * w], [0, 0.5 * w, 0.5 * h], [0, 0, 1.0]])
        ctx = V.Context(device=0, max_w=w, max_h=h, max_batch=nb)
        ctx.set_stream(st.cuda_stream)
        prm = V.orb_params(nfeatures=nfeat)
        z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)
        d_fr = torch.from_numpy(frames).to(dev)
        npair = 3 if mode == V.MODE_KEYFRAME else nb - 1
        b = dict(kps=z(nb, cap, 7, dt=torch.float32), desc=z(nb, cap, 32, dt=torch.uint8), counts=z(nb),
                 midx=torch.full((max(npair, 1), cap, 2), -7, dtype=torch.int32, device=dev), mdist=z(max(npair, 1), cap, 2),
                 mpass=z(max(npair, 1), cap, dt=torch.uint8), pose=z(max(npair, 1), 12, dt=torch.float64),
                 pts=z(max(npair, 1), cap, 3, dt=torch.float32), npts=z(max(npair, 1)), pmask=z(max(npair, 1), cap, dt=torch.uint8))
        io = V.BatchIO()
        io.d_gray = d_fr.data_ptr(); io.w = w; io.h = h; io.batch = nb; io.cap = cap
        io.ratio = 0.75; io.thr_px = 3.0; io.n_hyp = 256; io.seed = 7; io.pair_index_base = 0
        for i in range(9): io.K[i] = float(K.reshape(9)[i])
        io.d_kps = b["kps"].data_ptr(); io.d_desc = b["desc"].data_ptr(); io.d_counts = b["counts"].data_ptr()
        io.d_match_idx = b["midx"].data_ptr(); io.d_match_dist = b["mdist"].data_ptr(); io.d_match_pass = b["mpass"].data_ptr()
        io.d_pose = b["pose"].data_ptr(); io.d_points = b["pts"].data_ptr(); io.d_n_points = b["npts"].data_ptr()
        io.d_pose_mask = b["pmask"].data_ptr()
        keep = []
        if detector == V.DETECT_GRID:
            per = max(nfeat // 64, 1)
            gxy = z(nb, 64 * per, 2, dt=torch.float32); gn = z(nb, 66); gk = torch.full((nb, cap), -1, dtype=torch.int32, device=dev)
            io.detector = V.DETECT_GRID; io.d_grid_xy = gxy.data_ptr(); io.d_grid_n = gn.data_ptr(); io.d_grid_kept = gk.data_ptr()
            keep += [gxy, gn, gk]
        pairs = [(i, i + 1) for i in range(nb - 1)]
        if mode == V.MODE_TRACK:
            sel = z(nb - 1, cap, 2); seln = z(nb - 1)
            io.mode = V.MODE_TRACK; io.disp_frac = 0.02; io.thr_px = 1.0; io.d_sel_idx = sel.data_ptr(); io.d_sel_n = seln.data_ptr()
            keep += [sel, seln]
        elif mode == V.MODE_KEYFRAME:
            pairs = [(int(rng.integers(0, nb)), int(rng.integers(0, nb))) for _ in range(3)]
            q = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev); t = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
            P = np.zeros((3, 2, 3, 4)); P[:, :, :, :3] = K
            for j, (a, c2) in enumerate(pairs):
                P[j, 0, :, 3] = K @ np.array([-0.05 * a, 0, 0]); P[j, 1, :, 3] = K @ np.array([-0.05 * c2, 0, 0])
            dP1 = torch.from_numpy(np.ascontiguousarray(P[:, 0].reshape(-1, 12))).to(dev); dP2 = torch.from_numpy(np.ascontiguousarray(P[:, 1].reshape(-1, 12))).to(dev)
            dF = z(3, 9, dt=torch.float64)
            io.mode = V.MODE_KEYFRAME; io.ratio = 0.8; io.n_kf_pairs = 3; io.d_kf_query = q.data_ptr(); io.d_kf_train = t.data_ptr()
            io.d_kf_P1 = dP1.data_ptr(); io.d_kf_P2 = dP2.data_ptr(); io.d_kf_F = dF.data_ptr()
            keep += [q, t, dP1, dP2, dF]
        ok = True
        try:
            ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
            st.synchronize()
            flags = ctx.dev_status()
        except V.NativeError as e:
            tally["refused"] = tally.get("refused", 0) + 1
            print("refused", cfg, str(e)[:120], flush=True)
            ctx.close()
            continue
        cn = b["counts"].cpu().numpy()
        feats = []
        for f in range(nb):
            if detector == V.DETECT_GRID:
                xy, kept, d = host.grid_detect_compute(frames[f], prm, nfeat)
                k = np.zeros(len(kept), V.KP_DTYPE)
                k["x"], k["y"], k["size"], k["angle"], k["class_id"] = xy[kept, 0], xy[kept, 1], 31, -1, -1
            else:
                k, d = host.orb_detect_compute(frames[f], prm)[0]
            d = np.zeros((0, 32), np.uint8) if d is None else d
            feats.append((k, d))
        over = any(len(k) > cap for k, _ in feats)
        if bool(flags & 2) != over:
            ok = False
            print("MISMATCH capacity flag %d, overflow expected %s" % (flags, over), cfg, flush=True)
        if flags & ~2:
            ok = False
            print("MISMATCH unexpected flag %d" % flags, cfg, flush=True)
        kp_np = b["kps"].cpu().numpy().view(np.uint8).reshape(nb, cap, 28)
        for f in range(nb):
            k, d = feats[f]
            n = min(len(k), cap)
            if cn[f] != len(k) or not np.array_equal(kp_np[f, :n].reshape(-1).view(V.KP_DTYPE), k[:n]) or \
               not np.array_equal(b["desc"][f, :n].cpu().numpy(), d[:n]):
                ok = False
                print("MISMATCH frame %d: count %d vs %d" % (f, cn[f], len(k)), cfg, flush=True)
        if not over:   # (with truncated rows the match lists are those of the truncated sets: compared against exactly those)
            MI = b["midx"].cpu().numpy(); MD = b["mdist"].cpu().numpy(); MP = b["mpass"].cpu().numpy().astype(bool)
            P = b["pose"].cpu().numpy(); NP = b["npts"].cpu().numpy()
            for j, (qa, ta) in enumerate(pairs):
                nq, nt = len(feats[qa][0]), len(feats[ta][0])
                if nq == 0 or nt == 0:
                    if MP[j].any():
                        ok = False
                        print("MISMATCH pair %d: passes with an empty side" % j, cfg, flush=True)
                    continue
                idx, dist, ps = host.match_knn2_ratio(feats[qa][1], feats[ta][1], float(io.ratio))
                if not (np.array_equal(MI[j, :nq], idx) and np.array_equal(MD[j, :nq], dist) and np.array_equal(MP[j, :nq], ps)) or MP[j, nq:].any():
                    ok = False
                    print("MISMATCH pair %d (%d, %d): match lists" % (j, qa, ta), cfg, flush=True)
                if mode == V.MODE_TRACK:   # the kept list of the two tracking filters, in the reference's order: integer work, exact
                    eq, et, _ = G.track_select(np.stack([feats[qa][0]["x"], feats[qa][0]["y"]], 1), np.stack([feats[ta][0]["x"], feats[ta][0]["y"]], 1),
                                               idx, dist, ps, w, h, 0.02)
                    ns = int(keep[-1][j].item())
                    got = keep[-2][j, :ns].cpu().numpy()
                    if ns != len(eq) or not np.array_equal(got[:, 0], eq) or not np.array_equal(got[:, 1], et):
                        ok = False
                        print("MISMATCH pair %d: tracking filter list %d vs %d" % (j, ns, len(eq)), cfg, flush=True)
                if mode != V.MODE_KEYFRAME:
                    R = P[j, :9].reshape(3, 3)
                    sound = np.isnan(P[j]).all() or (np.allclose(R @ R.T, np.eye(3), atol=1e-8) and abs(np.linalg.det(R) - 1) < 1e-8
                                                     and abs(np.linalg.norm(P[j, 9:]) - 1) < 1e-8)
                    if not sound or not 0 <= NP[j] <= ps.sum():
                        ok = False
                        print("MISMATCH pair %d: unsound pose / count %d of %d" % (j, NP[j], ps.sum()), cfg, P[j], flush=True)
                elif not 0 <= NP[j] <= ps.sum():
                    ok = False
                    print("MISMATCH keyframe pair %d: %d points of %d matches" % (j, NP[j], ps.sum()), cfg, flush=True)
        key = "det%d mode%d" % (detector, mode)
        tally[key] = tally.get(key, 0) + 1
        bad += 0 if ok else 1
        done += 1
        ctx.close()
        del b, d_fr, keep
        if it % 5 == 0:
            print("... %d configurations, %d bad, %.0f s" % (done, bad, time.time() - t0), flush=True)
    print("fuzz_batched: %d configurations checked, %d with a mismatch, %.0f s; %s" % (done, bad, time.time() - t0, tally), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
