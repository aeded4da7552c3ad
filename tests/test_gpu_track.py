"""GPU parity of the tracking step (reference tracker.py:214-254): match -> displacement filter -> 2 x median distance filter ->
E-RANSAC at 1 px -> pose.  The kept match list is bit-exact against the oracle's restatement of the two Python filters
(matcher.py:109-169: Python-float hypot, stable sort, np.median); R, t within 1e-4 of the oracle run on that list."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import parallax_frames, synthetic_frame

pytestmark = pytest.mark.gpu
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])


def _xy(k):
    return np.stack([k["x"], k["y"]], 1)


def _oracle_track(O, G, fa, fb, ratio, frac, pair=0, n_hyp=1024):
    O.lib().orc_set_variant(0, 0)
    idx, dist = O.match_knn2(fa[1], fb[1])
    keep = O.ratio_test(idx, dist, ratio)
    return G.track_pair(_xy(fa[0]), _xy(fb[0]), idx, dist, keep, K, 640, 480, frac=frac, thr_px=1.0, n_hyp=n_hyp, seed=4096, pair=pair)


# frac 0.006 (3.4 px) only lets the background layer through: a single plane under pure translation, for which the essential
# matrix is degenerate - there only the filtered list is compared
@pytest.mark.parametrize("frac,ratio,check_pose", [(0.02, 0.75, True), (0.02, 0.9, True), (0.006, 0.75, False)])
def test_track_pair_host_api_vs_oracle(frac, ratio, check_pose):
    import vslam_amd as V
    from oracle import geom_oracle as G
    from oracle import orb_oracle as O
    frames = parallax_frames(3, seed=77, bg_step=3, fg_step=6)
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=2)
    prm = V.orb_params(nfeatures=2000)
    fa, fb = ctx.orb_detect_compute(frames[:2], prm)
    o = _oracle_track(O, G, fa, fb, ratio, frac)
    g = ctx.track_pair(fa[0], fa[1], fb[0], fb[1], 640, 480, K, ratio=ratio, disp_frac=frac, thr_px=1.0, n_hyp=1024, seed=4096)
    assert len(o["sel_q"]) > 100
    assert np.array_equal(g["sel"][:, 0], o["sel_q"]) and np.array_equal(g["sel"][:, 1], o["sel_t"])
    assert np.array_equal(g["sel_dist"], o["sel_d"])
    assert np.all(np.diff(g["sel_dist"]) >= 0)  # ascending distance
    if not check_pose:
        ctx.close()
        return
    assert np.linalg.norm(g["R"] - o["R"]) < 1e-4 and np.linalg.norm(g["t"] - o["t"]) < 1e-4
    assert (g["inlier"] != o["pose_mask"]).sum() <= 2 and abs(g["n_inliers"] - o["n_good"]) <= 2
    # the camera moves along +x: the previous-to-current translation of the scene is along -x or +x with |t_x| ~ 1
    assert abs(abs(g["t"][0, 0]) - 1) < 0.05
    ctx.close()


def test_track_pair_too_few_matches_fails_like_the_reference():
    import vslam_amd as V
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=2)
    prm = V.orb_params(nfeatures=500)
    fa, fb = ctx.orb_detect_compute(np.stack([synthetic_frame(1), synthetic_frame(2)]), prm)  # unrelated frames
    g = ctx.track_pair(fa[0], fa[1], fb[0], fb[1], 640, 480, K, ratio=0.6, disp_frac=0.001)
    assert len(g["sel"]) < 8 and np.isnan(g["R"]).all() and g["n_inliers"] == 0 and not g["inlier"].any()
    e = np.zeros(0, V.KP_DTYPE)
    g = ctx.track_pair(e, np.zeros((0, 32), np.uint8), fb[0], fb[1], 640, 480, K)
    assert len(g["sel"]) == 0 and np.isnan(g["t"]).all()
    ctx.close()


def test_track_mode_of_the_batched_call_vs_oracle():
    import torch
    import vslam_amd as V
    from oracle import geom_oracle as G
    from oracle import orb_oracle as O
    nb, cap = 16, 2048
    frames = parallax_frames(nb, seed=5, bg_step=3, fg_step=6)
    dev = torch.device("cuda", 0)
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = V.orb_params(nfeatures=2000)
    d_fr = torch.from_numpy(frames).to(dev)
    z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)
    kps = z(nb, cap, 7, dt=torch.float32); desc = z(nb, cap, 32, dt=torch.uint8); counts = z(nb)
    midx = z(nb - 1, cap, 2); mdist = z(nb - 1, cap, 2); mpass = z(nb - 1, cap, dt=torch.uint8)
    pose = z(nb - 1, 12, dt=torch.float64); pts = z(nb - 1, cap, 3, dt=torch.float32); npts = z(nb - 1)
    sel = z(nb - 1, cap, 2); seld = z(nb - 1, cap); seln = z(nb - 1); pmask = z(nb - 1, cap, dt=torch.uint8)
    io = V.BatchIO()
    io.d_gray = d_fr.data_ptr(); io.w = 640; io.h = 480; io.batch = nb; io.cap = cap
    io.ratio = 0.75; io.thr_px = 1.0; io.n_hyp = 1024; io.seed = 4096
    for i in range(9): io.K[i] = float(K.reshape(9)[i])
    io.d_kps = kps.data_ptr(); io.d_desc = desc.data_ptr(); io.d_counts = counts.data_ptr()
    io.d_match_idx = midx.data_ptr(); io.d_match_dist = mdist.data_ptr(); io.d_match_pass = mpass.data_ptr()
    io.d_pose = pose.data_ptr(); io.d_points = pts.data_ptr(); io.d_n_points = npts.data_ptr()
    io.mode = V.MODE_TRACK; io.disp_frac = 0.02
    io.d_sel_idx = sel.data_ptr(); io.d_sel_dist = seld.data_ptr(); io.d_sel_n = seln.data_ptr(); io.d_pose_mask = pmask.data_ptr()
    ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
    torch.cuda.synchronize()
    assert ctx.dev_status() == 0 and "track_filters" in dict(ctx.stage_times())
    host = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    feats = [host.orb_detect_compute(frames[i], prm)[0] for i in range(nb)]
    for i in (0, 1, 7, 14):
        o = _oracle_track(O, G, feats[i], feats[i + 1], 0.75, 0.02, pair=i)
        n = int(seln[i].item())
        assert n == len(o["sel_q"]) > 100
        s_ = sel[i, :n].cpu().numpy()
        assert np.array_equal(s_[:, 0], o["sel_q"]) and np.array_equal(s_[:, 1], o["sel_t"])
        assert np.array_equal(seld[i, :n].cpu().numpy(), o["sel_d"])
        got = pose[i].cpu().numpy()
        assert np.linalg.norm(got[:9].reshape(3, 3) - o["R"]) < 1e-4 and np.linalg.norm(got[9:] - o["t"].ravel()) < 1e-4, "pair %d" % i
        m = pmask[i].cpu().numpy().astype(bool)
        assert (m[o["sel_q"]] != o["pose_mask"]).sum() <= 2 and abs(int(npts[i].item()) - o["n_good"]) <= 2
        assert not m[~np.isin(np.arange(cap), o["sel_q"])].any()
    # the fused call of the drop-in utils module agrees with the batched mode on pair 0
    from orbslam2 import utils as geom
    from orbslam2.types import keypoints_from_array
    ok, T, inl = geom.track_from_last_frame(keypoints_from_array(feats[0][0]), feats[0][1], keypoints_from_array(feats[1][0]),
                                            feats[1][1], K, frames[0].shape, ratio_threshold=0.75, threshold_percent=0.02)
    assert ok and T.shape == (4, 4) and len(inl) > 50
    ctx.close(); host.close()


@pytest.mark.parametrize("detector", ["orb", "grid"])
@pytest.mark.parametrize("mode", ["init", "track", "keyframe"])
def test_every_mode_runs_on_every_detector(detector, mode):
    """mo_dev_frontend_batch: the three pose modes (MapInitializer / Tracker / LocalMapper) on both extraction paths (FAST pyramid /
    grid Shi-Tomasi) of one 12-frame batch: no capacity flag, finite poses (or F) for nearly every pair, consistent counts."""
    import ctypes as C
    import torch
    import vslam_amd as V
    from tests.helpers import parallax_frames
    from tests.test_gpu_dropin import _batch_io
    nb, cap = 12, 2048
    # fresh sensor noise per frame: without it the whole-pixel pans give IDENTICAL level-0 descriptors in consecutive frames, the
    # median match distance is 0 and the tracker's 2 x median filter (matcher.py:144-169) keeps nothing - in the reference as here
    rng = np.random.Generator(np.random.PCG64(5))
    frames = np.clip(parallax_frames(nb, seed=53).astype(np.float32) + rng.normal(0, 2.0, (nb, 480, 640)), 0, 255).round().astype(np.uint8)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    try:
        ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
        ctx.set_stream(st.cuda_stream)
        prm = V.orb_params(nfeatures=2000)
        io, b, K = _batch_io(torch, V, dev, torch.from_numpy(frames).to(dev), nb, cap, 512, want_mask=True)
        io.detector = V.DETECT_GRID if detector == "grid" else V.DETECT_ORB
        keep = []
        if mode == "track":
            sel = torch.zeros((nb - 1, cap, 2), dtype=torch.int32, device=dev); seln = torch.zeros(nb - 1, dtype=torch.int32, device=dev)
            io.mode = V.MODE_TRACK; io.disp_frac = 0.02; io.thr_px = 1.0; io.d_sel_idx = sel.data_ptr(); io.d_sel_n = seln.data_ptr()
            keep += [sel, seln]
        elif mode == "keyframe":
            pairs = [(0, 4), (4, 8), (8, 11)]
            q = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev); t = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
            P = np.zeros((len(pairs), 2, 3, 4)); P[:, :, :, :3] = K
            for j, (a, c2) in enumerate(pairs):
                P[j, 0, :, 3] = K @ np.array([-0.05 * a, 0, 0]); P[j, 1, :, 3] = K @ np.array([-0.05 * c2, 0, 0])
            dP1 = torch.from_numpy(np.ascontiguousarray(P[:, 0].reshape(-1, 12))).to(dev); dP2 = torch.from_numpy(np.ascontiguousarray(P[:, 1].reshape(-1, 12))).to(dev)
            dF = torch.zeros((len(pairs), 9), dtype=torch.float64, device=dev)
            io.mode = V.MODE_KEYFRAME; io.ratio = 0.8; io.n_kf_pairs = len(pairs); io.d_kf_query = q.data_ptr(); io.d_kf_train = t.data_ptr()
            io.d_kf_P1 = dP1.data_ptr(); io.d_kf_P2 = dP2.data_ptr(); io.d_kf_F = dF.data_ptr()
            keep += [q, t, dP1, dP2, dF]
        ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
        st.synchronize()
        assert ctx.dev_status() == 0
        cn = b["counts"].cpu().numpy()
        assert (cn > 300).all() and (cn <= 2000).all()
        if mode == "keyframe":
            F = keep[-1].cpu().numpy()
            assert np.isfinite(F).all() and (b["npts"][:3].cpu().numpy() > 50).all()
            X = b["pts"][:3].cpu().numpy(); m = b["pmask"][:3].cpu().numpy().astype(bool)
            assert np.isnan(X[~m]).all() and not np.isnan(X[m]).any()
        else:
            P = b["pose"].cpu().numpy()
            assert np.isfinite(P).all(axis=1).sum() >= nb - 2
            if mode == "track":
                assert (keep[1].cpu().numpy() > 30).sum() >= nb - 2
        ctx.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
