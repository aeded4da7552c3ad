"""GPU parity of the fundamental-matrix RANSAC (reference matcher.py:171-200 filter_matches_by_fundamental and
local_mapper.py:116-149 keyframe map growth) against the numpy oracle and synthetic ground truth."""
import numpy as np
import pytest

from oracle import geom_oracle as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vslam_amd as V
    c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=2)
    yield c
    c.close()


def _epi_dist(F, p1, p2):
    """symmetric point-to-epipolar-line distance in pixels"""
    h1 = np.concatenate([p1, np.ones((len(p1), 1))], 1); h2 = np.concatenate([p2, np.ones((len(p2), 1))], 1)
    l2 = h1 @ F.T; l1 = h2 @ F
    num = np.abs((h2 * l2).sum(1))
    return np.maximum(num / np.hypot(l2[:, 0], l2[:, 1]), num / np.hypot(l1[:, 0], l1[:, 1]))


@pytest.mark.parametrize("seed,n,of", [(4096, 2000, 0.3), (9, 100, 0.0), (11, 800, 0.5)])
def test_find_fundamental_vs_oracle_and_ground_truth(ctx, seed, n, of):
    s = G.synthetic_two_view(seed=seed, n=n, outlier_frac=of)
    F, mask = ctx.find_fundamental(s["p1"], s["p2"], thr_px=3.0, n_hyp=2048, seed=77)
    Fo, mo = G.find_fundamental_ransac8(s["p1"], s["p2"], thr_px=3.0, n_hyp=2048, seed=77)
    assert F is not None and abs(F[2, 2] - 1) < 1e-12
    assert np.linalg.norm(F - Fo) / np.linalg.norm(Fo) < 1e-4
    assert (mask != mo).sum() <= 2
    # ground truth: the true inliers are kept, the epipolar geometry fits them to a small fraction of a pixel, rank 2
    inl = ~s["outlier"]
    assert mask[inl].mean() > 0.99 and (inl.all() or mask[~inl].mean() < 0.2)
    assert _epi_dist(F, s["p1"][inl].astype(np.float64), s["p2"][inl].astype(np.float64)).max() < 0.05
    assert abs(np.linalg.det(F)) < 1e-9 * np.linalg.norm(F) ** 3
    # F agrees with K^-T [t]x R K^-1 up to scale
    Kinv = np.linalg.inv(s["K"]); t = s["t"].ravel()
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ft = Kinv.T @ tx @ s["R"] @ Kinv
    Ft = Ft / Ft[2, 2]
    assert np.linalg.norm(F - Ft) / np.linalg.norm(Ft) < 1e-3


def test_too_few_points_and_dropin_methods(ctx):
    from orbslam2 import utils as geom
    from orbslam2.matcher import DescriptorMatcher
    from orbslam2.types import DMatch, KeyPoint
    s = G.synthetic_two_view(seed=3, n=7, outlier_frac=0)
    F, mask = ctx.find_fundamental(s["p1"], s["p2"])
    assert F is None and not mask.any()
    # matcher.py:171-200: < 8 matches pass through with an all-true mask
    m = DescriptorMatcher()
    kp1 = [KeyPoint(float(x), float(y), 31) for x, y in s["p1"]]; kp2 = [KeyPoint(float(x), float(y), 31) for x, y in s["p2"]]
    ms = [DMatch(i, i, 0, 1.0) for i in range(7)]
    out, mk = m.filter_matches_by_fundamental(kp1, kp2, ms)
    assert out is ms and mk.dtype == bool and mk.all()
    s = G.synthetic_two_view(seed=5, n=600, outlier_frac=0.25)
    kp1 = [KeyPoint(float(x), float(y), 31) for x, y in s["p1"]]; kp2 = [KeyPoint(float(x), float(y), 31) for x, y in s["p2"]]
    ms = [DMatch(i, i, 0, 1.0) for i in range(600)]
    out, mk = m.filter_matches_by_fundamental(kp1, kp2, ms, threshold=3.0)
    assert mk.shape == (600,) and len(out) == mk.sum() and (mk & ~s["outlier"]).sum() > 0.99 * (~s["outlier"]).sum()
    Fm, mask = geom.calculate_fundamental_matrix(s["p1"], s["p2"], 3.0)
    assert Fm.shape == (3, 3) and mask.shape == (600, 1) and mask.dtype == np.uint8


def test_keyframe_map_growth_like_local_mapper():
    """local_mapper.py:116-149 on two real frames with known relative pose: ratio-0.8 matches -> F-RANSAC -> triangulation with the
    keyframe poses; the triangulated points must reproject onto their keypoints."""
    import vslam_amd as V
    from orbslam2 import utils as geom
    from orbslam2.types import keypoints_from_array
    from tests.helpers import parallax_frames
    frames = parallax_frames(2, seed=12, bg_step=4, fg_step=8)
    ctx = V.default_context()
    prm = V.orb_params(nfeatures=1500)
    (k1, d1), = ctx.orb_detect_compute(frames[0], prm)  # (the default context holds one frame at a time)
    (k2, d2), = ctx.orb_detect_compute(frames[1], prm)
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
    pose1 = np.eye(4); pose2 = np.eye(4); pose2[0, 3] = -1.0  # the scene moves left: the camera translates along +x
    pts, matches = geom.triangulate_new_map_points(keypoints_from_array(k1), d1, keypoints_from_array(k2), d2, pose1, pose2, K)
    assert len(matches) == len(pts) > 200 and pts.shape[1] == 3
    P2 = K @ pose2[:3]
    X = np.concatenate([pts.astype(np.float64), np.ones((len(pts), 1))], 1)
    pr = (P2 @ X.T).T
    pr = pr[:, :2] / pr[:, 2:3]
    obs = np.array([[k2["x"][m.trainIdx], k2["y"][m.trainIdx]] for m in matches])
    err = np.linalg.norm(pr - obs, axis=1)
    assert np.median(err) < 1.0  # two depth layers at 4 and 8 px disparity: z = 320 / disparity
    z = pts[:, 2]
    # (keypoints of coarse pyramid levels sit on a 1.2^L pixel grid: their disparities, hence depths, are quantised)
    assert ((np.abs(z - 80) < 25) | (np.abs(z - 40) < 10)).mean() > 0.8


def test_batched_keyframe_mode_vs_oracle_per_pair():
    """MO_MODE_KEYFRAME: LocalMapper._process_new_keyframe (reference local_mapper.py:116-149) for several keyframe pairs of a batch in
    one call - ratio-0.8 match keeping only queries with two neighbours, F-RANSAC at 3 px, triangulation of the inliers with the
    caller's two projection matrices.  Per pair: match lists equal the host call, the inlier mask / F equal the oracle's
    find_fundamental_ransac8 with the pair's own sampling stream, map points equal the oracle's DLT at 1e-4."""
    import ctypes as C
    import torch
    import vslam_amd as V
    from orbslam2 import utils as geom
    from tests.helpers import parallax_frames
    from tests.test_gpu_dropin import _batch_io
    nb, cap = 12, 2048
    frames = parallax_frames(nb, seed=41)
    pairs = [(0, 3), (3, 6), (6, 9), (2, 11), (5, 4)]
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    try:
        ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
        ctx.set_stream(st.cuda_stream)
        prm = V.orb_params(nfeatures=2000)
        d_fr = torch.from_numpy(frames).to(dev)
        io, b, K = _batch_io(torch, V, dev, d_fr, nb, cap, 1024, want_mask=True)
        # keyframe poses: the generator's camera moves 2 px of background parallax per frame along x; any poses do for the DLT
        poses = []
        for f in range(nb):
            T = np.eye(4); T[0, 3] = -0.05 * f
            poses.append(T)
        P1 = np.stack([geom.compute_projection_matrix(poses[q][:3, :3], poses[q][:3, 3], K) for q, _ in pairs])
        P2 = np.stack([geom.compute_projection_matrix(poses[t][:3, :3], poses[t][:3, 3], K) for _, t in pairs])
        d_q = torch.tensor([q for q, _ in pairs], dtype=torch.int32, device=dev)
        d_t = torch.tensor([t for _, t in pairs], dtype=torch.int32, device=dev)
        d_P1 = torch.from_numpy(P1.reshape(len(pairs), 12)).to(dev); d_P2 = torch.from_numpy(P2.reshape(len(pairs), 12)).to(dev)
        d_F = torch.zeros((len(pairs), 9), dtype=torch.float64, device=dev)
        io.mode = V.MODE_KEYFRAME; io.ratio = 0.8; io.thr_px = 3.0; io.n_kf_pairs = len(pairs)
        io.d_kf_query = d_q.data_ptr(); io.d_kf_train = d_t.data_ptr(); io.d_kf_P1 = d_P1.data_ptr(); io.d_kf_P2 = d_P2.data_ptr()
        io.d_kf_F = d_F.data_ptr(); io.pair_index_base = 100
        ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(io)))
        st.synchronize()
        assert ctx.dev_status() == 0
        assert [n for n, _ in ctx.stage_times()][-2:] == ["match_knn2_ratio", "keyframe_f_ransac_triangulate"]
        cn = b["counts"].cpu().numpy()
        kp = b["kps"].cpu().numpy()
        host = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
        for p, (q, t) in enumerate(pairs):
            dq, dt = b["desc"][q, :cn[q]].cpu().numpy(), b["desc"][t, :cn[t]].cpu().numpy()
            idx, dist, keep = host.match_knn2_ratio(dq, dt, 0.8)
            assert np.array_equal(b["midx"][p, :cn[q]].cpu().numpy(), idx) and np.array_equal(b["mpass"][p, :cn[q]].cpu().numpy().astype(bool), keep)
            sel = np.flatnonzero(keep & (idx[:, 1] >= 0))
            p1 = kp[q, sel, :2].astype(np.float32); p2 = kp[t, idx[sel, 0], :2].astype(np.float32)
            Fo, mo = G.find_fundamental_ransac8(p1, p2, thr_px=3.0, n_hyp=1024, seed=4096, pair=100 + p)
            F = d_F[p].cpu().numpy().reshape(3, 3)
            X = b["pts"][p].cpu().numpy(); mask_q = b["pmask"][p].cpu().numpy().astype(bool)
            assert Fo is not None and len(sel) > 200
            assert np.linalg.norm(F - Fo) / np.linalg.norm(Fo) < 1e-4, p
            got_mask = mask_q[sel]
            assert (got_mask != mo).sum() <= 2 and not mask_q[np.setdiff1d(np.arange(cap), sel)].any()
            assert int(b["npts"][p].item()) == int(mask_q.sum())
            both = got_mask & mo
            X4 = G.triangulate(P1[p], P2[p], p1[both].astype(np.float64), p2[both].astype(np.float64))  # unit-norm homogeneous points
            well = np.abs(X4[:, 3]) > 1e-2   # (w ~ 0: a point near infinity, X / w is ill-conditioned in the float32 the reference divides in)
            Xo = X4[:, :3] / X4[:, 3:4]
            e = np.linalg.norm(X[sel[both]] - Xo, axis=1) / np.maximum(np.linalg.norm(Xo, axis=1), 1e-9)
            assert well.mean() > 0.9 and np.median(e) < 1e-5 and e[well].max() < 1e-4, (p, np.median(e), e[well].max())
            assert np.isnan(X[~mask_q]).all() and not np.isnan(X[mask_q]).any()
        ctx.close(); host.close()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
