"""GPU tests of the one-frame-at-a-time API (frame_api.hip): resident single-frame results + tokens, the fused pair step
(mo_pair_frontend) behind matcher.match / track_from_last_frame / MapInitializer.initialize, and the sizes the reference allows that
rounds 1 - 3 refused: pairs with more than 4096 correspondences, frames with more than 8192 keypoints in the tracking filters.
Reference call pattern: src/orbslam2/tracker.py:87,168-170,214-254; src/orbslam2/initializer.py:67-120; tests/tester_map.py:57-75."""
import contextlib
import ctypes as C
import io

import numpy as np
import pytest

from tests.helpers import parallax_frames

pytestmark = pytest.mark.gpu
K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])


def _xy(k):
    return np.stack([k["x"], k["y"]], 1)


def test_tokens_name_resident_results_and_the_pair_step_uploads_nothing():
    """detect_and_compute leaves its result resident; the SAME arrays coming back take the token path, copies take the upload path, a
    token older than four extractions falls back to the arrays: identical match lists, kept lists and poses on every route, equal to
    the stand-alone entry points."""
    import vslam_amd as V
    frames = parallax_frames(6, seed=77, bg_step=3, fg_step=6)
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    prm = V.orb_params(nfeatures=2000)
    (ka, da), = ctx.orb_detect_compute(frames[0], prm)
    ta = ctx.last_token
    (kb, db), = ctx.orb_detect_compute(frames[1], prm)
    tb = ctx.last_token
    assert ta and tb and ta != tb
    assert not da.flags.writeable and not ka.flags.writeable      # (an in-place write cannot put the device copy out of date)
    assert V.resident_token(ctx, da, ka) == ta and V.resident_token(ctx, db) == tb and V.resident_token(ctx, da.copy()) == 0
    # matcher: resident route == upload route == the batched-pair entry point of rounds 1 - 3
    r_idx, r_dist, r_keep = ctx.match_knn2_ratio(da, db, 0.75)
    u_idx, u_dist, u_keep = ctx.match_knn2_ratio(da.copy(), db.copy(), 0.75)
    assert np.array_equal(r_idx, u_idx) and np.array_equal(r_dist, u_dist) and np.array_equal(r_keep, u_keep) and r_keep.sum() > 300
    n_idx, n_dist, n_keep = ctx.match_knn2_ratio(da, db, None)
    assert np.array_equal(n_idx, u_idx) and n_keep.all()
    # tracking step: tokens == arrays; the sampling stream follows pair_index
    g_tok = ctx.track_pair(ka, da, kb, db, 640, 480, K, n_hyp=1024)
    g_arr = ctx.track_pair(ka.copy(), da.copy(), kb.copy(), db.copy(), 640, 480, K, n_hyp=1024)
    for f in ("sel", "sel_dist", "inlier", "R", "t", "E"):
        assert np.array_equal(g_tok[f], g_arr[f]), f
    assert len(g_tok["sel"]) > 100 and g_tok["n_inliers"] == g_arr["n_inliers"] > 50
    g_p7 = ctx.track_pair(ka, da, kb, db, 640, 480, K, n_hyp=1024, pair_index=7)
    # (another sampling stream, the same consensus set after the refits on this scene: the pose may even come out bit-identical)
    assert np.array_equal(g_p7["sel"], g_tok["sel"]) and np.abs(g_p7["R"] - g_tok["R"]).max() < 1e-6
    # initialisation step: fused call == match + explicit-point two-view call
    f = ctx.pair_frontend(ka, da, kb, db, V.MODE_INIT, K, ratio=0.75, thr_px=3.0, n_hyp=1024)
    assert np.array_equal(f["idx"], u_idx) and np.array_equal(f["keep"], u_keep)
    q = np.flatnonzero(u_keep)
    e = ctx.init_two_view(_xy(ka)[q], _xy(kb)[u_idx[q, 0]], K, thr_px=3.0, n_hyp=1024)
    assert np.allclose(f["R"], e["R"], atol=1e-12) and np.allclose(f["t"], e["t"], atol=1e-12) and f["n_good"] == e["n_good"] > 100
    assert np.array_equal(f["pose_mask"][q], e["pose_mask"]) and np.array_equal(f["ransac_mask"][q], e["ransac_mask"])
    assert not f["pose_mask"][~u_keep].any()
    assert np.array_equal(f["X"][q][e["pose_mask"]], e["X"][e["pose_mask"]]) and np.isnan(f["X"][~f["pose_mask"]]).all()
    # four more extractions push frame a out of the slots: its token is stale, the arrays are uploaded again, same answer
    for i in range(2, 6):
        ctx.orb_detect_compute(frames[i], prm)
    g_old = ctx.track_pair(ka, da, kb, db, 640, 480, K, n_hyp=1024)
    for fld in ("sel", "sel_dist", "inlier", "R", "t"):
        assert np.array_equal(g_old[fld], g_arr[fld]), fld
    assert g_old["token1"] not in (ta, 0) and g_old["token2"] not in (tb, 0)
    # a detect-only call (no descriptors) leaves nothing a pair step could use
    ctx.orb_detect_compute(frames[0], prm, want_desc=False)
    assert ctx.last_token == 0
    ctx.close()


def test_tracker_loop_through_the_classes_vs_oracle():
    """tester_map.py:57-75 / tracker.py:87,214-254 through the drop-in classes: every frame extracted once, tracked against the previous
    one with BOTH resident; kept matches bit-exact and pose within 1e-4 of the oracle's restatement, frame after frame."""
    import vslam_amd as V
    from oracle import geom_oracle as G
    from oracle import orb_oracle as O
    from orbslam2 import utils as geom
    from orbslam2.extractor import ORBExtractor
    O.lib().orc_set_variant(0, 0)
    frames = parallax_frames(5, seed=5, bg_step=3, fg_step=6)
    ex = ORBExtractor(n_features=2000)
    last = None
    for i, fr in enumerate(frames):
        kps, desc = ex.detect_and_compute(fr)
        assert V.resident_token(V.default_context(), desc, kps.array) != 0
        if last is not None:
            ok, T, inl = geom.track_from_last_frame(last[0], last[1], kps, desc, K, fr.shape, pair_index=i - 1)
            idx, dist = O.match_knn2(last[1], desc)
            keep = O.ratio_test(idx, dist, 0.75)
            o = G.track_pair(_xy(last[0].array), _xy(kps.array), idx, dist, keep, K, 640, 480, frac=0.02, thr_px=1.0, n_hyp=4096, seed=4096,
                             pair=i - 1)
            assert ok and np.linalg.norm(T[:3, :3] - o["R"]) < 1e-4 and np.linalg.norm(T[:3, 3:4] - o["t"]) < 1e-4, i
            sel_q = np.array([m.queryIdx for m in inl]); sel_t = np.array([m.trainIdx for m in inl])
            om = o["pose_mask"]
            assert abs(len(inl) - int(om.sum())) <= 2
            if len(inl) == int(om.sum()):
                assert np.array_equal(sel_q, o["sel_q"][om]) and np.array_equal(sel_t, o["sel_t"][om])
        last = (kps, desc)


def test_initialize_one_call_equals_the_per_stage_route():
    """MapInitializer.initialize with the drop-in matcher is ONE device call; with any other matcher object it is matcher.match +
    one two-view call: same success flag, R, t, surviving matches and map points; and both equal the oracle at 1e-4."""
    from oracle import geom_oracle as G
    from orbslam2.extractor import ORBExtractor
    from orbslam2.initializer import MapInitializer
    from orbslam2.matcher import DescriptorMatcher
    frames = parallax_frames(2, seed=31, bg_step=4, fg_step=9)
    ex = ORBExtractor(n_features=2000)
    mt = DescriptorMatcher("bruteforce-hamming", ratio_threshold=0.75)

    class Wrapped:   # "some other matcher object": the per-stage route
        def match(self, a, b):
            return mt.match(a, b)
    out = []
    for m in (mt, Wrapped()):
        k0, d0 = ex.detect_and_compute(frames[0])
        k1, d1 = ex.detect_and_compute(frames[1])
        ini = MapInitializer(K)
        ini.set_first_frame(k0, d0, frames[0])
        with contextlib.redirect_stdout(io.StringIO()):
            out.append(ini.initialize(k1, d1, m, frames[1]))
        assert ini.initialization_done
    (ok_a, R_a, t_a, mp_a, ms_a), (ok_b, R_b, t_b, mp_b, ms_b) = out
    assert ok_a and ok_b and np.allclose(R_a, R_b, atol=1e-12) and np.allclose(t_a, t_b, atol=1e-12) and t_a.shape == (3, 1)
    assert [(m.queryIdx, m.trainIdx, m.distance) for m in ms_a] == [(m.queryIdx, m.trainIdx, m.distance) for m in ms_b] and len(ms_a) > 100
    assert len(mp_a) == len(mp_b) == len(ms_a)
    for a, b, m in zip(mp_a, mp_b, ms_a):
        assert np.array_equal(a["position"], b["position"]) and np.array_equal(a["color"], b["color"])
        assert a["keypoint_references"] == b["keypoint_references"] == {0: m.queryIdx, 1: m.trainIdx} and a["observed_frames"] == [0, 1]
    put = mt.match(d0, d1)
    p1 = np.float32([k0[m.queryIdx].pt for m in put]); p2 = np.float32([k1[m.trainIdx].pt for m in put])
    o = G.init_two_view(p1, p2, K, thr_px=3.0, n_hyp=4096, seed=4096)
    assert np.linalg.norm(R_a - o["R"]) < 1e-4 and np.linalg.norm(t_a - o["t"]) < 1e-4 and abs(len(mp_a) - o["n_good"]) <= 2


def _batch(torch, V, dev, frames, cap, n_hyp, Kc, w, h, ratio):
    nb = len(frames)
    z = lambda *s, dt=torch.int32: torch.zeros(s, dtype=dt, device=dev)
    b = dict(fr=torch.from_numpy(frames).to(dev), kps=z(nb, cap, 7, dt=torch.float32), desc=z(nb, cap, 32, dt=torch.uint8), counts=z(nb),
             midx=z(nb - 1, cap, 2), mdist=z(nb - 1, cap, 2), mpass=z(nb - 1, cap, dt=torch.uint8), pose=z(nb - 1, 12, dt=torch.float64),
             pts=z(nb - 1, cap, 3, dt=torch.float32), npts=z(nb - 1), sel=z(nb - 1, cap, 2), seld=z(nb - 1, cap), seln=z(nb - 1))
    o = V.BatchIO()
    o.d_gray = b["fr"].data_ptr(); o.w = w; o.h = h; o.batch = nb; o.cap = cap
    o.ratio = ratio; o.thr_px = 3.0; o.n_hyp = n_hyp; o.seed = 4096
    for i in range(9): o.K[i] = float(Kc.reshape(9)[i])
    o.d_kps = b["kps"].data_ptr(); o.d_desc = b["desc"].data_ptr(); o.d_counts = b["counts"].data_ptr()
    o.d_match_idx = b["midx"].data_ptr(); o.d_match_dist = b["mdist"].data_ptr(); o.d_match_pass = b["mpass"].data_ptr()
    o.d_pose = b["pose"].data_ptr(); o.d_points = b["pts"].data_ptr(); o.d_n_points = b["npts"].data_ptr()
    return o, b


@pytest.mark.parametrize("mode", ["init", "track"])
def test_pairs_with_more_than_4096_correspondences_vs_oracle(mode):
    """initializer.py:75-79 and tracker.py:238-242 hand EVERY match to findEssentialMat; ORBExtractor(n_features=6400) with the ratio
    test off is a legal configuration.  Batched call on 1280 x 720 frames, 6400 keypoints per frame (cap 6464: the tracking filters' key
    arrays leave LDS for their HBM slot), ratio < 0 = every query is a correspondence: init mode 6400 correspondences per pair against
    geom_oracle.init_two_view at 1e-4; track mode the kept list bit-exact against geom_oracle.track_select and the pose at 1e-4."""
    import torch
    import vslam_amd as V
    from oracle import geom_oracle as G
    nb, cap, nfeat, w, h = 3, 6464, 6400, 1280, 720
    frames = parallax_frames(nb, seed=61, w=w, h=h, bg_step=8, fg_step=16)
    rng = np.random.Generator(np.random.PCG64(9))
    frames = np.clip(frames.astype(np.float32) + rng.normal(0, 2.0, frames.shape), 0, 255).round().astype(np.uint8)
    Kc = np.array([[640.0, 0, 640.0], [0, 640.0, 360.0], [0, 0, 1.0]])
    dev = torch.device("cuda", 0)
    ctx = V.Context(device=0, max_w=w, max_h=h, max_batch=nb)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        prm = V.orb_params(nfeatures=nfeat)
        o, b = _batch(torch, V, dev, frames, cap, 512, Kc, w, h, -1.0)
        if mode == "track":
            o.mode = V.MODE_TRACK; o.disp_frac = 0.02; o.thr_px = 1.0
            o.d_sel_idx = b["sel"].data_ptr(); o.d_sel_dist = b["seld"].data_ptr(); o.d_sel_n = b["seln"].data_ptr()
        ctx._check(ctx.lib.mo_dev_frontend_batch(ctx.h, C.byref(prm), C.byref(o)))
        torch.cuda.synchronize()
        assert ctx.dev_status() == 0
        cn = b["counts"].cpu().numpy()
        assert (cn == nfeat).all()
        kp = b["kps"].cpu().numpy()
        P, NP = b["pose"].cpu().numpy(), b["npts"].cpu().numpy()
        for i in range(nb - 1):
            idx = b["midx"][i, :nfeat].cpu().numpy(); dist = b["mdist"][i, :nfeat].cpu().numpy()
            ps = b["mpass"][i, :nfeat].cpu().numpy().astype(bool)
            assert ps.all()                                        # ratio test off: every query passes
            xy1, xy2 = kp[i, :nfeat, :2], kp[i + 1, :nfeat, :2]
            if mode == "init":
                og = G.init_two_view(xy1, xy2[idx[:, 0]], Kc, thr_px=3.0, n_hyp=512, seed=4096, pair=i)
                assert np.linalg.norm(P[i, :9].reshape(3, 3) - og["R"]) < 1e-4 and np.linalg.norm(P[i, 9:] - og["t"].ravel()) < 1e-4, i
                assert abs(int(NP[i]) - og["n_good"]) <= 3 and og["n_good"] > 500
            else:
                og = G.track_pair(xy1, xy2, idx, dist, ps, Kc, w, h, frac=0.02, thr_px=1.0, n_hyp=512, seed=4096, pair=i)
                n_sel = int(b["seln"][i].item())
                s = b["sel"][i, :n_sel].cpu().numpy()
                assert n_sel == len(og["sel_q"]) > 1000 and s[:, 0].max() > 4096
                assert np.array_equal(s[:, 0], og["sel_q"]) and np.array_equal(s[:, 1], og["sel_t"])
                assert np.array_equal(b["seld"][i, :n_sel].cpu().numpy(), og["sel_d"])
                assert np.linalg.norm(P[i, :9].reshape(3, 3) - og["R"]) < 1e-4 and np.linalg.norm(P[i, 9:] - og["t"].ravel()) < 1e-4, i
    finally:
        ctx.close()


def test_initialize_with_6000_matches_ratio_test_off():
    """the same through the class API: ORBExtractor(n_features=6000) and a matcher whose ratio test is off hand ~ 6000 matches to
    MapInitializer.initialize (rounds 1 - 3: MO_ERR_UNSUPPORTED beyond 4096); R, t against the oracle on the same matches at 1e-4."""
    from oracle import geom_oracle as G
    from orbslam2.extractor import ORBExtractor
    from orbslam2.initializer import MapInitializer
    from orbslam2.matcher import DescriptorMatcher
    w, h = 1280, 720
    frames = parallax_frames(2, seed=61, w=w, h=h, bg_step=8, fg_step=16)
    Kc = np.array([[640.0, 0, 640.0], [0, 640.0, 360.0], [0, 0, 1.0]])
    ex = ORBExtractor(n_features=6000)
    base = DescriptorMatcher("bruteforce-hamming")

    class NoRatio:
        def match(self, a, b):
            return base.match(a, b, ratio_test=False)
    k0, d0 = ex.detect_and_compute(frames[0])
    k1, d1 = ex.detect_and_compute(frames[1])
    assert len(k0) == len(k1) == 6000
    ini = MapInitializer(Kc)
    ini.set_first_frame(k0, d0, frames[0])
    with contextlib.redirect_stdout(io.StringIO()):
        ok, R, t, pts, ms = ini.initialize(k1, d1, NoRatio(), frames[1])
    put = base.match(d0, d1, ratio_test=False)
    assert len(put) == 6000
    p1 = np.float32([k0.array[m.queryIdx][["x", "y"]].tolist() for m in put]); p2 = np.float32([k1.array[m.trainIdx][["x", "y"]].tolist() for m in put])
    o = G.init_two_view(p1, p2, Kc, thr_px=3.0, n_hyp=4096, seed=4096)
    assert ok and np.linalg.norm(R - o["R"]) < 1e-4 and np.linalg.norm(t - o["t"]) < 1e-4 and abs(len(pts) - o["n_good"]) <= 3 and len(pts) > 500


@pytest.mark.parametrize("chunk,detector", [(4, "orb"), (7, "orb"), (5, "grid")])
def test_frame_stream_equals_the_per_frame_loop(chunk, detector):
    """vslam_amd.stream.FrameStream (mo_stream: chunks through mo_dev_frontend_batch, double-buffered upload, halo frame) against the
    per-frame loop of the reference's driver (tester_map.py:57-75) through the single-frame API with pair_index = i: keypoints,
    descriptors and kept tracking matches bit-identical, poses bit-identical (the sampler is keyed by the global pair index), for chunk
    sizes that do and do not divide the sequence."""
    import vslam_amd as V
    from vslam_amd.stream import FrameStream
    nfr = 18
    frames = parallax_frames(nfr, seed=5, bg_step=3, fg_step=6)
    grid = detector == "grid"
    fs = FrameStream(K, chunk=chunk, n_features=2000, cap=2048, detector=V.DETECT_GRID if grid else V.DETECT_ORB, n_hyp=1024)
    try:
        got = list(fs.run(iter(frames)))
    finally:
        fs.close()
    assert [g.index for g in got] == list(range(nfr)) and got[0].pair is None
    # (copy=True, the default: the results outlive the stream.  Views of a copy=False stream are refused once their buffer is gone)
    fv = FrameStream(K, chunk=chunk, n_features=2000, cap=2048, detector=V.DETECT_GRID if grid else V.DETECT_ORB, n_hyp=64, copy=False)
    try:
        nst = fv.lanes * chunk + 1                          # one chunk more than there are lanes (mo_stream_lanes)
        stale = list(fv.run(np.concatenate([frames] * (nst // nfr + 1))[:nst]))
        assert len(stale[-1].keypoints) > 300              # the last chunk's buffer is still there
        with pytest.raises(RuntimeError):
            stale[0].keypoints                              # the first chunk's lane was reused by the last one
    finally:
        fv.close()
    with pytest.raises(RuntimeError):
        stale[-1].descriptors
    ctx = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    prm = V.orb_params(nfeatures=2000)
    last = None
    for i in range(nfr):
        if grid:
            xy, kept, d = ctx.grid_detect_compute(frames[i], prm, 2000)
            k = np.zeros(len(kept), V.KP_DTYPE)
            k["x"], k["y"], k["size"], k["angle"], k["class_id"] = xy[kept, 0], xy[kept, 1], 31, -1, -1
        else:
            (k, d), = ctx.orb_detect_compute(frames[i], prm)
        g = got[i]
        assert np.array_equal(g.keypoints, k) and np.array_equal(g.descriptors, d), i
        if last is not None:
            r = ctx.track_pair(last[0], last[1], k, d, 640, 480, K, n_hyp=1024, pair_index=i - 1)
            p = g.pair
            assert p["pair_index"] == i - 1
            assert np.array_equal(p["sel"], r["sel"]) and np.array_equal(p["sel_dist"], r["sel_dist"]), i
            assert p["ok"] == (len(r["sel"]) >= 8 and bool(np.isfinite(r["R"]).all()))
            if p["ok"]:
                assert np.array_equal(p["R"], r["R"]) and np.array_equal(p["t"], r["t"]), i
                assert np.array_equal(p["inlier"], r["inlier"]) and p["n_inliers"] == r["n_inliers"], i
        last = (k, d)
    ctx.close()


def test_grid_detector_results_are_resident_too():
    """ORBExtractor.distribute_keypoints(aligned=True) - Tracker's default detector (tracker.py:87) with keypoint i = descriptor row i -
    leaves its kept corners and descriptors resident like detect_and_compute: the tracking step on these very arrays takes the token
    route and equals the upload route; corners / descriptors equal the oracle (extractor.py:85-144)."""
    import vslam_amd as V
    from oracle import orb_oracle as O
    from orbslam2 import utils as geom
    from orbslam2.extractor import ORBExtractor
    frames = parallax_frames(3, seed=37, bg_step=3, fg_step=6)
    # (sensor noise: a noise-free whole-pixel pan gives identical patches, Hamming distance 0 for most matches, a median of 0 and - by the
    #  reference's strict "distance < 2 x median" - no kept match at all)
    rng = np.random.Generator(np.random.PCG64(3))
    frames = np.clip(frames.astype(np.float32) + rng.normal(0, 2.0, frames.shape), 0, 255).round().astype(np.uint8)
    ex = ORBExtractor(n_features=2000)
    ctx = V.default_context()
    out = []
    for f in frames:
        kps, desc = ex.distribute_keypoints(f, aligned=True)
        exy = O.grid_good_features(f, 2000)
        kin = np.zeros(len(exy), V.KP_DTYPE)
        kin["x"], kin["y"], kin["size"], kin["angle"], kin["class_id"] = exy[:, 0], exy[:, 1], 31, -1, -1
        kept, edesc = O.compute(f, O.params(nfeatures=2000), kin)
        assert np.array_equal(kps.array, kin[kept]) and np.array_equal(desc, edesc) and len(kept) > 300
        assert V.resident_token(ctx, desc, kps.array) != 0
        out.append((kps, desc))
    (k0, d0), (k1, d1), _ = out
    ok, T, inl = geom.track_from_last_frame(k0, d0, k1, d1, K, frames[0].shape)           # token route: both among the last four results
    ok2, T2, inl2 = geom.track_from_last_frame(k0, np.array(d0), k1, np.array(d1), K, frames[0].shape)   # copies: upload route
    assert ok and ok2 and np.array_equal(T, T2) and [(m.queryIdx, m.trainIdx) for m in inl] == [(m.queryIdx, m.trainIdx) for m in inl2]
    # the reference's own (misaligned) list is unchanged: all corners as KeyPoint objects next to the kept corners' descriptors
    allk, d_all = ex.distribute_keypoints(frames[0])
    assert len(allk) >= len(d_all) == len(d0) and np.array_equal(d_all, d0) and allk[0].size == 31 and allk[0].angle == -1
