"""GPU parity of the ORB extractor stages: HIP path through the C-ABI vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from tests.helpers import greedy_min_dist, gt_pair, load_png_bgr, synthetic_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vslam_amd as V
    c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=8)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import orb_oracle
    return orb_oracle


def _prm(V, O, order, **kw):
    p = V.orb_params(select_order=order, **kw)
    o = O.params(**kw)
    return p, o


@pytest.mark.parametrize("size", [(640, 480), (478, 850), (333, 257)])
def test_pyramid_and_blur_levels(ctx, O, size):
    import vslam_amd as V
    w, h = size
    img = synthetic_frame(7, w, h)
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX)
    lw, lh, _, _ = O.levels(w, h, o)
    for L in range(8):
        got = ctx.dbg_pyramid_level(img, p, L)
        exp = O.pyramid_level(img, o, L)
        assert got.shape == exp.shape == (lh[L], lw[L])
        assert np.array_equal(got, exp), "raw level %d" % L
        gotb = ctx.dbg_pyramid_level(img, p, L, blurred=True)
        expb = O.pyramid_level(img, o, L, blurred=True)
        assert np.array_equal(gotb, expb), "blurred level %d" % L


@pytest.mark.parametrize("thr", [7, 20])
def test_fast_nms_candidates(ctx, O, thr):
    import vslam_amd as V
    img = synthetic_frame(11)
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX, fast_threshold=thr)
    lw, lh, _, _ = O.levels(640, 480, o)
    for L in range(8):
        lvl = O.pyramid_level(img, o, L)
        f = O.fast_level(lvl, thr)
        f = f[(f[:, 0] >= 31) & (f[:, 0] < lw[L] - 31) & (f[:, 1] >= 31) & (f[:, 1] < lh[L] - 31)]
        got = ctx.dbg_fast_level(img, p, L)
        assert len(got) == len(f), "level %d count" % L
        assert np.array_equal(got, f), "level %d raster-ordered (x, y, score)" % L


@pytest.mark.parametrize("thr", [0, 1, 7])
def test_fast_nms_on_noise_uses_the_dense_fallback(ctx, O, thr):
    """Uniform noise: about a third of all pixels are FAST corners, far more per strip than the kernel's corner list holds
    (FAST_CORNER_CAP), so the non-max suppression takes its dense fallback; thresholds 0 and 1 also cover stored scores
    of 0 (corner with score 0 is never a candidate)."""
    import vslam_amd as V
    rng = np.random.Generator(np.random.PCG64(99 + thr))
    img = rng.integers(0, 256, size=(200, 320), dtype=np.uint8)
    img[40:90, 100:260] = rng.integers(100, 104, size=(50, 160), dtype=np.uint8)  # a low-contrast patch: scores near the threshold
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX, fast_threshold=thr, nlevels=3)
    lw, lh, _, _ = O.levels(320, 200, o)
    for L in range(3):
        lvl = O.pyramid_level(img, o, L)
        f = O.fast_level(lvl, thr)
        f = f[(f[:, 0] >= 31) & (f[:, 0] < lw[L] - 31) & (f[:, 1] >= 31) & (f[:, 1] < lh[L] - 31)]
        got = ctx.dbg_fast_level(img, p, L)
        assert len(got) == len(f) > 0, "level %d count" % L
        assert np.array_equal(got, f), "level %d raster-ordered (x, y, score)" % L
    O.lib().orc_set_variant(0, 0)  # libstdc++ order, like p
    (k, d), = ctx.orb_detect_compute(img, p)
    ek, ed = O.detect_and_compute(img, o)
    assert np.array_equal(k, ek) and np.array_equal(d, ed)


def _retain_cases():
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 3, 4, 7, 31, 32, 33, 34, 40, 41, 42, 64, 100, 257, 383, 385, 640, 1000, 1025, 3711, 5000, 16385, 16800, 20000):
        for kind in ("float", "ties", "sorted", "rsorted", "const"):
            if kind == "float": r = rng.normal(size=n).astype(np.float32)
            elif kind == "ties": r = rng.integers(0, 12, size=n).astype(np.float32)
            elif kind == "sorted": r = np.sort(rng.normal(size=n)).astype(np.float32)
            elif kind == "rsorted": r = np.sort(rng.normal(size=n))[::-1].astype(np.float32)
            else: r = np.full(n, 3.0, np.float32)
            for k in sorted({0, 1, 2, n // 3, n // 2, n - 1, n, n + 5}):
                cases.append((r, k))
    # median-of-3 killer: drives libstdc++ introselect into its heap-select fallback (n = 64 / 128: inside the register-resident tail)
    for n in (64, 128, 4096):
        a = np.zeros(n, np.float32)
        k = n // 2
        for i in range(1, k + 1):
            if i & 1:
                a[i - 1] = i; a[i] = k + i
            a[k + i - 1] = 2 * i
        cases.append((-a, n // 2)); cases.append((a, n // 2)); cases.append((a, 10))
    # every size of the register-resident tail (ranges of 4 .. 64 records), tie-heavy and distinct
    for n in range(4, 66):
        cases.append((rng.integers(0, 5, size=n).astype(np.float32), n // 2))
        cases.append((rng.normal(size=n).astype(np.float32), max(1, n // 3)))
    return cases


def test_retain_best_replay_both_orders(ctx, O):
    import vslam_amd as V
    for order in (V.ORDER_LIBSTDCXX, V.ORDER_MSVC):
        O.lib().orc_set_variant(order, 0)
        for r, k in _retain_cases():
            exp = O.retain_best(r, k)
            got = ctx.dbg_retain_best(r, k, order)
            assert np.array_equal(got, exp), "order %d n %d k %d" % (order, len(r), k)
    O.lib().orc_set_variant(1, 0)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("nfeat,thr", [(2000, 7), (500, 20), (200, 20)])
def test_detect_and_compute_bit_exact(ctx, O, order, nfeat, thr):
    import vslam_amd as V
    O.lib().orc_set_variant(order, 0)
    p, o = _prm(V, O, order, nfeatures=nfeat, fast_threshold=thr)
    for seed in (20250523, 20250524):
        img = synthetic_frame(seed)
        (kps, desc), = ctx.orb_detect_compute(img, p)
        ek, ed = O.detect_and_compute(img, o)
        assert len(kps) == len(ek)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(kps[f], ek[f]), f
        assert np.array_equal(desc, ed)
    O.lib().orc_set_variant(1, 0)


def test_detector_kat_through_the_c_abi(ctx):
    """The reference's 40 known-answer keypoint lists, reproduced by the HIP path itself (BGR input, MSVC order)."""
    import vslam_amd as V
    p = V.orb_params(nfeatures=200, fast_threshold=20, select_order=V.ORDER_MSVC)
    for pair in range(1, 21):
        d, gt = gt_pair(pair)
        for k in (1, 2):
            bgr = load_png_bgr(os.path.join(d, "img%d.png" % k))
            (kps, _), = ctx.orb_detect_compute(bgr, p, want_desc=False)
            got = greedy_min_dist([(float(a["x"]), float(a["y"])) for a in kps])
            assert got == [tuple(q) for q in gt["keypoints%d" % k]], "pair %d img %d" % (pair, k)


def test_batch_equals_single(ctx):
    import vslam_amd as V
    p = V.orb_params(nfeatures=1000)
    imgs = np.stack([synthetic_frame(100 + i) for i in range(5)])
    res = ctx.orb_detect_compute(imgs, p)
    for i in range(5):
        (k1, d1), = ctx.orb_detect_compute(imgs[i], p)
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1)


@pytest.mark.parametrize("nb", [9, 17, 23])
def test_batches_of_8k_plus_r_frames_equal_single(nb):
    """XCD-affine workgroup mapping (orb_kernels.hip xcd_map): the first 8 * (nb // 8) frames are dealt by id residue, the last
    nb % 8 keep the plain mapping - a sharded rank with a halo frame runs exactly this shape.  Every frame of the batch must
    equal its single-frame result."""
    import vslam_amd as V
    p = V.orb_params(nfeatures=800)
    imgs = np.stack([synthetic_frame(500 + i) for i in range(nb)])
    c = V.Context(device=0, max_w=640, max_h=480, max_batch=nb)
    try:
        res = c.orb_detect_compute(imgs, p)
        for i in range(nb):
            (k1, d1), = c.orb_detect_compute(imgs[i], p)
            assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1), i
    finally:
        c.close()


@pytest.mark.parametrize("edge,scale,nlev", [(19, 1.2, 8), (20, 1.2, 8), (48, 1.2, 8), (31, 1.5, 5), (40, 2.0, 4)])
def test_level_margins_follow_the_edge_threshold(O, edge, scale, nlev, monkeypatch):
    """Margins = f(edge_threshold): none at 19 - 22 (the descriptor radius), 28 / 24 px at 48; coarser scale factors move the source
    window of the next level.  Poisoned buffers, output equal to the oracle's."""
    import vslam_amd as V
    img = synthetic_frame(99)
    kw = dict(nfeatures=1500, fast_threshold=9, edge_threshold=edge, scale_factor=scale, nlevels=nlev)
    p, o = _prm(V, O, 0, **kw)
    O.lib().orc_set_variant(0, 0)
    ek, ed = O.detect_and_compute(img, o)
    assert len(ek) > 300
    c = V.Context(device=0, max_w=640, max_h=480, max_batch=1)
    c._check(c.lib.mo_dbg_set_poison(c.h, 171))
    try:
        (k, d), = c.orb_detect_compute(img, p)
        for f in ("x", "y", "angle", "response", "octave"):
            assert np.array_equal(k[f], ek[f]), f
        assert np.array_equal(d, ed)
    finally:
        c.close()
        O.lib().orc_set_variant(1, 0)


def test_level_ratio_above_two_keeps_the_gather_resize(O):
    """scale_factor 2.0 on a 666-px-wide frame: level 2 is 166 px wide from 333 (ratio 2.006), so four adjacent output columns can need
    9 consecutive source bytes - more than k_resize2's 8-byte window; mo_build_plan detects it per level and that level is resized by
    the gather kernel.  Pyramid levels and the final output must equal the oracle's."""
    import vslam_amd as V
    img = synthetic_frame(4242, 666, 500)
    kw = dict(nfeatures=1200, fast_threshold=9, scale_factor=2.0, nlevels=4)
    p, o = _prm(V, O, 0, **kw)
    O.lib().orc_set_variant(0, 0)
    c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=1)
    try:
        for L in range(4):
            assert np.array_equal(c.dbg_pyramid_level(img, p, L), O.pyramid_level(img, o, L)), L
        ek, ed = O.detect_and_compute(img, o)
        (k, d), = c.orb_detect_compute(img, p)
        for f in ("x", "y", "angle", "response", "octave"):
            assert np.array_equal(k[f], ek[f]), f
        assert np.array_equal(d, ed) and len(ek) > 300
    finally:
        c.close()
        O.lib().orc_set_variant(1, 0)


@pytest.mark.parametrize("size", [(640, 480), (478, 850), (333, 257)])
def test_skipped_level_margins_never_reach_a_result(O, size, monkeypatch):
    """The pipeline leaves the outer 8 px of pyramid levels 1.. and the outer 12 px of the blurred levels unwritten (nothing it
    computes reads them).  With the buffers poisoned by two different bytes before every extraction the output must still be the
    oracle's, bit for bit - keypoints, angles and descriptors."""
    import vslam_amd as V
    img = synthetic_frame(321, size[0], size[1])
    p, o = _prm(V, O, 0, nfeatures=2000, fast_threshold=7)
    O.lib().orc_set_variant(0, 0)
    ek, ed = O.detect_and_compute(img, o)
    for poison in (0, 255, 90):
        c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=2)
        c._check(c.lib.mo_dbg_set_poison(c.h, poison))
        try:
            for _ in range(2):
                (k, d), = c.orb_detect_compute(img, p)
                for f in ("x", "y", "angle", "response", "octave"):
                    assert np.array_equal(k[f], ek[f]), (poison, f)
                assert np.array_equal(d, ed), poison
        finally:
            c.close()
    O.lib().orc_set_variant(1, 0)


def test_stage_times_ring(ctx):
    """mo_stage_times_back: one event set per call, readable after later calls (bench.py reads a whole timed region after ONE sync)."""
    import vslam_amd as V
    p = V.orb_params(nfeatures=500)
    img = synthetic_frame(7)
    ctx.orb_detect_compute(img, p)
    assert ctx.stage_times() == []          # the single-call host entry points record no events unless asked to (mo_set_host_timing)
    ctx.set_host_timing(True)
    try:
        ctx.orb_detect_compute(img, p)
        ctx.match_knn2_ratio(np.zeros((4, 32), np.uint8), np.zeros((5, 32), np.uint8), 0.75)
        last, before = ctx.stage_times(0), ctx.stage_times(1)
        assert [n for n, _ in last] == ["h2d", "match_knn2_ratio", "d2h"]
        assert [n for n, _ in before] == ["h2d", "pyramid", "blur", "fast_nms", "select_harris", "angle_rbrief", "d2h"]
        assert all(ms >= 0.0 for _, ms in last + before)
        assert ctx.stage_times() == last
        with pytest.raises(V.NativeError):
            ctx.stage_times(V.TIMING_SLOTS)
    finally:
        ctx.set_host_timing(False)


def test_compute_given_keypoints(ctx, O):
    """orb.compute at caller keypoints: border drop, angle as supplied (-1), octave honoured, unsorted regroup."""
    import vslam_amd as V
    img = synthetic_frame(3)
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX)
    rng = np.random.default_rng(1)
    n = 300
    k = np.zeros(n, V.KP_DTYPE)
    k["x"] = rng.uniform(0, 640, n).astype(np.float32); k["y"] = rng.uniform(0, 480, n).astype(np.float32)
    k["size"] = 31; k["angle"] = -1; k["class_id"] = -1
    kept, desc = ctx.orb_compute(img, p, k)
    ekept, edesc = O.compute(img, o, k)
    assert np.array_equal(kept, ekept) and np.array_equal(desc, edesc)
    assert len(kept) < n  # some were inside the 31-px border band
    k["octave"] = rng.integers(0, 4, n); k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kept, desc = ctx.orb_compute(img, p, k)
    ekept, edesc = O.compute(img, o, k)
    assert np.array_equal(kept, ekept) and np.array_equal(desc, edesc)


def test_no_keypoints_on_flat_image(ctx):
    import vslam_amd as V
    p = V.orb_params()
    (kps, desc), = ctx.orb_detect_compute(np.full((480, 640), 128, np.uint8), p)
    assert len(kps) == 0 and desc is None


@pytest.mark.parametrize("order", [0, 1])
def test_full_hd_frame_uses_hbm_scratch_path(order, O):
    """1920x1080: level 0 has far more candidates than the 48 KB LDS window -> the selection replay runs out of its
    HBM scratch slot (and, at threshold 3, past the 65535-element wave-partition limit)."""
    import vslam_amd as V
    ctx = V.Context(device=0, max_w=1920, max_h=1080, max_batch=1)
    img = synthetic_frame(77, 1920, 1080)
    O.lib().orc_set_variant(order, 0)
    for nfeat, thr in ((2000, 7), (5000, 3)):
        p, o = _prm(V, O, order, nfeatures=nfeat, fast_threshold=thr)
        (kps, desc), = ctx.orb_detect_compute(img, p)
        ek, ed = O.detect_and_compute(img, o)
        assert len(kps) == len(ek)
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(kps[f], ek[f]), f
        assert np.array_equal(desc, ed)
    O.lib().orc_set_variant(1, 0)
    ctx.close()


def test_small_and_odd_sizes(ctx, O):
    import vslam_amd as V
    for (w, h), nlev in (((97, 131), 3), ((64, 64), 1), ((641, 479), 8), ((200, 150), 8)):
        img = synthetic_frame(9, w, h)
        p, o = _prm(V, O, V.ORDER_LIBSTDCXX, nfeatures=300, nlevels=nlev)
        O.lib().orc_set_variant(0, 0)
        (kps, desc), = ctx.orb_detect_compute(img, p)
        ek, ed = O.detect_and_compute(img, o)
        assert np.array_equal(kps, ek.astype(kps.dtype)) or all(np.array_equal(kps[f], ek[f]) for f in kps.dtype.names)
        if len(ek):
            assert np.array_equal(desc, ed)
        else:
            assert desc is None
    O.lib().orc_set_variant(1, 0)


def test_bad_arguments_fail_loudly(ctx):
    import vslam_amd as V
    with pytest.raises(V.NativeError):
        ctx.orb_detect_compute(np.zeros((32, 32), np.uint8), V.orb_params())          # below 64x64
    with pytest.raises(V.NativeError):
        ctx.orb_detect_compute(np.zeros((2000, 2000), np.uint8), V.orb_params())      # above the context's max size
    bad = V.orb_params()
    bad.wta_k = 3
    with pytest.raises(V.NativeError):
        ctx.orb_detect_compute(np.zeros((480, 640), np.uint8), bad)


def test_capacity_retry_and_huge_nfeatures(ctx, O):
    """cap smaller than the result -> MO_ERR_CAPACITY with the needed counts (the binding retries); nfeatures larger than
    the number of corners -> no retainBest at all, raster order per level."""
    import ctypes as C
    import vslam_amd as V
    img = synthetic_frame(13)
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX, nfeatures=1500)
    O.lib().orc_set_variant(0, 0)
    kps = np.zeros((1, 100), V.KP_DTYPE); desc = np.zeros((1, 100, 32), np.uint8); counts = np.zeros(1, np.int32)
    rc = ctx.lib.mo_orb_detect_compute(ctx.h, C.byref(p), img.ctypes.data_as(C.c_void_p), 640, 480, 640, 1, 1,
                                       kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), 100,
                                       counts.ctypes.data_as(C.c_void_p))
    assert rc == V.MO_ERR_CAPACITY and counts[0] == 1500
    (k, d), = ctx.orb_detect_compute(img, p, cap=100)   # wrapper grows the buffers and retries
    ek, ed = O.detect_and_compute(img, o)
    assert len(k) == 1500 and np.array_equal(d, ed)
    p, o = _prm(V, O, V.ORDER_LIBSTDCXX, nfeatures=200000, fast_threshold=20)
    (k, d), = ctx.orb_detect_compute(img, p, cap=60000)
    ek, ed = O.detect_and_compute(img, o)
    assert len(k) == len(ek) and np.array_equal(k["x"], ek["x"]) and np.array_equal(k["y"], ek["y"]) and np.array_equal(d, ed)
    O.lib().orc_set_variant(1, 0)


def test_response_ties_beyond_the_level_slot_grow_the_slots(O):
    """retainBest keeps EVERY element that ties with the quota boundary (KeyPointsFilter::retainBest: 'first among those equal to
    the n-th'), so a level of a periodic pattern - an 8-px checkerboard: every corner has the same response - keeps far more than
    its quota.  The per-level slots (4 x quota + 256) overflowed there (MO_ERR_CAPACITY until round 3; found by
    tools/fuzz_parity.py); they now grow up to the level's candidate capacity: the host call repeats by itself, a device-resident
    call raises status bit 0 once and succeeds when repeated."""
    import ctypes as C
    import torch
    import vslam_amd as V
    yy, xx = np.mgrid[0:264, 0:924]
    img = (((yy // 8 + xx // 8) & 1) * 255).astype(np.uint8)
    kw = dict(nfeatures=200, scale_factor=1.5, nlevels=8, fast_threshold=5, edge_threshold=25)
    O.lib().orc_set_variant(1, 0)
    ek, ed = O.detect_and_compute(img, O.params(**kw))
    assert len(ek) > 2000   # ties: an order of magnitude more than nfeatures
    c = V.Context(device=0, max_w=924, max_h=264, max_batch=2)
    try:
        prm = V.orb_params(select_order=V.ORDER_MSVC, **kw)
        (kps, desc), = c.orb_detect_compute(img, prm)
        assert all(np.array_equal(kps[f], ek[f]) for f in kps.dtype.names) and np.array_equal(desc, ed)
    finally:
        c.close()
    # device-resident call on a fresh context: flag once, then the repeated call fits
    c = V.Context(device=0, max_w=924, max_h=264, max_batch=2)
    try:
        dev = torch.device("cuda", 0)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        cap = len(ek) + 64
        fr = torch.from_numpy(np.stack([img, img])).to(dev)
        kp = torch.zeros((2, cap, 7), dtype=torch.float32, device=dev); ds = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        cn = torch.zeros(2, dtype=torch.int32, device=dev)
        io = V.BatchIO()
        io.d_gray = fr.data_ptr(); io.w = 924; io.h = 264; io.batch = 2; io.cap = cap
        io.d_kps = kp.data_ptr(); io.d_desc = ds.data_ptr(); io.d_counts = cn.data_ptr()
        c._check(c.lib.mo_dev_frontend_batch(c.h, C.byref(prm), C.byref(io)))
        torch.cuda.synchronize()
        assert c.dev_status() & 1
        for _ in range(4):
            c._check(c.lib.mo_dev_frontend_batch(c.h, C.byref(prm), C.byref(io)))
            torch.cuda.synchronize()
            if c.dev_status() == 0:
                break
        else:
            raise AssertionError("the slots did not grow")
        assert cn.cpu().tolist() == [len(ek), len(ek)]
        got = kp[1, :len(ek)].cpu().numpy().view(np.uint8).reshape(-1).view(V.KP_DTYPE)
        assert all(np.array_equal(got[f], ek[f]) for f in got.dtype.names) and np.array_equal(ds[1, :len(ek)].cpu().numpy(), ed)
    finally:
        c.close()
        O.lib().orc_set_variant(1, 0)


def test_tall_frame_in_a_single_frame_context(O):
    """A context for one frame at a time cuts the FAST strips to 2 rows (more workgroups, a shorter chain per call); a level may have at most
    2047 strips.  The tallest frame mo_create accepts, 4095 rows, has 2017 two-row strips on level 0 - just below that - and must give the
    oracle's result, as the same frame does in a context built for batches (8-row strips, 505 of them)."""
    import vslam_amd as V
    w, h = 256, 4095
    img = synthetic_frame(404, w, h)
    O.lib().orc_set_variant(0, 0)
    p, o = _prm(V, O, 0, nfeatures=3000, fast_threshold=9)
    ek, ed = O.detect_and_compute(img, o)
    assert len(ek) > 1500
    for max_batch in (1, 8):
        ctx = V.Context(device=0, max_w=w, max_h=h, max_batch=max_batch)
        try:
            (k, d), = ctx.orb_detect_compute(img, p)
        finally:
            ctx.close()
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(k[f], ek[f]), (max_batch, f)
        assert np.array_equal(d, ed), max_batch
    O.lib().orc_set_variant(1, 0)


def test_4k_frame(O):
    """3840 x 2160 (263 FAST strips on level 0: above the 256 the selection kernel's prefix table held until round 3; level 0 alone
    has several hundred thousand candidates): detect_and_compute, the grid detector and the matcher on the result, bit-exact."""
    import vslam_amd as V
    w, h = 3840, 2160
    ctx = V.Context(device=0, max_w=w, max_h=h, max_batch=1)
    img = synthetic_frame(78, w, h)
    try:
        O.lib().orc_set_variant(0, 0)
        p, o = _prm(V, O, 0, nfeatures=4000, fast_threshold=7)
        (kps, desc), = ctx.orb_detect_compute(img, p)
        ek, ed = O.detect_and_compute(img, o)
        assert len(kps) == len(ek) == 4000
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(kps[f], ek[f]), f
        assert np.array_equal(desc, ed)
        exy = O.grid_good_features(img, 2000)
        assert np.array_equal(ctx.grid_good_features(img, 2000), exy)
        idx, dist, ps = ctx.match_knn2_ratio(desc, desc[::-1].copy(), 0.75)
        eidx, edist = O.match_knn2(ed, ed[::-1].copy())
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    finally:
        O.lib().orc_set_variant(1, 0)
        ctx.close()
