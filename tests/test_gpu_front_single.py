"""GPU parity of the single-frame pyramid + blur kernel (csrc/front_single.hip: one launch whose workgroups chain the levels of
their own tile through LDS): every raw and blurred level against the CPU oracle, bit-exact, and against the batched path's kernels
(k_resize2 / k_resize / k_blur) on frames too large for the oracle to finish quickly.  Calls go through the C-ABI probe
mo_dbg_pyramid_level (bit 1 of `blurred` selects the single-frame kernel); the end-to-end route of one- and two-frame calls is what
tests/test_gpu_orb.py and tests/test_gpu_frame_api.py already compare with the oracle.

Reference: cv2.ORB's pyramid + GaussianBlur behind /root/reference/src/orbslam2/extractor.py:50-67."""
import numpy as np
import pytest

from tests.helpers import synthetic_frame

pytestmark = pytest.mark.gpu

FUSED = 2


@pytest.fixture(scope="module")
def ctx():
    import vslam_amd as V
    c = V.Context(device=0, max_w=2048, max_h=2048, max_batch=4)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import orb_oracle
    return orb_oracle


def _levels(ctx, img, p, nlevels, route):
    return [(ctx.dbg_pyramid_level(img, p, L, blurred=route), ctx.dbg_pyramid_level(img, p, L, blurred=route | 1)) for L in range(nlevels)]


# (width, height), parameters; the geometries a tracker runs on must take the kernel (no silent fallback to the batched kernels)
ORACLE_CASES = [
    ((640, 480), dict(), True),
    ((478, 850), dict(), True),
    ((333, 257), dict(), True),     # width not a multiple of 4: level 0 is staged byte-wise
    ((65, 67), dict(), False),      # coarsest level 18 x 19: one or two tiles
    ((64, 64), dict(), False),
    ((666, 500), dict(scale_factor=2.0, nlevels=4), True),   # level ratio above 2 on one level (the gather kernel's case)
    ((640, 480), dict(scale_factor=1.1, nlevels=12), True),
    ((500, 375), dict(scale_factor=1.5, nlevels=3), True),
]


@pytest.mark.parametrize("size,kw,must_cover", ORACLE_CASES)
def test_levels_equal_the_oracle(ctx, O, size, kw, must_cover):
    import vslam_amd as V
    w, h = size
    img = synthetic_frame(31 + w, w, h)
    p = V.orb_params(select_order=V.ORDER_LIBSTDCXX, **kw)
    o = O.params(**kw)
    nl = kw.get("nlevels", 8)
    try:
        got = _levels(ctx, img, p, nl, FUSED)
    except V.NativeError as e:  # MO_ERR_UNSUPPORTED: a geometry the tile boxes do not cover keeps the batched kernels
        assert not must_cover, "the single-frame kernel must cover %dx%d %r: %s" % (w, h, kw, e)
        pytest.skip("not covered by the single-frame kernel: %s" % e)
    for L in range(nl):
        raw, blr = got[L]
        exp = O.pyramid_level(img, o, L)
        assert raw.shape == exp.shape
        if L > 0:  # (level 0 is the input itself)
            assert np.array_equal(raw, exp), "raw level %d: %d pixels differ" % (L, int((raw != exp).sum()))
        expb = O.pyramid_level(img, o, L, blurred=True)
        assert np.array_equal(blr, expb), "blurred level %d: %d pixels differ" % (L, int((blr != expb).sum()))


@pytest.mark.parametrize("size", [(1280, 720), (1920, 1080), (1022, 770)])
def test_levels_equal_the_batched_kernels_on_large_frames(ctx, size):
    import vslam_amd as V
    w, h = size
    img = synthetic_frame(5 + h, w, h)
    p = V.orb_params(select_order=V.ORDER_LIBSTDCXX)
    ref = _levels(ctx, img, p, 8, 0)
    got = _levels(ctx, img, p, 8, FUSED)
    for L in range(8):
        if L > 0:
            assert np.array_equal(got[L][0], ref[L][0]), "raw level %d" % L
        assert np.array_equal(got[L][1], ref[L][1]), "blurred level %d" % L


def test_one_and_two_frame_calls_equal_a_larger_batch(ctx):
    """Calls on one or two frames take the single-frame kernel, larger batches k_resize2 + k_blur: the same frames must give the
    same keypoints and descriptors either way."""
    import vslam_amd as V
    frames = [synthetic_frame(900 + i) for i in range(3)]
    p = V.orb_params(select_order=V.ORDER_LIBSTDCXX, nfeatures=1500)
    big = ctx.orb_detect_compute(np.stack(frames), p)
    two = ctx.orb_detect_compute(np.stack(frames[:2]), p)
    (one_k, one_d), = ctx.orb_detect_compute(frames[2], p)
    for i, (k, d) in enumerate(list(two) + [(one_k, one_d)]):
        bk, bd = big[i]
        assert len(k) == len(bk) > 500
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(k[f], bk[f]), (i, f)
        assert np.array_equal(d, bd), i
