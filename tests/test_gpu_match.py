"""GPU parity of the brute-force Hamming 2-NN + ratio test vs the CPU oracle (bit-exact indices/distances)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["valu", "mfma"])
def ctx(request):
    """Both matcher kernels: the default XOR + popcount one and the opt-in matrix-core one (chosen at context creation)."""
    import os
    import vslam_amd as V
    old = os.environ.get("VSLAM_AMD_MATCHER")
    os.environ["VSLAM_AMD_MATCHER"] = request.param
    try:
        c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=4)
    finally:
        if old is None:
            os.environ.pop("VSLAM_AMD_MATCHER", None)
        else:
            os.environ["VSLAM_AMD_MATCHER"] = old
    yield c
    c.close()


def _check(ctx, q, t, ratio):
    from oracle import orb_oracle as O
    idx, dist, ps = ctx.match_knn2_ratio(q, t, ratio)
    eidx, edist = O.match_knn2(q, t)
    assert np.array_equal(idx, eidx)
    assert np.array_equal(dist, edist)
    eps = O.ratio_test(eidx, edist, ratio if ratio is not None else 1.0, enabled=ratio is not None)
    assert np.array_equal(ps, eps)


@pytest.mark.parametrize("nq,nt", [(2000, 2000), (1, 1), (5, 1), (1, 7), (257, 513), (2000, 3)])
@pytest.mark.parametrize("ratio", [0.75, 0.85, 0.0, None])  # 0.0 is a threshold (nothing with two neighbours passes), None = test off
def test_random_descriptors(ctx, nq, nt, ratio):
    rng = np.random.default_rng(nq * 131 + nt)
    _check(ctx, rng.integers(0, 256, (nq, 32), dtype=np.uint8), rng.integers(0, 256, (nt, 32), dtype=np.uint8), ratio)


def test_ties_and_duplicates(ctx):
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    t = np.concatenate([base, base, base[::-1]])  # every train descriptor appears 3 times -> distance ties
    q = base.copy()
    q[:, 0] ^= 1
    _check(ctx, q, t, 0.75)
    _check(ctx, np.zeros((10, 32), np.uint8), np.zeros((10, 32), np.uint8), 0.75)
    _check(ctx, np.zeros((3, 32), np.uint8), np.full((4, 32), 255, np.uint8), None)  # distance 256


def test_self_match_config2(ctx):
    """BASELINE config 2: 2000-descriptor self match -> best is self at distance 0."""
    rng = np.random.default_rng(2)
    d = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
    idx, dist, ps = ctx.match_knn2_ratio(d, d, 0.75)
    assert np.array_equal(idx[:, 0], np.arange(2000)) and (dist[:, 0] == 0).all() and ps.all()
    _check(ctx, d, d, 0.75)


def test_batched(ctx):
    rng = np.random.default_rng(3)
    q = rng.integers(0, 256, (3, 300, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (3, 200, 32), dtype=np.uint8)
    idx, dist, ps = ctx.match_knn2_ratio(q, t, 0.8)
    for b in range(3):
        i1, d1, p1 = ctx.match_knn2_ratio(q[b], t[b], 0.8)
        assert np.array_equal(idx[b], i1) and np.array_equal(dist[b], d1) and np.array_equal(ps[b], p1)


def test_property_random_shapes_and_ties(ctx):
    """hypothesis: random sizes incl. nt in {1, 2}, heavy duplicates -> (idx, dist, pass) equal the oracle bit for bit."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 300), st.integers(1, 300), st.integers(1, 8), st.sampled_from([0.6, 0.75, 0.85, 1.0]),
           st.integers(0, 2 ** 31 - 1))
    def run(nq, nt, pool, ratio, seed):
        rng = np.random.default_rng(seed)
        words = rng.integers(0, 256, (pool, 32), dtype=np.uint8)       # few distinct descriptors -> many exact ties
        q = words[rng.integers(0, pool, nq)].copy()
        t = words[rng.integers(0, pool, nt)].copy()
        flip = rng.random(nq) < 0.5
        q[flip, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        _check(ctx, q, t, ratio)

    run()


@pytest.mark.parametrize("nq,nt", [(700, 1000), (100, 4097), (513, 129), (16000, 300), (17000, 300)])
def test_sliced_train_set_ties(ctx, nq, nt):
    """Few workgroups -> the default kernel slices the train tiles over gridDim.z and merges per-slice keys (match_launch_pairs);
    a pool of 5 distinct descriptors puts exact distance ties in every slice, so the merged (idx, dist) must still be the
    lowest-index ones.  16000 queries is the last size that slices (32 workgroups), 17000 the first that does not."""
    rng = np.random.default_rng(nq + nt)
    words = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    q = words[rng.integers(0, 5, nq)].copy()
    t = words[rng.integers(0, 5, nt)].copy()
    q[::3, 7] ^= 4
    _check(ctx, q, t, 0.75)
    _check(ctx, rng.integers(0, 256, (nq, 32), dtype=np.uint8), rng.integers(0, 256, (nt, 32), dtype=np.uint8), None)
