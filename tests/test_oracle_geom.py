"""CPU: the numpy two-view oracle recovers the synthetic ground truth."""
import numpy as np

from oracle import geom_oracle as G


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def test_two_view_ground_truth():
    s = G.synthetic_two_view(seed=4096)
    r = G.init_two_view(s["p1"], s["p2"], s["K"], thr_px=3.0, n_hyp=1024, seed=4096)  # 1024 keeps the CPU test fast
    assert rel(r["R"], s["R"]) < 1e-4
    assert rel(r["t"], s["t"]) < 1e-4
    good = r["pose_mask"] & ~s["outlier"]
    assert good.sum() > 1200
    err = np.linalg.norm(r["X"][good] - s["X"][good], axis=1) / np.linalg.norm(s["X"][good], axis=1)
    assert err.max() < 1e-4
    assert (r["pose_mask"] & s["outlier"]).sum() < 30  # only chance inliers


def test_sampler_is_deterministic_and_distinct():
    a = G.sample8(4096, 7, 100)
    assert a == G.sample8(4096, 7, 100) and len(set(a)) == 8 and all(0 <= i < 100 for i in a)
    assert G.sample8(4096, 8, 100) != a
    assert sorted(G.sample8(1, 0, 8)) == list(range(8))


def test_too_few_points():
    s = G.synthetic_two_view(seed=1, n=7, outlier_frac=0)
    r = G.init_two_view(s["p1"], s["p2"], s["K"])
    assert r["n_good"] == 0 and r["R"] is None
