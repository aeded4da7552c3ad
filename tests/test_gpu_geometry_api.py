"""GPU: the remaining geometry entry points of the drop-in utils module - recover_pose on ANY essential matrix
(reference utils.py:129-134) and undistort_image (utils.py:40-52) - against the numpy oracle."""
import numpy as np
import pytest

from oracle import geom_oracle as G
from tests.helpers import synthetic_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import vslam_amd as V
    c = V.Context(device=0, max_w=1024, max_h=1024, max_batch=1)
    yield c
    c.close()


def test_recover_pose_accepts_any_essential_matrix(ctx):
    from orbslam2 import utils as geom
    s = G.synthetic_two_view(seed=8, n=400, outlier_frac=0.2)
    t = s["t"].ravel()
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ s["R"] * 3.7  # ground-truth essential matrix, arbitrary scale: NOT something calculate_essential_matrix returned
    mask = (~s["outlier"]).astype(np.uint8).reshape(-1, 1)
    n, R, tt, m = geom.recover_pose(E, s["p1"], s["p2"], s["K"], mask)
    on, oR, ot, om = G.recover_pose(E, s["p1"], s["p2"], s["K"], ~s["outlier"])
    assert np.linalg.norm(R - oR) < 1e-9 and np.linalg.norm(tt - ot) < 1e-9
    assert np.linalg.norm(R - s["R"]) < 1e-9 and np.linalg.norm(tt - s["t"]) < 1e-9
    assert m.shape == (400, 1) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 255}
    assert ((m.ravel() != 0) != om).sum() <= 1 and abs(n - on) <= 1 and n > 300
    assert not (m.ravel() != 0)[s["outlier"]].any()  # the input mask is honoured
    # negated E and no mask: same rotation, cheirality still picks the camera-in-front solution
    n2, R2, t2, m2 = geom.recover_pose(-E, s["p1"], s["p2"], s["K"])
    assert np.linalg.norm(R2 - s["R"]) < 1e-9 and np.linalg.norm(t2 - s["t"]) < 1e-9 and n2 >= n
    # interleaved pairs: each call stands alone
    s2 = G.synthetic_two_view(seed=9, n=300, outlier_frac=0.0)
    Ea, ma = geom.calculate_essential_matrix(s["p1"], s["p2"], s["K"], threshold=3.0)
    Eb, mb = geom.calculate_essential_matrix(s2["p1"], s2["p2"], s2["K"], threshold=3.0)
    _, Ra, ta, _ = geom.recover_pose(Ea, s["p1"], s["p2"], s["K"], ma)
    _, Rb, tb, _ = geom.recover_pose(Eb, s2["p1"], s2["p2"], s2["K"], mb)
    assert np.linalg.norm(Ra - s["R"]) < 1e-4 and np.linalg.norm(Rb - s2["R"]) < 1e-4 and np.linalg.norm(ta - s["t"]) < 1e-4
    # the raw context call also returns the triangulated points
    r = ctx.recover_pose(E, s["p1"], s["p2"], s["K"], mask)
    good = r["mask"]
    e = np.linalg.norm(r["X"][good] - s["X"][good], axis=1) / np.linalg.norm(s["X"][good], axis=1)
    assert e.max() < 1e-4 and np.isnan(r["X"][~good]).all()


@pytest.mark.parametrize("dist", [(0.0, 0.0, 0.0, 0.0, 0.0), (-0.28, 0.07, 0.0002, -0.0003, 0.01), (0.12, -0.2, 0.001, 0.002, 0.05)])
def test_undistort_equals_oracle(ctx, dist):
    from orbslam2 import utils as geom
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])  # configs/monocular.yaml:3
    gray = synthetic_frame(3)
    out = geom.undistort_image(gray, K, np.array(dist))
    assert out.shape == gray.shape and out.dtype == np.uint8
    assert np.array_equal(out, G.undistort(gray, K, dist))
    if not any(dist):
        assert np.array_equal(out, gray)  # zero coefficients: the identity map (the reference skips the call, run_video.py:148)
    else:
        assert not np.array_equal(out, gray)
    bgr = np.stack([gray, np.roll(gray, 5, 0), np.roll(gray, 7, 1)], axis=2)
    assert np.array_equal(geom.undistort_image(bgr, K, np.array(dist)), G.undistort(bgr, K, dist))
    odd = np.ascontiguousarray(gray[:333, :257])
    Ko = np.array([[200.0, 0, 130.5], [0, 210.0, 160.25], [0, 0, 1.0]])
    assert np.array_equal(ctx.undistort(odd, Ko, dist), G.undistort(odd, Ko, dist))


def test_undistort_on_device_ahead_of_the_extractor(ctx):
    """mo_dev_undistort feeding mo_dev_orb_detect_compute on the same stream == host undistort + host extract."""
    import ctypes as C
    import torch
    import vslam_amd as V
    K = np.array([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]])
    dist = np.array([-0.2, 0.05, 0.0, 0.0, 0.0])
    frames = np.stack([synthetic_frame(60 + i) for i in range(4)])
    dev = torch.device("cuda", 0)
    c2 = V.Context(device=0, max_w=640, max_h=480, max_batch=4)
    c2.set_stream(torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(frames).to(dev); d_ud = torch.empty_like(d_in)
    Kc = np.ascontiguousarray(K.reshape(9)); dc = np.ascontiguousarray(dist)
    c2._check(c2.lib.mo_dev_undistort(c2.h, d_in.data_ptr(), 640, 480, 1, 4, Kc.ctypes.data, dc.ctypes.data, d_ud.data_ptr()))
    cap = 1024
    prm = V.orb_params(nfeatures=500)
    kps = torch.zeros((4, cap, 7), dtype=torch.float32, device=dev); desc = torch.zeros((4, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    c2._check(c2.lib.mo_dev_orb_detect_compute(c2.h, C.byref(prm), d_ud.data_ptr(), 640, 480, 4, kps.data_ptr(), desc.data_ptr(), cap,
                                               counts.data_ptr()))
    torch.cuda.synchronize()
    assert c2.dev_status() == 0
    for i in (0, 3):
        ud = G.undistort(frames[i], K, dist)
        assert np.array_equal(d_ud[i].cpu().numpy(), ud)
        (k, d), = ctx.orb_detect_compute(ud, prm)
        n = int(counts[i].item())
        assert n == len(k) and np.array_equal(desc[i, :n].cpu().numpy(), d)
    c2.close()
