"""CPU-side checks of the boundary: the library loads, exports every symbol include/vslam_amd.h declares,
fails loudly without a GPU, and the host-side list logic of the drop-in classes behaves like the reference's."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    import vslam_amd as V
    lib = V.load_library()
    hdr = open(os.path.join(ROOT, "include", "vslam_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(mo_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert names == set(V.SIGNATURES), names ^ set(V.SIGNATURES)


def test_struct_layouts():
    import ctypes as C
    import vslam_amd as V
    assert V.KP_DTYPE.itemsize == 28
    assert C.sizeof(V.OrbParams) == 40


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof of every struct that crosses the boundary as a C compiler sees include/vslam_amd.h == the ctypes mirror
    (mo_batch_io gets fields appended per round; mo_frame_ref / mo_pair_params / mo_pair_out are the one-frame-at-a-time API)."""
    import ctypes as C
    import subprocess
    import vslam_amd as V
    structs = [("mo_batch_io", V.BatchIO), ("mo_orb_params", V.OrbParams), ("mo_frame_ref", V.FrameRef), ("mo_pair_params", V.PairParams),
               ("mo_pair_out", V.PairOut), ("mo_stream_params", V.StreamParams), ("mo_stream_result", V.StreamResult)]
    body = ""
    for cname, cls in structs:
        body += '  printf("%%zu\\n", sizeof(%s));\n' % cname
        body += "".join('  printf("%%zu\\n", offsetof(%s, %s));\n' % (cname, f[0]) for f in cls._fields_)
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "vslam_amd.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    k = 0
    for cname, cls in structs:
        assert got[k] == C.sizeof(cls), cname
        offs = [getattr(cls, f[0]).offset for f in cls._fields_]
        assert got[k + 1:k + 1 + len(offs)] == offs, cname
        k += 1 + len(offs)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_fails_loudly_without_gpu():
    import vslam_amd as V
    assert V.device_count() == 0
    with pytest.raises(V.NativeUnavailable):
        V.Context()
    from orbslam2.extractor import ORBExtractor
    with pytest.raises(V.NativeUnavailable):
        ORBExtractor(n_features=100).detect_and_compute(np.zeros((480, 640), np.uint8))


def test_matcher_host_logic():
    from orbslam2.matcher import DescriptorMatcher
    from orbslam2.types import DMatch, KeyPoint
    m = DescriptorMatcher(ratio_threshold=0.8)
    assert m.match(None, np.zeros((3, 32), np.uint8)) == []
    assert m.match(np.zeros((0, 32), np.uint8), np.zeros((3, 32), np.uint8)) == []
    with pytest.raises(ValueError):
        DescriptorMatcher("sift")
    kp1 = [KeyPoint(0, 0, 31), KeyPoint(10, 10, 31), KeyPoint(100, 100, 31)]
    kp2 = [KeyPoint(3, 4, 31), KeyPoint(200, 10, 31), KeyPoint(101, 100, 31)]
    ms = [DMatch(0, 0, 0, 10.0), DMatch(1, 1, 0, 30.0), DMatch(2, 2, 0, 20.0)]
    kept = m.filter_matches_by_geometric_distance(kp1, kp2, ms, 0.02, (480, 640))  # limit 11.2 px
    assert [k.queryIdx for k in kept] == [0, 2]
    by_d = m.filter_matches_by_distance(ms)  # median 20 -> threshold 40, sorted by distance
    assert [k.distance for k in by_d] == [10.0, 20.0, 30.0]
    assert [k.distance for k in m.filter_matches_by_distance(ms, 25.0)] == [10.0, 20.0]
    assert m.filter_matches_by_distance([]) == []


def test_initializer_guards():
    from orbslam2.initializer import MapInitializer
    from orbslam2.utils import compute_projection_matrix, convert_to_3d_points
    ini = MapInitializer(np.eye(3), min_matches=10)
    assert ini.initialize([], None, None, None) == (False, None, None, None, None)
    P = compute_projection_matrix(np.eye(3), np.array([1.0, 2.0, 3.0]), np.diag([2.0, 2.0, 1.0]))
    assert P.shape == (3, 4) and np.allclose(P[:, 3], [2, 4, 3])
    X = convert_to_3d_points(np.array([[2.0, 4], [4, 8], [6, 12], [2, 4]], np.float32))
    assert np.allclose(X, [[1, 2, 3], [1, 2, 3]])


def test_option_b_shadowing(tmp_path):
    """INTEGRATION.md Option B: with <repo>/visual-slam_amd ahead of the reference's src on sys.path, a tracker module living in the
    REFERENCE's orbslam2 directory resolves its relative imports (tracker.py:9-11) to this repo's modules.  The reference tree is
    imitated by a stand-in with the same shape (no __init__.py, a tracker with the three relative imports, and decoy modules that
    must lose)."""
    import subprocess
    import sys
    ref = tmp_path / "src" / "orbslam2"
    ref.mkdir(parents=True)
    (ref / "__init.py__").write_text("")  # the reference's misnamed init file: the directory is a namespace portion
    for name in ("extractor", "matcher", "initializer", "utils"):
        (ref / (name + ".py")).write_text("DECOY = True\nclass ORBExtractor: pass\nclass DescriptorMatcher: pass\nclass MapInitializer: pass\n")
    (ref / "tracker.py").write_text("from .extractor import ORBExtractor\nfrom .matcher import DescriptorMatcher\n"
                                    "from .initializer import MapInitializer\nfrom .utils import compute_projection_matrix\n")
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import orbslam2.tracker as t, orbslam2.extractor as e\n"
            "assert t.__file__.startswith(%r), t.__file__\n"
            "assert not hasattr(e, 'DECOY') and 'visual-slam_amd' in e.__file__, e.__file__\n"
            "assert t.ORBExtractor is e.ORBExtractor and 'visual-slam_amd' in sys.modules[t.DescriptorMatcher.__module__].__file__\n"
            "print('ok')" % (os.path.join(ROOT, "visual-slam_amd"), str(tmp_path / "src"), str(ref)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_keypoints_from_array_fast_path_equals_constructor():
    """orbslam2.types.keypoints_from_array fills the slots of the cv2.KeyPoint stand-in directly (no __init__ conversions): same
    objects, field for field and type for type, as KeyPoint(*row); round trip through keypoints_to_array is the identity."""
    import vslam_amd as V
    from orbslam2.types import HAVE_CV2, KeyPoint, keypoints_from_array, keypoints_to_array
    rng = np.random.default_rng(5)
    arr = np.zeros(257, V.KP_DTYPE)
    arr["x"] = rng.uniform(0, 640, 257).astype(np.float32); arr["y"] = rng.uniform(0, 480, 257).astype(np.float32)
    arr["size"] = 31 * 1.2 ** rng.integers(0, 8, 257); arr["angle"] = rng.uniform(0, 360, 257); arr["response"] = rng.uniform(0, 1e-2, 257)
    arr["octave"] = rng.integers(0, 8, 257); arr["class_id"] = -1
    got = keypoints_from_array(arr)
    assert len(got) == 257 and len(tuple(got)) == 257
    for k, row in zip(got, arr.tolist()):
        ref = KeyPoint(*row)
        for f in ("pt", "size", "angle", "response", "octave", "class_id"):
            a, b = getattr(k, f), getattr(ref, f)
            assert a == b and type(a) is type(b), f
    if not HAVE_CV2:
        assert type(got[0].pt[0]) is float and type(got[0].octave) is int
    assert np.array_equal(keypoints_to_array(got), arr)
    assert keypoints_from_array(arr[:0]) == ()


def test_lazy_keypoint_sequence_behaves_like_the_tuple():
    """orbslam2.types.KeyPointSeq (what detect_and_compute returns): a sequence that creates KeyPoint objects on demand.  Same
    answers as the materialised tuple for len / index / negative index / slice / iteration / == / +; an object handed out keeps its
    identity and its attribute writes (they reach keypoints_to_array and points_of); an untouched sequence converts back without
    creating objects."""
    from collections.abc import Sequence
    import vslam_amd as V
    from orbslam2.types import KeyPointSeq, keypoints_from_array, keypoints_to_array, points_of
    rng = np.random.default_rng(6)
    arr = np.zeros(50, V.KP_DTYPE)
    arr["x"] = rng.uniform(0, 640, 50).astype(np.float32); arr["y"] = rng.uniform(0, 480, 50).astype(np.float32)
    arr["size"] = 31; arr["angle"] = rng.uniform(0, 360, 50); arr["octave"] = rng.integers(0, 8, 50); arr["class_id"] = -1
    seq = keypoints_from_array(arr.copy())
    assert isinstance(seq, (KeyPointSeq, Sequence)) and seq.pristine and len(seq) == 50
    assert keypoints_to_array(seq) is seq.array                       # no objects, no copy
    idx = [3, 7, 7, 49, 0]
    want = np.stack([arr["x"][idx], arr["y"][idx]], axis=1)
    assert np.array_equal(points_of(seq, idx), want) and points_of(seq, idx).dtype == np.float32 and seq.pristine
    assert points_of(seq, []).shape == (0, 2)
    k7 = seq[7]
    assert k7 is seq[7] and k7 is seq[-43] and not seq.pristine       # identity per index
    assert k7.pt == (float(arr["x"][7]), float(arr["y"][7])) and k7.octave == int(arr["octave"][7])
    with pytest.raises(IndexError):
        seq[50]
    with pytest.raises(IndexError):
        seq[-51]
    k7.class_id = 1234                                                # a caller writes to its object
    assert np.array_equal(points_of(seq, idx), want)                  # (object path now: same values)
    back = keypoints_to_array(seq)
    assert back["class_id"][7] == 1234 and np.array_equal(back["x"], arr["x"]) and np.array_equal(back["octave"], arr["octave"])
    full = tuple(seq)
    assert full[7] is k7 and len(full) == 50 and seq[7] is k7          # materialising keeps the handed-out object
    assert seq[2:5] == full[2:5] and seq[::-1] == full[::-1]
    assert seq == full and seq != keypoints_from_array(arr.copy())  # (index 7 was written to)
    assert (seq + (1, 2))[-2:] == (1, 2) and ((1,) + seq)[0] == 1 and len(seq + seq) == 100
    assert [k.octave for k in seq] == arr["octave"].tolist()
    assert sum(1 for _ in reversed(seq)) == 50 and full[3] in seq and seq.index(full[3]) == 3


def test_lazy_keypoint_list_behaves_like_the_list(monkeypatch):
    """orbslam2.types.KeyPointList (what distribute_keypoints / extract_features(distributed=True) return for ALL corners; the reference
    builds a Python list there, extractor.py:133): the lazy sequence answering like a list for everything the reference's callers do
    with it (len, indexing, iteration: tracker.py:238-239, tests/test_orb_extractor.py:84-90) plus ==, + and slices; the plain list
    on request."""
    from orbslam2.types import KeyPointList, keypoints_at, keypoints_at_lazy, keypoints_to_array
    xy = np.array([[1.5, 2.5], [10, 20], [30, 40], [7, 9]], np.float32)
    lazy, plain = keypoints_at_lazy(xy, 31), keypoints_at(xy, 31)
    assert isinstance(lazy, KeyPointList) and type(plain) is list and len(lazy) == 4 and lazy.pristine
    assert keypoints_to_array(lazy) is lazy.array                     # no objects, no copy
    k = lazy[1]
    assert k is lazy[1] and k is lazy[-3] and not lazy.pristine
    assert (k.pt, k.size, k.angle, k.response, k.octave, k.class_id) == ((10.0, 20.0), 31.0, -1.0, 0.0, 0, -1)
    assert [(q.pt, q.size, q.angle, q.octave, q.class_id) for q in lazy] == [(q.pt, q.size, q.angle, q.octave, q.class_id) for q in plain]
    assert type(lazy[1:3]) is list and lazy[1:3][0] is k and type(lazy + [1]) is list and type([1] + lazy) is list and len(lazy + lazy) == 8
    assert lazy == list(lazy) and not (lazy == tuple(lazy))           # a list equals lists only
    with pytest.raises(IndexError):
        lazy[4]
    with pytest.raises(TypeError):
        hash(lazy)
    empty = keypoints_at_lazy(np.zeros((0, 2), np.float32), 31)
    assert empty == [] and len(empty) == 0 and list(empty) == [] and not empty
    monkeypatch.setenv("VSLAM_AMD_KEYPOINTS", "tuple")
    assert type(keypoints_at_lazy(xy, 31)) is list


def test_lazy_match_list_behaves_like_the_list(monkeypatch):
    """orbslam2.types.DMatchList (what DescriptorMatcher.match, track_from_last_frame and MapInitializer.initialize return; the reference
    builds Python lists of cv2.DMatch: matcher.py:60-107): len / index / slice / iteration / == / + like a list, objects on demand with a
    stable identity, the arrays for the drop-in's own filters - whose array form must keep the reference's order (sorted() is stable)."""
    from orbslam2.types import DMatchList, dmatches_from_arrays, match_arrays
    rng = np.random.default_rng(3)
    n = 200
    q, t, d = rng.permutation(500)[:n], rng.integers(0, 500, n), rng.integers(0, 40, n).astype(np.float64)   # (many equal distances)
    lazy = dmatches_from_arrays(q, t, d)
    assert isinstance(lazy, DMatchList) and len(lazy) == n and lazy.pristine and match_arrays(lazy)[0] is lazy.q and lazy.pristine
    m = lazy[5]
    assert m is lazy[5] and m is lazy[-195] and not lazy.pristine
    assert (m.queryIdx, m.trainIdx, m.imgIdx, m.distance) == (int(q[5]), int(t[5]), 0, float(d[5])) and type(m.queryIdx) is int and type(m.distance) is float
    full = list(lazy)
    assert full[5] is m and lazy == full and not (lazy == tuple(full)) and type(lazy[2:6]) is list and lazy[2:6] == full[2:6]
    assert type(lazy + [1]) is list and type([1] + lazy) is list and len(lazy + lazy) == 2 * n
    qa, ta, da = match_arrays(lazy)                                   # (objects were handed out: read off them)
    assert np.array_equal(qa, q) and np.array_equal(ta, t) and np.array_equal(da, d)
    with pytest.raises(IndexError):
        lazy[n]
    with pytest.raises(TypeError):
        hash(lazy)
    empty = dmatches_from_arrays([], [], [])
    assert empty == [] and not empty and list(empty) == []
    # filter_matches_by_distance on arrays == the reference's sorted(...) / median / comprehension on objects
    ref_sorted = sorted(full, key=lambda x: x.distance)
    thr = np.median([x.distance for x in ref_sorted]) * 2.0
    ref = [x for x in ref_sorted if x.distance < thr]
    fresh = dmatches_from_arrays(q, t, d)
    order = np.argsort(fresh.d, kind="stable")
    got = [fresh[i] for i in order[fresh.d[order] < np.median(fresh.d) * 2.0].tolist()]
    assert [(x.queryIdx, x.trainIdx, x.distance) for x in got] == [(x.queryIdx, x.trainIdx, x.distance) for x in ref] and len(ref) > 50
    monkeypatch.setenv("VSLAM_AMD_KEYPOINTS", "tuple")
    assert type(dmatches_from_arrays(q, t, d)) is list


def test_keypoints_as_plain_tuple_on_request(monkeypatch):
    import vslam_amd as V
    from orbslam2.types import keypoints_from_array
    monkeypatch.setenv("VSLAM_AMD_KEYPOINTS", "tuple")
    got = keypoints_from_array(np.zeros(3, V.KP_DTYPE))
    assert type(got) is tuple and len(got) == 3
