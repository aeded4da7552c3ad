"""cv2.KeyPoint / cv2.DMatch stand-ins.

When cv2 is importable the real classes are used (the caller's drawKeypoints / drawMatches need them); otherwise
duck-typed objects with the same attributes and constructor argument order."""
import os as _os
from collections.abc import Sequence as _Sequence

try:  # pragma: no cover - cv2 is absent in the build container
    import cv2 as _cv2
    KeyPoint = _cv2.KeyPoint
    DMatch = _cv2.DMatch
    HAVE_CV2 = True
except Exception:
    HAVE_CV2 = False

    class KeyPoint:
        __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

        def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
            self.pt = (float(x), float(y))
            self.size = float(size)
            self.angle = float(angle)
            self.response = float(response)
            self.octave = int(octave)
            self.class_id = int(class_id)

        def __repr__(self):
            return "KeyPoint(pt=%r, size=%g, angle=%g, response=%g, octave=%d)" % (
                self.pt, self.size, self.angle, self.response, self.octave)

    class DMatch:
        __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

        def __init__(self, queryIdx=-1, trainIdx=-1, imgIdx=0, distance=float("inf")):
            # cv2.DMatch(queryIdx, trainIdx, distance) and (queryIdx, trainIdx, imgIdx, distance) both exist
            self.queryIdx = int(queryIdx)
            self.trainIdx = int(trainIdx)
            self.imgIdx = int(imgIdx)
            self.distance = float(distance)

        def __repr__(self):
            return "DMatch(%d -> %d, %g)" % (self.queryIdx, self.trainIdx, self.distance)


def _materialize(arr):
    """structured mo_keypoint array -> tuple of KeyPoint objects"""
    # one bulk conversion to Python scalars (tolist) instead of seven numpy scalar reads per keypoint: 3 ms -> 0.7 ms for 2000
    rows = arr.tolist()
    if HAVE_CV2:
        return tuple(KeyPoint(*t) for t in rows)
    # the stand-in class: tolist() already yields Python floats / ints, so the slots are filled directly instead of through
    # __init__'s seven conversions (another quarter of the time for 2000 keypoints)
    new, out = object.__new__, []
    append = out.append
    for x, y, size, angle, response, octave, class_id in rows:
        k = new(KeyPoint)
        k.pt = (x, y); k.size = size; k.angle = angle; k.response = response; k.octave = octave; k.class_id = class_id
        append(k)
    return tuple(out)


class KeyPointSeq(_Sequence):
    """The keypoints of one detectAndCompute call: an immutable sequence that behaves like the tuple of cv2.KeyPoint objects the
    reference gets (len, indexing, slicing, iteration, ==, + with tuples), but builds a KeyPoint object only when one is asked for.
    2000 Python objects cost 0.46 ms per frame - more than the device work (0.27 ms) - and a tracker touches only the matched ones;
    the drop-in classes themselves (compute, the match filters, initialize, track_from_last_frame) read `.array`, the structured
    mo_keypoint records, without creating any object.  An object handed out once stays the object of its index (identity and
    attribute writes are kept); `tuple(seq)` gives the plain tuple."""
    __slots__ = ("array", "_objs", "_some")

    def __init__(self, array):
        self.array = array      # [n] KP_DTYPE, owned by this sequence
        self._objs = None       # tuple of all objects once a caller iterated
        self._some = None       # {index: object} handed out one by one before that

    def __len__(self):
        return len(self.array)

    def _all(self):
        if self._objs is None:
            objs = _materialize(self.array)
            if self._some:
                objs = list(objs)
                for i, k in self._some.items():
                    objs[i] = k
                objs = tuple(objs)
            self._objs, self._some = objs, None
        return self._objs

    def __getitem__(self, i):
        if self._objs is not None:
            return self._objs[i]
        if isinstance(i, slice):
            return self._all()[i]
        n = len(self.array)
        j = int(i)
        if j < 0:
            j += n
        if not 0 <= j < n:
            raise IndexError("tuple index out of range")
        if self._some is None:
            self._some = {}
        k = self._some.get(j)
        if k is None:
            k = self._some[j] = KeyPoint(*self.array[j].item())
        return k

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        return self._all() == (other._all() if isinstance(other, KeyPointSeq) else other)

    def __hash__(self):
        return hash(self._all())

    def __add__(self, other):
        return self._all() + tuple(other)

    def __radd__(self, other):
        return tuple(other) + self._all()

    def __repr__(self):
        return "KeyPointSeq(%d keypoints)" % len(self.array)

    @property
    def pristine(self):
        """no object was handed out: `.array` is the whole truth (an object a caller holds may have been written to)"""
        return self._objs is None and not self._some


class KeyPointList(KeyPointSeq):
    """distribute_keypoints' list of ALL corners.  The reference builds a Python list of cv2.KeyPoint (extractor.py:133) and its
    callers index it, take its length and hand it on (tracker.py:87-146, 238-239; tests/test_orb_extractor.py:84-90): this is the same
    lazy sequence answering like a list (== with lists, + and slices give lists; `list(seq)` is the plain list) - 1 400 objects per
    frame cost 0.29 ms, more than the whole device call (0.11 ms).  What a list has and this has not: it cannot be modified in place."""
    __slots__ = ()

    def __getitem__(self, i):
        r = KeyPointSeq.__getitem__(self, i)
        return list(r) if isinstance(i, slice) else r

    def __eq__(self, other):
        if isinstance(other, KeyPointSeq):
            return self._all() == other._all()
        return isinstance(other, list) and list(self._all()) == other

    __hash__ = None  # (like a list)

    def __add__(self, other):
        return list(self._all()) + list(other)

    def __radd__(self, other):
        return list(other) + list(self._all())

    def __repr__(self):
        return "KeyPointList(%d keypoints)" % len(self.array)


def keypoints_at_lazy(xy, size=31.0):
    """keypoints_at as a KeyPointList over records (the plain list with VSLAM_AMD_KEYPOINTS=tuple, like keypoints_from_array)"""
    if _os.environ.get("VSLAM_AMD_KEYPOINTS", "lazy").lower() == "tuple":
        return keypoints_at(xy, size)
    import numpy as np
    from vslam_amd import KP_DTYPE
    rec = np.zeros(len(xy), KP_DTYPE)
    if len(xy):
        rec["x"], rec["y"] = xy[:, 0], xy[:, 1]
    rec["size"], rec["angle"], rec["class_id"] = float(size), -1.0, -1
    return KeyPointList(rec)


def keypoints_from_array(arr):
    """structured mo_keypoint array -> what detectAndCompute returns (cv2: a tuple of KeyPoint; here the lazy KeyPointSeq, or the
    plain tuple with VSLAM_AMD_KEYPOINTS=tuple for a caller that insists on the type)"""
    if _os.environ.get("VSLAM_AMD_KEYPOINTS", "lazy").lower() == "tuple":
        return _materialize(arr)
    return KeyPointSeq(arr)


def keypoints_at(xy, size=31.0):
    """[KeyPoint(x, y, size) for x, y in xy] (the reference's distribute_keypoints list, extractor.py:133) with the stand-in's slots
    filled directly: cv2's defaults angle -1, response 0, octave 0, class_id -1"""
    rows = xy.tolist()
    if HAVE_CV2:
        return [KeyPoint(x, y, size) for x, y in rows]
    new, out, size = object.__new__, [], float(size)
    append = out.append
    for x, y in rows:
        k = new(KeyPoint)
        k.pt = (x, y); k.size = size; k.angle = -1.0; k.response = 0.0; k.octave = 0; k.class_id = -1
        append(k)
    return out


def keypoints_to_array(kps):
    import numpy as np
    from vslam_amd import KP_DTYPE
    if isinstance(kps, KeyPointSeq) and kps.pristine:
        return kps.array
    return np.array([(k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id) for k in kps], KP_DTYPE).reshape(-1)


def points_of(kps, indices):
    """float32 [n, 2] of kps[i].pt for i in indices (the reference's np.float32([kps[m.queryIdx].pt for m in matches]))"""
    import numpy as np
    if isinstance(kps, KeyPointSeq) and kps.pristine:
        a = kps.array[np.asarray(indices, dtype=np.intp)]
        return np.stack([a["x"], a["y"]], axis=1).astype(np.float32).reshape(-1, 2)
    return np.float32([kps[i].pt for i in indices]).reshape(-1, 2)


def _dmatch_objects(q, t, d):
    """lists of Python scalars -> list of DMatch(queryIdx, trainIdx, 0, distance); the stand-in class gets its slots filled directly
    (a quarter less time than 1000 __init__ calls with their int() / float())"""
    if HAVE_CV2:
        return [DMatch(a, b, 0, c) for a, b, c in zip(q, t, d)]
    new, out = object.__new__, []
    append = out.append
    for a, b, c in zip(q, t, d):
        m = new(DMatch)
        m.queryIdx = a; m.trainIdx = b; m.imgIdx = 0; m.distance = c
        append(m)
    return out


class DMatchList(_Sequence):
    """The matches of one call (DescriptorMatcher.match, the tracking helpers, MapInitializer): the reference builds a Python list of cv2.DMatch;
    this is that list over three arrays (query index, train index, distance), answering like a list (len, indexing, slices, iteration, == with
    lists, +) and creating a DMatch object when one is asked for - 750 objects cost 0.1 ms of a 0.16 ms match call, 250 a tenth of a tracked
    frame, and the drop-in classes' own filters read the arrays.  An object handed out stays the object of its index; `list(seq)` is the plain
    list.  What a list has and this has not: it cannot be modified in place."""
    __slots__ = ("q", "t", "d", "_objs", "_some")

    def __init__(self, query, train, distance):
        import numpy as np
        self.q = np.ascontiguousarray(query, dtype=np.int64)
        self.t = np.ascontiguousarray(train, dtype=np.int64)
        self.d = np.ascontiguousarray(distance, dtype=np.float64)
        self._objs = None       # list of all objects once a caller iterated
        self._some = None       # {index: object} handed out one by one before that

    def __len__(self):
        return len(self.q)

    def _all(self):
        if self._objs is None:
            objs = _dmatch_objects(self.q.tolist(), self.t.tolist(), self.d.tolist())
            if self._some:
                for i, m in self._some.items():
                    objs[i] = m
            self._objs, self._some = objs, None
        return self._objs

    def __getitem__(self, i):
        if self._objs is not None:
            r = self._objs[i]
            return list(r) if isinstance(i, slice) else r
        if isinstance(i, slice):
            return list(self._all()[i])
        n = len(self.q)
        j = int(i)
        if j < 0:
            j += n
        if not 0 <= j < n:
            raise IndexError("list index out of range")
        if self._some is None:
            self._some = {}
        m = self._some.get(j)
        if m is None:
            m = self._some[j] = _dmatch_objects([int(self.q[j])], [int(self.t[j])], [float(self.d[j])])[0]
        return m

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        if isinstance(other, DMatchList):
            return self._all() == other._all()
        return isinstance(other, list) and self._all() == other

    __hash__ = None  # (like a list)

    def __add__(self, other):
        return self._all() + list(other)

    def __radd__(self, other):
        return list(other) + self._all()

    def __repr__(self):
        return "DMatchList(%d matches)" % len(self.q)

    @property
    def pristine(self):
        """no object was handed out: the arrays are the whole truth (an object a caller holds may have been written to)"""
        return self._objs is None and not self._some


def match_arrays(matches):
    """(queryIdx, trainIdx, distance) of a match list as arrays: the arrays of an untouched DMatchList, else read off the objects"""
    import numpy as np
    if isinstance(matches, DMatchList) and matches.pristine:
        return matches.q, matches.t, matches.d
    return (np.array([m.queryIdx for m in matches], np.int64), np.array([m.trainIdx for m in matches], np.int64),
            np.array([m.distance for m in matches], np.float64))


def dmatches_from_arrays(query, train, distance):
    """three equal-length arrays -> the list of DMatch(queryIdx, trainIdx, 0, distance) as a DMatchList (objects on demand); the plain list
    with VSLAM_AMD_KEYPOINTS=tuple, like keypoints_from_array"""
    import numpy as np
    if _os.environ.get("VSLAM_AMD_KEYPOINTS", "lazy").lower() == "tuple":
        return _dmatch_objects(np.asarray(query).tolist(), np.asarray(train).tolist(), np.asarray(distance, dtype=np.float64).tolist())
    return DMatchList(query, train, distance)
