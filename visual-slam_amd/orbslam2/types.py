"""cv2.KeyPoint / cv2.DMatch stand-ins.

When cv2 is importable the real classes are used (the caller's drawKeypoints / drawMatches need them); otherwise
duck-typed objects with the same attributes and constructor argument order."""
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


def keypoints_from_array(arr):
    """structured mo_keypoint array -> tuple of KeyPoint objects (cv2 returns a tuple)"""
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


def keypoints_to_array(kps):
    import numpy as np
    from vslam_amd import KP_DTYPE
    return np.array([(k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id) for k in kps], KP_DTYPE).reshape(-1)
