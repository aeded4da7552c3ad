"""Shared helpers for the parity tests: fixture loading and seeded synthetic inputs."""
import os

import numpy as np
import yaml

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_png_bgr(path):
    """Decode a PNG to a BGR uint8 array (what cv2.imread returns for these files)."""
    from PIL import Image
    rgb = np.array(Image.open(path).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def gt_pair(i):
    d = os.path.join(GOLDEN, "gt_pairs", "pair%02d" % i)
    with open(os.path.join(d, "gt.yaml")) as f:
        gt = yaml.safe_load(f)
    return d, gt


def greedy_min_dist(points, min_dist=10):
    """gt_match_annotator.py:64-72 filter_close_keypoints on (x, y) float pairs, order-sensitive."""
    kept = []
    for x, y in points:
        if all((x - px) ** 2 + (y - py) ** 2 >= min_dist ** 2 for px, py in kept):
            kept.append((x, y))
    return kept


def synthetic_frame(seed, w=640, h=480):
    """Seeded textured frame: bilinear-upsampled 8-px random cells + 400 random grey rectangles +
    N(0,3) noise.  Plenty of FAST-7 corners on every pyramid level."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cw, ch = w // 8 + 2, h // 8 + 2
    cells = rng.uniform(0, 255, size=(ch, cw))
    ys = (np.arange(h) + 0.5) / 8.0
    xs = (np.arange(w) + 0.5) / 8.0
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    img = (cells[y0][:, x0] * (1 - fy) * (1 - fx) + cells[y0][:, x0 + 1] * (1 - fy) * fx +
           cells[y0 + 1][:, x0] * fy * (1 - fx) + cells[y0 + 1][:, x0 + 1] * fy * fx)
    for _ in range(400):
        rw, rh = rng.integers(6, 41, size=2)
        x = rng.integers(0, w - 1); y = rng.integers(0, h - 1)
        img[y:y + rh, x:x + rw] = rng.uniform(0, 255)
    img = img + rng.normal(0, 3, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def parallax_frames(nb, seed=31, w=640, h=480, bg_step=2, fg_step=4):
    """nb frames of a camera translating along x past a two-depth scene: the background pans bg_step px / frame, the
    foreground patches (where a coarse mask image is bright) fg_step px / frame.  Non-planar, so the essential matrix of
    consecutive frames is well posed.  uint8 [nb, h, w]."""
    wide = w + fg_step * nb + 8
    bg = synthetic_frame(seed, wide, h)
    fg = synthetic_frame(seed + 1, wide, h)
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    cells = rng.uniform(0, 255, size=(h // 60 + 2, wide // 60 + 2))
    mk = np.kron(cells, np.ones((60, 60)))[:h, :wide] > 150
    out = np.empty((nb, h, w), np.uint8)
    for i in range(nb):
        xb, xf = bg_step * i, fg_step * i
        out[i] = np.where(mk[:, xf:xf + w], fg[:, xf:xf + w], bg[:, xb:xb + w])
    return out
