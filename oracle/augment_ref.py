"""TEST INFRASTRUCTURE ONLY — numpy / scipy restatement of the training augmentation, one sample at a time
like the reference's DataLoader workers (src/training/mytransforms.py: Flip :129-232, Contrast :65-126, Scaling :304-362,
Rotate :259-301, Blur :38-62, Noise :235-256, ToTensor :365-406), driven by the SAME per-sample parameters that
training/device_augment.py draws.  Used by tests/ and tools/bench_augment.py (CPU timing beside the device pipeline);
never by the product path.

"parity unpinned" for the pipeline as a whole: the reference pipeline depends on imgaug / scikit-image (not installed
here) and on unseeded generators, so no golden vectors exist; each operation below is the published formula of the call
the reference makes (np.flip / np.rot90, np.percentile + rescale_intensity, scipy.ndimage.gaussian_filter, affine warp with
bilinear / nearest sampling and constant border, additive Gaussian noise, min-max normalisation).
"""
import math

import numpy as np
from scipy import ndimage as ndi

FLIPS = [lambda a: a, lambda a: np.flip(a, 1), lambda a: np.flip(a, 0), lambda a: np.rot90(a), lambda a: np.rot90(a, 2),
         lambda a: np.rot90(a, 3), lambda a: np.rot90(np.flip(a, 1)), lambda a: np.rot90(np.flip(a, 0))]


def _warp(a, m, order):
    """m: destination -> source map (x, y) as in mseg_aug_affine; scipy works in (row, col)"""
    mat = np.array([[m[4], m[3]], [m[1], m[0]]], np.float64)
    off = np.array([m[5], m[2]], np.float64)
    return ndi.affine_transform(a.astype(np.float64), mat, off, order=order, mode="grid-constant", cval=0.0)


def scale_matrix(sx, sy, H, W):
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    return (1.0 / sx, 0.0, cx - cx / sx, 0.0, 1.0 / sy, cy - cy / sy)


def rotation_matrix(deg, H, W):
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    a = math.radians(float(deg))
    c, s = math.cos(a), math.sin(a)
    return (c, s, cx - c * cx - s * cy, -s, c, cy + s * cx - c * cy)


def augment_sample(img, labels, p, i, min_value=0.0, max_value=65535.0, noise_rng=None):
    """img: (H, W) uint16; labels: list of ((H, W) array, 'linear' | 'nearest'); p: dict of parameter arrays, i: sample index.
    Returns (image fp64 in [-1, 1], [labels])."""
    H, W = img.shape
    v = FLIPS[int(p["flip"][i])](img).astype(np.float64)
    labs = [FLIPS[int(p["flip"][i])](l).astype(np.float64) for l, _ in labels]
    mode, a, b = (float(x) for x in p["contrast"][i][:3])
    if mode == 1:
        p0, p1 = np.percentile(v, (a, b))
        v = np.round(np.clip((v - p0) / (p1 - p0), 0, 1) * 65535) if p1 > p0 else np.zeros_like(v)
    elif mode == 2:
        u = v / 65535.0
        u = (u - u.mean()) * a + u.mean()
        mn, rg = u.min(), u.max() - u.min()
        u = np.power((u - mn) / float(rg + 1e-7), b) * rg + mn
        v = np.floor(np.clip(u, 0, 1) * 65535)
    warps = []
    if p["scale_apply"][i]:
        warps.append(scale_matrix(float(p["scale_xy"][i][0]), float(p["scale_xy"][i][1]), H, W))
    if p["rot_apply"][i]:
        warps.append(rotation_matrix(float(p["rot_deg"][i]), H, W))
    for m in warps:
        v = _warp(v, m, 1)
        labs = [_warp(l, m, 0 if mode_ == "nearest" else 1) for l, (_, mode_) in zip(labs, labels)]
    if p["blur_sigma"][i] > 0:
        v = ndi.gaussian_filter(v, float(p["blur_sigma"][i]), order=0)
    if p["noise_frac"][i] > 0 and noise_rng is not None:
        v = np.round(np.clip(v + noise_rng.normal(0.0, float(p["noise_frac"][i]) * v.max(), v.shape), 0, 65535))
    v = 2 * (np.clip(v, min_value, max_value) - min_value) / (max_value - min_value) - 1
    return v, labs
