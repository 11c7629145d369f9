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


def clahe(v, tiles=8, bins=256, gray=16384, clip=0.01):
    """Zuiderveld's CLAHE with scikit-image's defaults, the variant csrc/augment.hip implements (module docstring there)."""
    H, W = v.shape
    g = np.clip(np.floor(v * ((gray - 1) / 65535.0) + 0.5), 0, gray - 1).astype(np.int64)
    b = g // (gray // bins)
    maps = np.zeros((tiles, tiles, bins), np.float32)
    for ty in range(tiles):
        for tx in range(tiles):
            y0, y1 = ty * H // tiles, (ty + 1) * H // tiles
            x0, x1 = tx * W // tiles, (tx + 1) * W // tiles
            hist = np.bincount(b[y0:y1, x0:x1].ravel(), minlength=bins).astype(np.int64)
            npx = (y1 - y0) * (x1 - x0)
            clim = max(1, int(np.float32(clip) * np.float32(npx)))
            excess = int(np.maximum(hist - clim, 0).sum())
            incr, upper = excess // bins, clim - excess // bins
            for k in range(bins):
                h = hist[k]
                if h > clim:
                    hist[k] = clim
                elif h > upper:
                    excess -= clim - h
                    hist[k] = clim
                else:
                    excess -= incr
                    hist[k] = h + incr
            guard = 0
            while excess > 0 and guard < 64:
                for k in range(bins):
                    if excess <= 0:
                        break
                    if hist[k] < clim:
                        hist[k] += 1
                        excess -= 1
                guard += 1
            scale = np.float32(gray - 1) / np.float32(max(npx, 1))
            maps[ty, tx] = np.minimum(np.cumsum(hist.astype(np.float32)) * scale, np.float32(gray - 1))
    yy, xx = np.mgrid[0:H, 0:W]
    fy = (yy.astype(np.float32) + 0.5) * tiles / np.float32(H) - 0.5
    fx = (xx.astype(np.float32) + 0.5) * tiles / np.float32(W) - 0.5
    ty0, tx0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    ay, ax = fy - ty0, fx - tx0
    ty1, tx1 = np.minimum(ty0 + 1, tiles - 1), np.minimum(tx0 + 1, tiles - 1)
    ty0, tx0 = np.maximum(ty0, 0), np.maximum(tx0, 0)
    m = (1 - ay) * ((1 - ax) * maps[ty0, tx0, b] + ax * maps[ty0, tx1, b]) + \
        ay * ((1 - ax) * maps[ty1, tx0, b] + ax * maps[ty1, tx1, b])
    return np.floor(np.clip(m / np.float32(gray - 1), 0, 1) * 65535.0)


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
    elif mode == 3:
        v = clahe(v)
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
