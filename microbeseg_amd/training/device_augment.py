"""Training augmentation of a crop batch on the MI355X.

The reference augments every crop on the CPU inside the DataLoader workers (``src/training/mytransforms.py``,
``augmentors`` :12-35): Flip(p=1) -> Contrast(p=0.45) -> Scaling(p=0.25) -> Rotate(p=0.25) -> Blur(p=0.3) -> Noise(p=0.3)
-> ToTensor.  Here the DataLoader delivers the raw crops (uint16 image + labels), and ``DeviceAugment`` applies the same
pipeline to the whole batch in HBM (csrc/augment.hip).  What is kept from the reference:

  * the DECISION TREE and parameter ranges, drawn per sample with the same generators in the same order
    (``random.random() < p``, ``random.randint``, ``random.uniform``, ``np.random.uniform``): Flip picks one of the eight
    symmetries; Contrast picks CLAHE | percentile stretch (0.2/99.8 or 0.1/99.9) | contrast (0.75..1.25) + gamma
    (0.7..1.3); Scaling draws x / y factors in 0.85..1.15; Rotate an angle in -45..45 degrees; Blur sigma in 1..2;
    Noise sigma = 1..5 % of the image maximum;
  * the per-operation arithmetic (tests compare every kernel with the numpy / scipy formula it replaces);
  * geometry is applied to image AND labels (bilinear for float labels, nearest for uint8 labels), intensity operations to
    the image only; the output equals ToTensor's: image in [-1, 1] (min-max normalisation), labels unchanged in type.

Differences (the reference pipeline is not numerically pinned — unseeded RNG, uint16 round trips between stages):
intermediate images stay fp32 instead of being rounded to uint16 after every stage; the CLAHE branch of Contrast is
Zuiderveld's algorithm with scikit-image's defaults (8 x 8 tiles, 256 bins, clip limit 0.01) without the library's
padding / rounding details; imgaug's affine warps are reproduced with centre ((W-1)/2, (H-1)/2), bilinear / nearest sampling and a constant 0 border.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from .. import _lib


def draw_parameters(n, rng=random, nprng=np.random):
    """Per-sample random decisions, in the order the reference's transforms consume their generators.
    Returns a dict of numpy arrays (length n)."""
    flip = np.zeros(n, np.int32)
    contrast = np.zeros((n, 4), np.float32)          # {mode, a, b, -}
    scale_apply = np.zeros(n, np.int32)
    scale_xy = np.ones((n, 2), np.float32)
    rot_apply = np.zeros(n, np.int32)
    rot_deg = np.zeros(n, np.float32)
    blur_sigma = np.zeros(n, np.float32)
    noise_frac = np.zeros(n, np.float32)
    for i in range(n):
        if rng.random() < 1.0:                        # Flip(p=1.0): mytransforms.py:147-148
            flip[i] = rng.randint(0, 7)
        if rng.random() < 0.45:                       # Contrast(p=0.45): :84-90
            h = rng.randint(0, 2)
            if h == 0:                                # CLAHE (equalize_adapthist, clip_limit 0.01)
                contrast[i] = (3, 0, 0, 0)
            elif h == 1:
                contrast[i] = (1, 0.2, 99.8, 0) if rng.randint(0, 1) == 0 else (1, 0.1, 99.9, 0)
            else:
                f = nprng.uniform(0.75, 1.25)
                g = nprng.uniform(0.7, 1.3)
                contrast[i] = (2, f, g, 0)
        if rng.random() < 0.25:                       # Scaling(p=0.25): :322-326
            scale_apply[i] = 1
            scale_xy[i] = (rng.uniform(0.85, 1.15), rng.uniform(0.85, 1.15))
        if rng.random() < 0.25:                       # Rotate(p=0.25): :277-279
            rot_apply[i] = 1
            rot_deg[i] = rng.uniform(-45, 45)
        if rng.random() < 0.3:                        # Blur(p=0.3): :56-58
            blur_sigma[i] = rng.random() + 1.0
        if rng.random() < 0.3:                        # Noise(p=0.3): :250-252
            noise_frac[i] = rng.randint(1, 5) / 100
    return dict(flip=flip, contrast=contrast, scale_apply=scale_apply, scale_xy=scale_xy, rot_apply=rot_apply,
                rot_deg=rot_deg, blur_sigma=blur_sigma, noise_frac=noise_frac)


def scale_matrices(scale_xy, H, W):
    """destination -> source maps of imgaug Affine(scale={'x': sx, 'y': sy}) about the image centre"""
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    m = np.zeros((len(scale_xy), 6), np.float32)
    for i, (sx, sy) in enumerate(scale_xy):
        m[i] = (1.0 / sx, 0.0, cx - cx / sx, 0.0, 1.0 / sy, cy - cy / sy)
    return m


def rotation_matrices(deg, H, W):
    """destination -> source maps of imgaug Affine(rotate=deg) (positive = clockwise on screen) about the image centre"""
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    m = np.zeros((len(deg), 6), np.float32)
    for i, d in enumerate(deg):
        a = math.radians(float(d))
        c, s = math.cos(a), math.sin(a)
        # forward: p' = R(a)(p - c) + c  =>  source p = R(-a)(p' - c) + c
        m[i] = (c, s, cx - c * cx - s * cy, -s, c, cy + s * cx - c * cy)
    return m


class DeviceAugment:
    """Callable: (img uint16 (N, H, W) on the device, list of label planes) -> (image fp32 (N, 1, H, W) in [-1, 1], labels)."""

    def __init__(self, label_type, min_value, max_value, seed=None):
        self.label_type = label_type
        self.min_value, self.max_value = float(min_value), float(max_value)
        self._py = random.Random(seed) if seed is not None else random
        self._np = np.random.default_rng(seed) if seed is not None else np.random
        self._seed = int(seed) if seed is not None else random.getrandbits(31)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def apply(self, img_u16, labels, params):
        """img_u16: int16-view / uint16-valued integer tensor (N, H, W) on the device; labels: list of (plane tensor
        (N, H, W), 'linear' | 'nearest').  params: dict from draw_parameters.  Returns (img (N, 1, H, W) fp32, [labels])."""
        lib = _lib.load()
        dev = img_u16.device
        N, H, W = img_u16.shape
        st = self._stream()
        f32 = lambda: torch.empty((N, H, W), dtype=torch.float32, device=dev)   # noqa: E731
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)        # noqa: E731
        if (params["flip"] >= 3).any() and H != W:
            raise RuntimeError("rot90-type flips need square crops")
        codes = up(params["flip"])
        a, b = f32(), f32()
        src = img_u16.contiguous()
        if src.dtype in (torch.int16, torch.uint16):
            _lib.check(lib.mseg_aug_u16_to_f32(src.data_ptr(), a.data_ptr(), src.numel(), st), "aug_u16_to_f32")
        else:
            a.copy_(src.to(torch.float32))
        # Flip
        _lib.check(lib.mseg_aug_flip(a.data_ptr(), b.data_ptr(), N, H, W, codes.data_ptr(), st), "aug_flip")
        a, b = b, a
        # Contrast
        if (params["contrast"][:, 0] > 0).any():
            stats = torch.empty((N, 3), dtype=torch.float32, device=dev)
            hist = torch.empty((N, 65536), dtype=torch.int32, device=dev)
            par = torch.zeros((N, 8), dtype=torch.float32, device=dev)
            choice = up(params["contrast"])
            _lib.check(lib.mseg_aug_stats(a.data_ptr(), N, H, W, stats.data_ptr(), hist.data_ptr(), st), "aug_stats")
            _lib.check(lib.mseg_aug_contrast_params(stats.data_ptr(), hist.data_ptr(), choice.data_ptr(), N, H * W,
                                                    par.data_ptr(), st), "aug_contrast_params")
            _lib.check(lib.mseg_aug_contrast(a.data_ptr(), b.data_ptr(), N, H, W, par.data_ptr(), st), "aug_contrast")
            a, b = b, a
            if (params["contrast"][:, 0] == 3).any():
                cws = torch.empty(lib.mseg_aug_clahe_workspace_bytes(N), dtype=torch.uint8, device=dev)
                _lib.check(lib.mseg_aug_clahe(a.data_ptr(), b.data_ptr(), N, H, W, choice.data_ptr(), cws.data_ptr(), st),
                           "aug_clahe")
                a, b = b, a
        # Scaling, Rotate (image + labels)
        warps = []
        if params["scale_apply"].any():
            warps.append((up(scale_matrices(params["scale_xy"], H, W)), up(params["scale_apply"])))
        if params["rot_apply"].any():
            warps.append((up(rotation_matrices(params["rot_deg"], H, W)), up(params["rot_apply"])))
        for mats, apply in warps:
            _lib.check(lib.mseg_aug_affine(a.data_ptr(), b.data_ptr(), N, H, W, mats.data_ptr(), apply.data_ptr(), 0, st),
                       "aug_affine")
            a, b = b, a
        out_labels = []
        for plane, mode in labels:
            la = plane.to(torch.float32).contiguous()
            lb = torch.empty_like(la)
            _lib.check(lib.mseg_aug_flip(la.data_ptr(), lb.data_ptr(), N, H, W, codes.data_ptr(), st), "aug_flip")
            la, lb = lb, la
            for mats, apply in warps:
                _lib.check(lib.mseg_aug_affine(la.data_ptr(), lb.data_ptr(), N, H, W, mats.data_ptr(), apply.data_ptr(),
                                               1 if mode == "nearest" else 0, st), "aug_affine")
                la, lb = lb, la
            out_labels.append(la)
        # Blur
        if (params["blur_sigma"] > 0).any():
            sig = up(params["blur_sigma"])
            tmp = f32()
            _lib.check(lib.mseg_aug_blur(a.data_ptr(), tmp.data_ptr(), b.data_ptr(), N, H, W, sig.data_ptr(), st),
                       "aug_blur")
            a, b = b, a
        # Noise + ToTensor normalisation
        stats = torch.empty((N, 3), dtype=torch.float32, device=dev)
        _lib.check(lib.mseg_aug_stats(a.data_ptr(), N, H, W, stats.data_ptr(), None, st), "aug_stats")
        frac = up(params["noise_frac"])
        self._seed = (self._seed * 1103515245 + 12345) & 0x7fffffff
        _lib.check(lib.mseg_aug_noise_normalize(a.data_ptr(), b.data_ptr(), N, H, W, frac.data_ptr(), stats.data_ptr(),
                                                C.c_uint32(self._seed), self.min_value, self.max_value, st),
                   "aug_noise_normalize")
        return b.view(N, 1, H, W), out_labels

    def __call__(self, img_u16, labels):
        params = draw_parameters(img_u16.shape[0], self._py, self._np)
        return self.apply(img_u16, labels, params)
