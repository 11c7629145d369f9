"""GPU (MI355X): training augmentation kernels through the C ABI, each against the numpy / scipy
formula of the reference transform it replaces (src/training/mytransforms.py), plus the whole DeviceAugment pipeline."""
import numpy as np
import pytest
import torch
from scipy import ndimage as ndi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aug():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd.training import device_augment
    return device_augment


def _params(n):
    return dict(flip=np.zeros(n, np.int32), contrast=np.zeros((n, 4), np.float32), scale_apply=np.zeros(n, np.int32),
                scale_xy=np.ones((n, 2), np.float32), rot_apply=np.zeros(n, np.int32), rot_deg=np.zeros(n, np.float32),
                blur_sigma=np.zeros(n, np.float32), noise_frac=np.zeros(n, np.float32))


def _img(rng, n, h, w):
    return rng.integers(0, 65536, (n, h, w)).astype(np.uint16)


def _run(aug, img, labels, p, lo=0, hi=65535):
    da = aug.DeviceAugment("distance", lo, hi, seed=7)
    t = torch.from_numpy(img.view(np.int16)).cuda()
    labs = [(torch.from_numpy(l).cuda(), m) for l, m in labels]
    out, ol = da.apply(t, labs, p)
    return out.cpu().numpy()[:, 0], [o.cpu().numpy() for o in ol]


def _norm(v, lo=0.0, hi=65535.0):
    return 2 * (np.clip(v, lo, hi) - lo) / (hi - lo) - 1


def test_flip_codes_match_numpy(aug):
    rng = np.random.default_rng(0)
    img = _img(rng, 8, 48, 48)
    lab = rng.random((8, 48, 48)).astype(np.float32)
    p = _params(8)
    p["flip"][:] = np.arange(8)
    out, (ol,) = _run(aug, img, [(lab, "linear")], p)
    ops = [lambda a: a, lambda a: np.flip(a, 1), lambda a: np.flip(a, 0), lambda a: np.rot90(a), lambda a: np.rot90(a, 2),
           lambda a: np.rot90(a, 3), lambda a: np.rot90(np.flip(a, 1)), lambda a: np.rot90(np.flip(a, 0))]
    for k, op in enumerate(ops):
        assert np.array_equal(out[k], _norm(op(img[k]).astype(np.float32)).astype(np.float32))
        assert np.array_equal(ol[k], op(lab[k]))


def test_affine_matches_scipy(aug):
    rng = np.random.default_rng(1)
    n, h, w = 4, 40, 56
    img = ndi.gaussian_filter(rng.random((n, h, w)), 2.0)
    img = (img / img.max() * 60000).astype(np.uint16)
    lab8 = np.kron(rng.integers(0, 3, (n, h // 8, w // 8)), np.ones((8, 8))).astype(np.float32)   # blocky class labels
    p = _params(n)
    p["scale_apply"][:] = (1, 1, 0, 0)
    p["scale_xy"][:] = ((1.1, 0.9), (0.87, 1.13), (1, 1), (1, 1))
    p["rot_apply"][:] = (0, 1, 1, 0)
    p["rot_deg"][:] = (0, 30.0, -17.5, 0)
    out, (ol,) = _run(aug, img, [(lab8, "nearest")], p)
    cx, cy = (w - 1) / 2, (h - 1) / 2
    for k in range(n):
        ref = img[k].astype(np.float64)
        refl = lab8[k].astype(np.float64)
        if p["scale_apply"][k]:
            sx, sy = p["scale_xy"][k]
            mat = np.array([[1 / sy, 0], [0, 1 / sx]])                       # (row, col) convention of scipy
            off = np.array([cy - cy / sy, cx - cx / sx])
            ref = ndi.affine_transform(ref, mat, off, order=1, mode="grid-constant", cval=0.0)
            refl = ndi.affine_transform(refl, mat, off, order=0, mode="grid-constant", cval=0.0)
        if p["rot_apply"][k]:
            a = np.deg2rad(p["rot_deg"][k])
            c, s = np.cos(a), np.sin(a)
            mat = np.array([[c, -s], [s, c]])                                # row' = -s x + c y ..., see rotation_matrices
            off = np.array([cy - (c * cy - s * cx), cx - (s * cy + c * cx)])
            ref = ndi.affine_transform(ref, mat, off, order=1, mode="grid-constant", cval=0.0)
            refl = ndi.affine_transform(refl, mat, off, order=0, mode="grid-constant", cval=0.0)
        assert np.abs(out[k] - _norm(ref)).max() < 2e-4
        inner = np.s_[3:-3, 3:-3]
        assert (ol[k][inner] != refl[inner]).mean() < 0.02                   # nearest: only block edges may differ (fp32 vs fp64 coords)


def test_blur_matches_scipy_gaussian_filter(aug):
    rng = np.random.default_rng(2)
    img = _img(rng, 3, 37, 53)
    p = _params(3)
    p["blur_sigma"][:] = (1.0, 1.6, 0.0)
    out, _ = _run(aug, img, [], p)
    for k, s in enumerate(p["blur_sigma"]):
        ref = ndi.gaussian_filter(img[k].astype(np.float64), s, order=0) if s > 0 else img[k].astype(np.float64)
        assert np.abs(out[k] - _norm(ref)).max() < 1e-4


def test_contrast_branches_match_numpy(aug):
    rng = np.random.default_rng(3)
    img = (ndi.gaussian_filter(rng.random((3, 64, 64)), 3.0) * 9e4).clip(0, 65535).astype(np.uint16)
    p = _params(3)
    p["contrast"][:] = ((1, 0.2, 99.8, 0), (1, 0.1, 99.9, 0), (2, 1.2, 0.8, 0))
    out, _ = _run(aug, img, [], p)
    for k in range(2):
        p0, p1 = np.percentile(img[k], tuple(p["contrast"][k, 1:3]))
        ref = np.round(np.clip((img[k].astype(np.float64) - p0) / (p1 - p0), 0, 1) * 65535)      # rescale_intensity
        assert np.abs(out[k] - _norm(ref)).max() < 1e-4
    v = img[2].astype(np.float32) / 65535                                                         # mytransforms.py:103-122
    f, g = 1.2, 0.8
    v = (v - v.mean()) * f + v.mean()
    mn, rng_ = v.min(), v.max() - v.min()
    v = np.power((v - mn) / float(rng_ + 1e-7), g) * rng_ + mn
    ref = np.floor(np.clip(v, 0, 1) * 65535)
    assert np.abs(out[2] - _norm(ref)).max() < 2e-4


def test_noise_statistics_and_normalisation(aug):
    rng = np.random.default_rng(4)
    img = np.full((2, 128, 128), 30000, np.uint16)
    img[:, 0, 0] = 40000                                                     # the maximum that sigma refers to
    p = _params(2)
    p["noise_frac"][:] = (0.05, 0.0)
    out, _ = _run(aug, img, [], p)
    assert np.array_equal(out[1], _norm(img[1].astype(np.float32)).astype(np.float32))
    d = (out[0] + 1) / 2 * 65535 - 30000
    d = d.ravel()[1:]
    assert abs(d.mean()) < 40 and abs(d.std() - 0.05 * 40000) < 60
    out2, _ = _run(aug, img, [], p, lo=10000, hi=35000)                      # ToTensor clips to [min, max]
    assert out2.max() <= 1.0 and out2.min() >= -1.0 and out2[1, 0, 0] == 1.0


def test_pipeline_shapes_ranges_and_label_consistency(aug):
    rng = np.random.default_rng(5)
    n, s = 16, 64
    img = _img(rng, n, s, s)
    cell = rng.random((n, s, s)).astype(np.float32)
    da = aug.DeviceAugment("distance", 0, 65535, seed=11)
    out, (c2, b2) = da(torch.from_numpy(img.view(np.int16)).cuda(),
                       [(torch.from_numpy(cell).cuda(), "linear"), (torch.from_numpy(cell).cuda(), "linear")])
    assert out.shape == (n, 1, s, s) and out.dtype == torch.float32
    assert float(out.min()) >= -1.0 and float(out.max()) <= 1.0
    assert torch.equal(c2, b2)                                               # same geometry for both label planes
    assert float(c2.min()) >= 0.0 and float(c2.max()) <= 1.0 + 1e-6


def test_pipeline_matches_cpu_restatement_with_identical_parameters(aug):
    """Whole pipeline (no noise: its generator differs) vs oracle/augment_ref.py driven by the same drawn parameters."""
    import pathlib
    import random
    import sys
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
    from oracle import augment_ref
    rng = np.random.default_rng(6)
    n, s = 24, 64
    img = (ndi.gaussian_filter(rng.random((n, s, s)), 2.0) * 1.2e5).clip(0, 65535).astype(np.uint16)
    cell = ndi.gaussian_filter(rng.random((n, s, s)), 1.5).astype(np.float32)
    cls = np.kron(rng.integers(0, 3, (n, s // 8, s // 8)), np.ones((8, 8))).astype(np.float32)
    p = aug.draw_parameters(n, random.Random(5), np.random.default_rng(5))
    p["noise_frac"][:] = 0
    assert p["scale_apply"].any() and p["rot_apply"].any() and (p["blur_sigma"] > 0).any()
    assert {1, 2, 3} <= set(p["contrast"][:, 0].astype(int))
    out, (oc, ol) = _run(aug, img, [(cell, "linear"), (cls, "nearest")], p)
    for i in range(n):
        want, (wc, wl) = augment_ref.augment_sample(img[i], [(cell[i], "linear"), (cls[i], "nearest")], p, i)
        tol = 3e-3 if p["contrast"][i, 0] == 3 else 5e-4      # CLAHE: a pixel on a bin edge may fall in the neighbour bin
        assert np.abs(out[i] - want).max() < tol, i
        assert np.abs(oc[i] - wc).max() < 1e-4, i
        assert (ol[i][4:-4, 4:-4] != wl[4:-4, 4:-4]).mean() < 0.03, i
