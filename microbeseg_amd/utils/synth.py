"""Synthetic crops / frames with the on-disk layout of a microbeSEG training set:
random ellipses -> instance mask, uint16 image, cell / neighbour distance labels (fp32 in [0,1]) and 3-class boundary
labels.  Used by tests, bench.py and the examples — there is no network access to real data.  numpy PCG64 seeded."""
import numpy as np

from . import tiffio


def _edt(mask):
    from scipy import ndimage as ndi
    return ndi.distance_transform_edt(mask)


def synth_instance_mask(rng, size, n_cells, rmin=4.0, rmax=14.0):
    H = W = size
    mask = np.zeros((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    k, tries = 0, 0
    while k < n_cells and tries < 40 * n_cells:
        tries += 1
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b, th = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax), rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = ((u / a) ** 2 + (v / b) ** 2 <= 1) & (mask == 0)
        if blob.sum() < 12:
            continue
        k += 1
        mask[blob] = k
    return mask


def synth_crop(rng, size=256, n_cells=None):
    """-> dict(img uint16, mask uint16, cell_dist f32, neighbor_dist f32, boundary uint8)"""
    from scipy import ndimage as ndi
    n_cells = n_cells if n_cells is not None else max(2, int(40 * (size / 256.0) ** 2))
    mask = synth_instance_mask(rng, size, n_cells)
    fg = mask > 0
    img = ndi.gaussian_filter(fg.astype(np.float32), 1.5) * rng.uniform(0.4, 0.9) + rng.normal(0, 0.02, fg.shape)
    img = np.clip((img - img.min()) / max(img.max() - img.min(), 1e-6), 0, 1)
    img = (img * 65535).astype(np.uint16)
    cell = np.zeros(fg.shape, np.float32)
    border = np.zeros(fg.shape, np.float32)
    inner = np.zeros(fg.shape, bool)
    for i in range(1, int(mask.max()) + 1):
        m = mask == i
        d = _edt(m)
        cell[m] = (d[m] / d.max()).astype(np.float32)
        inner |= ndi.binary_erosion(m)
        other = fg & ~m
        if other.any():
            border[m] = np.clip(1.0 - _edt(~other)[m] / 6.0, 0, 1)
    boundary = np.zeros(fg.shape, np.uint8)
    boundary[inner] = 1
    boundary[fg & ~inner] = 2
    return dict(img=img, mask=mask, cell_dist=cell, neighbor_dist=border, boundary=boundary)


def write_training_set(root, n_train, n_val, size=256, seed=1234, label_types=("distance", "boundary")):
    """Write ``root/{train,val}/{img,mask,cell_dist,neighbor_dist,boundary}_<id>.tif``."""
    root.mkdir(parents=True, exist_ok=True)
    for mode, n, s in (("train", n_train, seed), ("val", n_val, seed + 3087)):
        d = root / mode
        d.mkdir(exist_ok=True)
        rng = np.random.Generator(np.random.PCG64(s))
        for i in range(n):
            c = synth_crop(rng, size)
            tiffio.imwrite(d / f"img_{i:03d}.tif", c["img"])
            tiffio.imwrite(d / f"mask_{i:03d}.tif", c["mask"])
            if "distance" in label_types:
                tiffio.imwrite(d / f"cell_dist_{i:03d}.tif", c["cell_dist"])
                tiffio.imwrite(d / f"neighbor_dist_{i:03d}.tif", c["neighbor_dist"])
            if "boundary" in label_types:
                tiffio.imwrite(d / f"boundary_{i:03d}.tif", c["boundary"])
    return root


def synth_prediction_maps(rng, H, W, n_cells, rmin=8.0, rmax=20.0, noise=0.01):
    """Network-output-like (cell, border) fp32 maps for post-processing benchmarks/tests: cell = per-cell radial
    distance profile in [0,1] (max over cells), border = high where two cells overlap or touch; Gaussian noise makes
    values tie-free like real predictions.  Bounding-box based, so 2048x2048 / 2500 cells takes ~1 s."""
    cell = np.zeros((H, W), np.float32)
    border = np.zeros((H, W), np.float32)
    for _ in range(n_cells):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b, th = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax), rng.uniform(0, np.pi)
        r = int(np.ceil(max(a, b))) + 1
        y0, y1, x0, x1 = max(0, int(cy) - r), min(H, int(cy) + r + 1), max(0, int(cx) - r), min(W, int(cx) + r + 1)
        if y1 <= y0 or x1 <= x0:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = np.clip(1.0 - np.sqrt((u / a) ** 2 + (v / b) ** 2), 0, 1).astype(np.float32)
        sub = cell[y0:y1, x0:x1]
        border[y0:y1, x0:x1] = np.maximum(border[y0:y1, x0:x1], np.minimum(sub, blob) * 3.0)
        cell[y0:y1, x0:x1] = np.maximum(sub, blob)
    cell += rng.normal(0, noise, cell.shape).astype(np.float32)
    border = np.clip(border + rng.normal(0, noise, cell.shape).astype(np.float32), 0, 1).astype(np.float32)
    return cell, border
