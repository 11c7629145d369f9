#!/opt/conda/bin/python3.9
"""Generate tests/golden/postproc_*.npz with the REAL reference post-processing (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tools/gen_golden_postproc.py
Imports /root/reference/src/inference/postprocessing.py (scipy 1.7.1 / scikit-image 0.18.3 of the conda env) and
stores only inputs + the outputs the reference produced for them (plus a few third-party known answers:
scipy gaussian_filter, skimage label / watershed on raw arrays).
"""
import pathlib
import sys

import numpy as np
from scipy import ndimage as ndi

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from src.inference.postprocessing import distance_postprocessing, boundary_postprocessing  # noqa: E402
from skimage import measure  # noqa: E402
from skimage.segmentation import watershed  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def synth_cells(rng, H, W, n_cells, rmin=4, rmax=12, touching=False):
    """Instance mask of random ellipses (touching=True lets them abut) -> (cell map, border map) like the net outputs."""
    mask = np.zeros((H, W), np.int32)
    yy, xx = np.mgrid[0:H, 0:W]
    k = 0
    tries = 0
    while k < n_cells and tries < 50 * n_cells:
        tries += 1
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax)
        th = rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = (u / a) ** 2 + (v / b) ** 2 <= 1
        if touching:
            blob &= mask == 0
            if blob.sum() < 12:
                continue
        elif (ndi.binary_dilation(blob, iterations=2) & (mask > 0)).any():
            continue
        k += 1
        mask[blob] = k
    cell = np.zeros((H, W), np.float32)
    for i in range(1, k + 1):
        m = mask == i
        d = ndi.distance_transform_edt(m)
        cell[m] = (d[m] / d.max()).astype(np.float32)
    fg = mask > 0
    # neighbour-distance style border map: high where a pixel of one cell is close to another cell
    border = np.zeros((H, W), np.float32)
    for i in range(1, k + 1):
        other = (mask > 0) & (mask != i)
        if other.any():
            d = ndi.distance_transform_edt(~other)
            m = mask == i
            border[m] = np.clip(1.0 - d[m] / 6.0, 0, 1)
    cell = cell + rng.normal(0, 0.01, cell.shape).astype(np.float32)
    border = border + rng.normal(0, 0.01, border.shape).astype(np.float32)
    return cell.astype(np.float32), border.astype(np.float32), mask


def save(name, **arrs):
    np.savez_compressed(OUT / f"postproc_{name}.npz", **arrs)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in arrs.items() if hasattr(v, "shape") and v.ndim > 0})


def distance_case(name, cell, border, pairs):
    out = {"cell": cell, "border": border, "th": np.array(pairs, np.float64)}
    for j, (th_cell, th_seed) in enumerate(pairs):
        # callers pass (H,W,1) float32 arrays and keyword thresholds (infer.py:358-365)
        lab = distance_postprocessing(border_prediction=border[..., None].copy(), cell_prediction=cell[..., None].copy(),
                                      th_seed=th_seed, th_cell=th_cell)
        out[f"labels_hw1_{j}"] = lab
        lab2 = distance_postprocessing(border_prediction=border.copy(), cell_prediction=cell.copy(), th_seed=th_seed,
                                       th_cell=th_cell)
        out[f"labels_2d_{j}"] = lab2
        print("   ", name, (th_cell, th_seed), "instances", int(lab.max()), "hw1==2d", bool((lab == lab2).all()))
    save(name, **out)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    grid = [(0.10, 0.45), (0.05, 0.35), (0.125, 0.45)]   # defaults + eval grid corners (eval.py:128)

    rng = np.random.Generator(np.random.PCG64(31))
    cell, border, _ = synth_cells(rng, 96, 112, 14)
    distance_case("distance_generic", cell, border, grid)

    rng = np.random.Generator(np.random.PCG64(32))
    cell, border, _ = synth_cells(rng, 160, 128, 40, touching=True)
    distance_case("distance_touching", cell, border, grid)

    rng = np.random.Generator(np.random.PCG64(33))
    cell, border, _ = synth_cells(rng, 64, 80, 6)
    distance_case("distance_no_seeds", (cell * 0.4).astype(np.float32), border, [(0.10, 0.45)])

    rng = np.random.Generator(np.random.PCG64(34))
    cell, border, _ = synth_cells(rng, 80, 64, 30, rmin=1.2, rmax=2.2)
    distance_case("distance_small_seeds", cell, border, [(0.10, 0.45), (0.05, 0.35)])

    rng = np.random.Generator(np.random.PCG64(35))
    cell, border, _ = synth_cells(rng, 128, 96, 30, touching=True)
    q = (np.round(cell * 16) / 16).astype(np.float32)     # heavy exact ties in the flood
    qb = (np.round(border * 8) / 8).astype(np.float32)
    distance_case("distance_quantised_ties", q, qb, [(0.10, 0.45), (0.05, 0.35)])

    rng = np.random.Generator(np.random.PCG64(36))
    cell, border, _ = synth_cells(rng, 256, 256, 45, rmin=6, rmax=16, touching=True)
    distance_case("distance_256", cell, border, [(0.10, 0.45)])

    # seeds outside the mask (th_seed < th_cell is allowed on the CLI: markers are multiplied by the mask)
    rng = np.random.Generator(np.random.PCG64(37))
    cell, border, _ = synth_cells(rng, 72, 72, 10)
    distance_case("distance_seed_below_cell", cell, (border * 0).astype(np.float32), [(0.60, 0.30)])

    # ---- boundary method: softmax probabilities (H,W,3) -----------------------------------------------------
    for name, seed, touching in (("boundary_generic", 41, False), ("boundary_touching", 42, True)):
        rng = np.random.Generator(np.random.PCG64(seed))
        cell, border, mask = synth_cells(rng, 112, 96, 25, touching=touching)
        fg = mask > 0
        er = ndi.binary_erosion(fg, iterations=1)
        inner = np.zeros_like(fg)
        for i in range(1, mask.max() + 1):
            inner |= ndi.binary_erosion(mask == i, iterations=1)
        logits = np.stack([np.where(~fg, 3.0, -1.0), np.where(inner, 3.0, -1.0), np.where(fg & ~inner, 2.0, -2.0)], -1)
        logits = logits + rng.normal(0, 0.8, logits.shape)
        e = np.exp(logits - logits.max(-1, keepdims=True))
        probs = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        lab = boundary_postprocessing(probs.copy())
        print("   ", name, "instances", int(lab.max()))
        save(name, probs=probs, labels=lab)

    # ---- third-party known answers ------------------------------------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(51))
    g_in = rng.normal(0, 1, (97, 113)).astype(np.float32)
    save("gauss_sigma05", x=g_in, y=ndi.gaussian_filter(g_in[..., None], sigma=0.5)[..., 0],
         y2d=ndi.gaussian_filter(g_in, sigma=0.5), tiny=ndi.gaussian_filter(g_in[:3, :2].copy(), sigma=0.5))

    rng = np.random.Generator(np.random.PCG64(52))
    b = rng.random((70, 90)) > 0.62
    lab_bool = measure.label(b, background=0)
    li = lab_bool.copy()
    li[np.isin(li, [2, 5, 9, 14])] = 0
    save("label_order", bin=b, label_bool=lab_bool.astype(np.int32), int_in=li.astype(np.int32),
         relabel_2d=measure.label(li, background=0).astype(np.int32),
         relabel_hw1=measure.label(li[..., None], background=0)[..., 0].astype(np.int32))

    for name, seed, quant in (("watershed_float", 61, 0), ("watershed_ties", 62, 8), ("watershed_const", 63, 1)):
        rng = np.random.Generator(np.random.PCG64(seed))
        H, W = 96, 104
        img = ndi.gaussian_filter(rng.normal(0, 1, (H, W)), 2.0)
        if quant == 1:
            img = np.ones((H, W))
        elif quant:
            img = np.round((img - img.min()) / (img.max() - img.min()) * quant) / quant
        mask = ndi.gaussian_filter(rng.normal(0, 1, (H, W)), 3.0) > -0.05
        markers = np.zeros((H, W), np.int32)
        pts = rng.integers(0, H * W, 40)
        for k, pidx in enumerate(pts):
            y, x = divmod(int(pidx), W)
            markers[max(0, y - 1):y + 2, max(0, x - 1):x + 2] = k + 1
        out = watershed(image=img, markers=markers, mask=mask, watershed_line=False)
        out3 = watershed(image=img[..., None], markers=markers[..., None], mask=mask[..., None], watershed_line=False)
        assert (out == out3[..., 0]).all()
        save(name, image=img.astype(np.float64), markers=markers, mask=mask, out=out.astype(np.int32))


if __name__ == "__main__":
    main()
