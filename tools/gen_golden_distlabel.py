#!/opt/conda/bin/python3.9
"""Generate tests/golden/labels_distance.npz with the REAL reference distance_label (and its bottom_hat_closing helper)
(src/training/train_data_representations.py:40-72, 261-361), build container only.

Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tools/gen_golden_distlabel.py
Compatibility of the conda env with the reference's pins (numpy 1.21): `np.float` (an alias of the builtin float that numpy
1.24 removed) is restored before the import; cv2 (imported at module level, used only by adapted_border_label) is an empty
placeholder.  Only inputs and the arrays the reference produced are stored."""
import pathlib
import sys
import types

import numpy as np

np.float = float
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from src.training.train_data_representations import (distance_label, bottom_hat_closing, cell_distance_label,  # noqa: E402
                                                     j4_label)

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def blobs(rng, H, W, n, rmin, rmax, gap=0):
    """random ellipses; gap > 0 erodes every cell by `gap` pixels afterwards -> narrow background gaps between neighbours"""
    from scipy import ndimage as ndi
    mask = np.zeros((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, n + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b, th = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax), rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = ((u / a) ** 2 + (v / b) ** 2 <= 1) & (mask == 0)
        mask[blob] = k
    if gap:
        out = np.zeros_like(mask)
        for k in np.unique(mask)[1:]:
            out[ndi.binary_erosion(mask == k, iterations=gap)] = k
        mask = out
    return mask


out = {}
cases = [(96, 96, 14, 5, 11, 0, 24), (128, 160, 40, 5, 12, 1, 30), (128, 128, 60, 4, 9, 1, 20), (80, 120, 10, 8, 16, 2, 40),
         (64, 64, 3, 6, 10, 0, 12)]
for i, (H, W, n, r0, r1, gap, sr) in enumerate(cases):
    rng = np.random.Generator(np.random.PCG64(1300 + i))
    m = blobs(rng, H, W, n, r0, r1, gap)
    cell, nb = distance_label(m.copy(), sr)
    closed, corr = bottom_hat_closing(m.copy())
    out[f"m{i}"] = m
    out[f"sr{i}"] = np.int32(sr)
    out[f"cell{i}"] = cell
    out[f"neighbor{i}"] = nb
    out[f"closed{i}"] = closed.astype(np.int32)
    out[f"corr{i}"] = corr.astype(np.float32)
    out[f"celld{i}"] = cell_distance_label(m.copy(), sr)
    out[f"cellc{i}"] = cell_distance_label(m.copy(), sr, apply_clipping=True)
    out[f"j4{i}"] = j4_label(m.astype(np.int32))
    # the search radius of the label creation comes from the largest major axis (src/training/train.py:73-78)
    from skimage.measure import regionprops
    out[f"mal{i}"] = np.array([c.major_axis_length for c in regionprops(m.astype(np.int32))], np.float64)
    print(i, m.shape, "cells", len(np.unique(m)) - 1, "gaps", int(closed.max()), "cell max", float(cell.max()), "nb max",
          float(nb.max()), "nb>0 px", int((nb > 0).sum()))
np.savez_compressed(OUT / "labels_distance.npz", **out)
print("wrote", OUT / "labels_distance.npz")
