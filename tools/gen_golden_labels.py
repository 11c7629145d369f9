#!/opt/conda/bin/python3.9
"""Generate tests/golden/labels_boundary.npz with the REAL reference label creation for the boundary method
(src/training/train_data_representations.py: boundary_label :75-99, border_label :102-125), build container only.

Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tools/gen_golden_labels.py
The reference module imports cv2 at module level (used only by adapted_border_label); cv2 is not installed here, an empty
placeholder module satisfies the import.  Only inputs and the arrays the reference produced are stored."""
import pathlib
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from src.training.train_data_representations import boundary_label, border_label  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def blobs(rng, H, W, n, rmin, rmax, touching):
    mask = np.zeros((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, n + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b, th = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax), rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = (u / a) ** 2 + (v / b) ** 2 <= 1
        if touching:
            blob &= mask == 0
        mask[blob] = k * 3 + 1            # non-contiguous ids
    return mask


out = {}
for i, (H, W, n, touching) in enumerate([(48, 64, 8, True), (96, 80, 30, True), (128, 128, 40, False), (33, 57, 5, True),
                                         (64, 64, 0, True)]):
    rng = np.random.Generator(np.random.PCG64(800 + i))
    m = blobs(rng, H, W, n, 4, 11, touching)
    if i == 3:
        m[0, :] = 7; m[:, 0] = 9                      # instances on the image border
    out[f"m{i}"] = m
    out[f"boundary{i}"] = boundary_label(m)
    out[f"border{i}"] = border_label(m)
    print(i, m.shape, "ids", len(np.unique(m)) - 1, "boundary px", int((out[f'boundary{i}'] == 2).sum()),
          "border px", int((out[f'border{i}'] == 2).sum()))
np.savez_compressed(OUT / "labels_boundary.npz", **out)
print("wrote", OUT / "labels_boundary.npz")
