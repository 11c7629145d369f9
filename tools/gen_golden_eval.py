#!/opt/conda/bin/python3.9
"""Generate tests/golden/eval_*.npz with the REAL reference evaluation helpers (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tools/gen_golden_eval.py
Imports /root/reference/src/evaluation/stats_utils.py (get_fast_aji_plus) and /root/reference/src/utils/utils.py
(border_correction) plus skimage.measure.label of the conda env (scikit-image 0.18.3, scipy 1.7.1), exactly the calls of
EvalWorker.calc_scores (src/evaluation/eval.py:248-254).  stats_utils imports cv2 at module level without using it in
get_fast_aji_plus; cv2 is not installed here, so an empty placeholder module satisfies that import.
Only inputs and the numbers the reference produced for them are stored.
"""
import pathlib
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from src.evaluation.stats_utils import get_fast_aji_plus  # noqa: E402
from src.utils.utils import border_correction  # noqa: E402
from skimage import measure  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def blobs(rng, H, W, n, rmin=4, rmax=12, touching=True):
    """instance mask of random ellipses; ids are NOT contiguous (some removed) and neighbours may touch"""
    mask = np.zeros((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(1, n + 1):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        a, b = rng.uniform(rmin, rmax), rng.uniform(rmin, rmax)
        th = rng.uniform(0, np.pi)
        u = (yy - cy) * np.cos(th) + (xx - cx) * np.sin(th)
        v = -(yy - cy) * np.sin(th) + (xx - cx) * np.cos(th)
        blob = (u / a) ** 2 + (v / b) ** 2 <= 1
        if touching:
            blob &= mask == 0
        mask[blob] = k
    return mask


def perturb(rng, gt, drop=0.15, shift=2):
    """a 'prediction' derived from the ground truth: shifted, some cells dropped / merged / split"""
    H, W = gt.shape
    dy, dx = rng.integers(-shift, shift + 1, 2)
    pred = np.roll(np.roll(gt, dy, 0), dx, 1).copy()
    ids = np.unique(pred)[1:]
    for i in ids:
        r = rng.uniform()
        if r < drop:
            pred[pred == i] = 0
        elif r < drop + 0.1 and len(ids) > 1:
            pred[pred == i] = rng.choice(ids)          # merge into another id (possibly disconnected -> label() splits)
        elif r < drop + 0.2:
            ys, xs = np.nonzero(pred == i)
            if len(ys) > 6:
                half = ys > np.median(ys)
                pred[ys[half], xs[half]] = pred.max() + 1   # split
    return pred


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    out = {}
    cases = [(0, 64, 64, 6), (1, 96, 128, 20), (2, 128, 128, 45), (3, 200, 160, 70), (4, 256, 256, 140), (5, 80, 80, 1)]
    for seed, H, W, n in cases:
        rng = np.random.Generator(np.random.PCG64(500 + seed))
        gt = blobs(rng, H, W, n)
        pred = perturb(rng, gt)
        if seed == 5:
            pred[:] = 0
            pred[20:40, 20:40] = 3
        out[f"c{seed}_gt"] = gt
        out[f"c{seed}_pred"] = pred
        g2, p2 = border_correction(gt.copy()), border_correction(pred.copy())
        out[f"c{seed}_gt_bc"] = g2
        out[f"c{seed}_pred_bc"] = p2
        gl, pl = measure.label(g2), measure.label(p2)
        out[f"c{seed}_gt_lab"] = gl.astype(np.int32)
        out[f"c{seed}_pred_lab"] = pl.astype(np.int32)
        aji = get_fast_aji_plus(true=gl, pred=pl) if np.max(p2) > 0 else 0
        out[f"c{seed}_aji"] = np.float64(aji)
        print(seed, H, W, "gt ids", gl.max(), "pred ids", pl.max(), "aji+", aji)
    # border_correction with a non-default width and labels touching only the border
    rng = np.random.Generator(np.random.PCG64(77))
    m = blobs(rng, 60, 90, 15, rmin=3, rmax=8)
    out["bc_in"] = m
    out["bc_w10"] = border_correction(m.copy())
    out["bc_w3"] = border_correction(m.copy(), border_width=3)
    # measure.label on label images: equal-valued 8-neighbours connect, different values never do
    chk = np.zeros((12, 14), np.int32)
    chk[1:4, 1:4] = 5; chk[4:6, 4:7] = 5; chk[1:3, 8:12] = 2; chk[3:5, 10:13] = 7; chk[8:11, 2:5] = 5; chk[7, 5] = 5
    chk[9:11, 8:10] = 2; chk[8, 10] = 2
    out["label_in"] = chk
    out["label_out"] = measure.label(chk).astype(np.int32)
    np.savez_compressed(OUT / "eval_aji.npz", **out)
    print("wrote", OUT / "eval_aji.npz")


if __name__ == "__main__":
    main()
