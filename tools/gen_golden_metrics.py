#!/opt/conda/bin/python3.9
"""Generate tests/golden/eval_metrics.npz with the REAL reference metric helpers of src/evaluation/stats_utils.py
(get_fast_aji :16-95, get_fast_pq :183-285, get_fast_dice_2 :288-325, get_dice_1 :328-338, get_dice_2 :341-362,
remap_label :365-395, pair_coordinates :398-437) on the label pairs of tests/golden/eval_aji.npz (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tools/gen_golden_metrics.py
cv2 is imported by the module without being used by these functions: an empty placeholder satisfies the import.
Only inputs and the numbers the reference produced are stored."""
import pathlib
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
from src.evaluation.stats_utils import (get_fast_aji, get_fast_pq, get_fast_dice_2, get_dice_1, get_dice_2,  # noqa: E402
                                        remap_label, pair_coordinates)

G = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
src = np.load(G / "eval_aji.npz")
out = {}
for c in range(5):                               # case 5 has an empty / single-instance prediction
    gl, pl = src[f"c{c}_gt_lab"], src[f"c{c}_pred_lab"]
    out[f"c{c}_aji"] = np.float64(get_fast_aji(gl, pl))
    for tag, thr in (("pq50", 0.5), ("pq30", 0.3)):
        (dq, sq, pq), (pt, pp, ut, up) = get_fast_pq(gl, pl, match_iou=thr)
        out[f"c{c}_{tag}"] = np.array([dq, sq, pq], np.float64)
        out[f"c{c}_{tag}_pt"] = np.asarray(pt, np.int64)
        out[f"c{c}_{tag}_pp"] = np.asarray(pp, np.int64)
        out[f"c{c}_{tag}_ut"] = np.asarray(ut, np.int64)
        out[f"c{c}_{tag}_up"] = np.asarray(up, np.int64)
    out[f"c{c}_dice2"] = np.float64(get_fast_dice_2(gl, pl))
    out[f"c{c}_dice2_slow"] = np.float64(get_dice_2(gl, pl))
    out[f"c{c}_dice1"] = np.float64(get_dice_1(gl, pl))
    raw = src[f"c{c}_pred"]                      # non-contiguous ids
    out[f"c{c}_remap"] = remap_label(raw.astype(np.int32)).astype(np.int32)
    out[f"c{c}_remap_size"] = remap_label(raw.astype(np.int32), by_size=True).astype(np.int32)
    print(c, "aji", out[f"c{c}_aji"], "pq", out[f"c{c}_pq50"], "dice2", out[f"c{c}_dice2"], "dice1", out[f"c{c}_dice1"])
rng = np.random.Generator(np.random.PCG64(9))
A = rng.uniform(0, 100, (30, 2)).astype(np.float32)
B = (A[rng.permutation(30)[:24]] + rng.normal(0, 2.0, (24, 2))).astype(np.float32)
pairing, ua, ub = pair_coordinates(A, B, 4.0)
out["pc_A"], out["pc_B"], out["pc_pairing"], out["pc_ua"], out["pc_ub"] = A, B, pairing.astype(np.int64), ua, ub
np.savez_compressed(G / "eval_metrics.npz", **out)
print("wrote", G / "eval_metrics.npz")
