"""CPU: oracle/eval_ref.py against vectors produced by the real reference (tools/gen_golden_eval.py): skimage
measure.label on label images, utils.border_correction, stats_utils.get_fast_aji_plus."""
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import eval_ref  # noqa: E402

G = np.load(ROOT / "tests" / "golden" / "eval_aji.npz")
CASES = sorted({k.split("_")[0] for k in G.files if k.startswith("c")})


def test_label_image_known_answer():
    assert np.array_equal(eval_ref.label_image(G["label_in"]), G["label_out"])


@pytest.mark.parametrize("c", CASES)
def test_border_correction_label_and_aji(c):
    gt, pred = G[f"{c}_gt"], G[f"{c}_pred"]
    g2, p2 = eval_ref.border_correction(gt), eval_ref.border_correction(pred)
    assert np.array_equal(g2, G[f"{c}_gt_bc"]) and np.array_equal(p2, G[f"{c}_pred_bc"])
    gl, pl = eval_ref.label_image(g2), eval_ref.label_image(p2)
    assert np.array_equal(gl, G[f"{c}_gt_lab"]) and np.array_equal(pl, G[f"{c}_pred_lab"])
    want = float(G[f"{c}_aji"])
    got = eval_ref.score_pair(pred, gt)
    assert got == pytest.approx(want, rel=1e-12, abs=1e-15)


def test_border_width():
    assert np.array_equal(eval_ref.border_correction(G["bc_in"]), G["bc_w10"])
    assert np.array_equal(eval_ref.border_correction(G["bc_in"], border_width=3), G["bc_w3"])


GM = np.load(ROOT / "tests" / "golden" / "eval_metrics.npz")


@pytest.mark.parametrize("c", range(5))
def test_other_metrics_oracle_matches_reference(c):
    """get_fast_aji / get_fast_pq / dice / remap_label restatements vs the real reference functions"""
    gl, pl = G[f"c{c}_gt_lab"], G[f"c{c}_pred_lab"]
    assert abs(eval_ref.fast_aji(gl, pl) - float(GM[f"c{c}_aji"])) < 1e-12
    for tag, thr in (("pq50", 0.5), ("pq30", 0.3)):
        v, pt, pp = eval_ref.fast_pq(gl, pl, thr)
        assert np.allclose(v, GM[f"c{c}_{tag}"], rtol=0, atol=1e-12)
        assert np.array_equal(pt, GM[f"c{c}_{tag}_pt"]) and np.array_equal(pp, GM[f"c{c}_{tag}_pp"])
    assert abs(eval_ref.dice_2(gl, pl) - float(GM[f"c{c}_dice2"])) < 1e-12
    assert abs(float(GM[f"c{c}_dice2_slow"]) - float(GM[f"c{c}_dice2"])) < 1e-12
    assert abs(eval_ref.dice_1(gl, pl) - float(GM[f"c{c}_dice1"])) < 1e-12
    assert np.array_equal(eval_ref.remap_label(G[f"c{c}_pred"]), GM[f"c{c}_remap"])
    assert np.array_equal(eval_ref.remap_label(G[f"c{c}_pred"], by_size=True), GM[f"c{c}_remap_size"])
