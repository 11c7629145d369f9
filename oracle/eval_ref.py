"""TEST INFRASTRUCTURE ONLY — CPU restatement of the evaluation helpers around the hot path.

Used by tests/, never by the product path (which runs csrc/evalk.hip through the C ABI).  Each function cites the
reference code it follows; all of them are pinned by tests/golden/eval_aji.npz, which tools/gen_golden_eval.py produced
with the real reference functions (scikit-image 0.18.3 / scipy 1.7.1).

  label_image        skimage.measure.label(label_image) as called in src/evaluation/eval.py:253,304
  border_correction  src/utils/utils.py:25-47
  aji_plus           get_fast_aji_plus, src/evaluation/stats_utils.py:98-179
"""
import numpy as np
from scipy import ndimage as ndi
from scipy.optimize import linear_sum_assignment


def label_image(img):
    """Connected components of a label image: 8-neighbours with EQUAL non-zero value connect, background 0, new ids
    1..K in raster order of each component's first pixel (skimage.measure.label defaults for 2-D input)."""
    img = np.asarray(img)
    out = np.zeros(img.shape, np.int64)
    nxt = 0
    st = np.ones((3, 3), bool)
    for v in np.unique(img):
        if v == 0:
            continue
        lab, k = ndi.label(img == v, structure=st)
        out[lab > 0] = lab[lab > 0] + nxt
        nxt += k
    # renumber by first occurrence in raster order
    flat = out.ravel()
    ids, first = np.unique(flat, return_index=True)
    order = np.argsort(first[ids > 0])
    remap = np.zeros(nxt + 1, np.int64)
    remap[ids[ids > 0][order]] = np.arange(1, len(order) + 1)
    return remap[out]


def border_correction(mask, border_width=10):
    """Delete instances that are not visible inside the field of interest (the frame minus a border)."""
    mask = np.array(mask, copy=True)
    foi = mask[border_width:mask.shape[0] - border_width, border_width:mask.shape[1] - border_width]
    keep = np.unique(foi)
    keep = keep[keep > 0]
    ids = np.unique(mask)
    ids = ids[ids > 0]
    for i in ids:
        if i not in keep:
            mask[mask == i] = 0
    return mask


def pair_counts(true, pred):
    """areas and pairwise intersections of two contiguous label images: (area_t[nt+1], area_p[np+1], inter[nt+1][np+1])"""
    true = np.asarray(true).astype(np.int64).ravel()
    pred = np.asarray(pred).astype(np.int64).ravel()
    nt, npd = int(true.max(initial=0)), int(pred.max(initial=0))
    inter = np.bincount(true * (npd + 1) + pred, minlength=(nt + 1) * (npd + 1)).reshape(nt + 1, npd + 1)
    return np.bincount(true, minlength=nt + 1), np.bincount(pred, minlength=npd + 1), inter


def aji_from_counts(area_t, area_p, inter):
    """get_fast_aji_plus from the integer counts (ids contiguous 1..n, as measure.label returns them)."""
    nt, npd = len(area_t) - 1, len(area_p) - 1
    pairwise_inter = inter[1:, 1:].astype(np.float64)
    # union = |t| + |p| - inter for overlapping pairs, 0 elsewhere (the reference fills only overlapping pairs)
    pairwise_union = (area_t[1:, None] + area_p[None, 1:]).astype(np.float64) - pairwise_inter
    pairwise_union[pairwise_inter == 0] = 0.0
    pairwise_iou = pairwise_inter / (pairwise_union + 1.0e-6)
    paired_true, paired_pred = linear_sum_assignment(-pairwise_iou)
    paired_iou = pairwise_iou[paired_true, paired_pred]
    paired_true = paired_true[paired_iou > 0.0]
    paired_pred = paired_pred[paired_iou > 0.0]
    overall_inter = pairwise_inter[paired_true, paired_pred].sum()
    overall_union = pairwise_union[paired_true, paired_pred].sum()
    unpaired_t = np.setdiff1d(np.arange(nt), paired_true)
    unpaired_p = np.setdiff1d(np.arange(npd), paired_pred)
    overall_union += area_t[1:][unpaired_t].sum() + area_p[1:][unpaired_p].sum()
    return overall_inter / overall_union


def aji_plus(true, pred):
    return aji_from_counts(*pair_counts(true, pred))


def score_pair(prediction, ground_truth):
    """One test image of EvalWorker.calc_scores (eval.py:248-256): border correction, relabel, AJI+ (0 for an empty
    prediction)."""
    p = border_correction(prediction)
    g = border_correction(ground_truth)
    if np.max(p) > 0:
        return aji_plus(label_image(g), label_image(p))
    return 0
