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


# ---- the other metrics of stats_utils.py (get_fast_aji :16-95, get_fast_pq :183-285, dice :288-362, remap_label :365-395),
# restated on the pair statistics; pinned by tests/golden/eval_metrics.npz (tools/gen_golden_metrics.py) ------------------
def _pair_stats(true, pred):
    t, p = np.asarray(true).astype(np.int64), np.asarray(pred).astype(np.int64)
    nt, npd = int(t.max()), int(p.max())
    inter = np.zeros((nt + 1, npd + 1), np.int64)
    np.add.at(inter, (t.ravel(), p.ravel()), 1)
    return inter.sum(1), inter.sum(0), inter


def fast_aji(true, pred):
    at, ap, inter = _pair_stats(true, pred)
    i = inter[1:, 1:].astype(np.float64)
    u = np.where(i > 0, at[1:, None] + ap[None, 1:] - i, 0.0)
    iou = i / (u + 1.0e-6)
    best, val = iou.argmax(1), iou.max(1)
    rows = np.nonzero(val > 0)[0]
    cols = best[rows]
    union = u[rows, cols].sum() + at[1:][np.setdiff1d(np.arange(len(at) - 1), rows)].sum() + \
        ap[1:][np.setdiff1d(np.arange(len(ap) - 1), cols)].sum()
    return i[rows, cols].sum() / union


def fast_pq(true, pred, match_iou=0.5):
    from scipy.optimize import linear_sum_assignment
    at, ap, inter = _pair_stats(true, pred)
    i = inter[1:, 1:].astype(np.float64)
    iou = np.where(i > 0, i / (at[1:, None] + ap[None, 1:] - i), 0.0)
    if match_iou >= 0.5:
        rows, cols = np.nonzero(iou > match_iou)
    else:
        rows, cols = linear_sum_assignment(-iou)
        keep = iou[rows, cols] > match_iou
        rows, cols = rows[keep], cols[keep]
    tp = len(rows)
    fn, fp = (len(at) - 1) - len(set(rows)), (len(ap) - 1) - len(set(cols))
    dq = tp / (tp + 0.5 * fp + 0.5 * fn)
    sq = iou[rows, cols].sum() / (tp + 1.0e-6)
    return np.array([dq, sq, dq * sq]), rows + 1, cols + 1


def dice_2(true, pred):
    at, ap, inter = _pair_stats(true, pred)
    i = inter[1:, 1:]
    return 2 * i.sum() / ((at[1:, None] + ap[None, 1:]) * (i > 0)).sum()


def dice_1(true, pred):
    t, p = np.asarray(true) > 0, np.asarray(pred) > 0
    return 2.0 * (t & p).sum() / (t.sum() + p.sum())


def remap_label(pred, by_size=False):
    pred = np.asarray(pred)
    ids = [i for i in np.unique(pred) if i != 0]
    if by_size:
        ids = sorted(ids, key=lambda i: -(pred == i).sum())      # stable: equal sizes keep id order
    out = np.zeros(pred.shape, np.int32)
    for k, i in enumerate(ids):
        out[pred == i] = k + 1
    return out
