"""AJI+ of the evaluation, same call surface as the reference's ``src/evaluation/stats_utils.py`` (get_fast_aji_plus
:98-179, from HoVer-Net) plus the fused per-image scoring of ``EvalWorker.calc_scores`` (eval.py:248-256).

MI355X path: the pixel work — border correction, connected-component relabelling and the area / pairwise-intersection
counts — runs in csrc/postproc.hip (mseg_eval_relabel, mseg_eval_pair_counts); what remains on the host is the maximal
unique pairing on the small IoU matrix, for which the reference itself calls scipy.optimize.linear_sum_assignment.
No CPU fallback: without a GPU these functions raise (oracle/eval_ref.py is the CPU checker used by the tests).
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .. import _lib

_ws_cache = {}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd evaluation needs the MI355X HIP path (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(dev, H, W):
    lib = _lib.load()
    need = lib.mseg_eval_workspace_bytes(H, W)
    if need == 0:
        raise RuntimeError(f"unsupported frame size {H}x{W}")
    ws = _ws_cache.get(str(dev))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _ws_cache[str(dev)] = ws
    return ws, need


def _to_u16_device(mask, dev):
    if isinstance(mask, torch.Tensor):
        if mask.dtype != torch.int16:
            raise RuntimeError("device masks are int16 views of uint16 labels")
        return mask.to(dev).contiguous()
    m = np.ascontiguousarray(mask)
    if m.min(initial=0) < 0 or m.max(initial=0) > 65535:
        raise RuntimeError("instance masks must fit uint16")
    return torch.from_numpy(m.astype(np.uint16).view(np.int16)).to(dev)


def relabel_device(mask, border_width=10):
    """border_correction(mask, border_width) followed by skimage.measure.label, on the device.
    mask: (H, W) numpy integer array or int16 CUDA tensor holding uint16 labels.  Returns (int32 CUDA labels, K)."""
    lib = _lib.load()
    dev = mask.device if isinstance(mask, torch.Tensor) and mask.is_cuda else _device()
    m = _to_u16_device(mask, dev)
    H, W = m.shape
    ws, need = _workspace(dev, H, W)
    out = torch.empty((H, W), dtype=torch.int32, device=dev)
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.mseg_eval_relabel(m.data_ptr(), H, W, int(border_width), out.data_ptr(), n.data_ptr(), ws.data_ptr(),
                                     need, _stream()), "eval_relabel")
    return out, int(n.item())


def pair_counts_device(true_lab, pred_lab, nt, npd):
    """areas and pairwise intersections of two contiguous int32 CUDA label images -> numpy int64 arrays"""
    lib = _lib.load()
    H, W = true_lab.shape
    dev = true_lab.device
    at = torch.empty(nt + 1, dtype=torch.int32, device=dev)
    ap = torch.empty(npd + 1, dtype=torch.int32, device=dev)
    inter = torch.empty((nt + 1, npd + 1), dtype=torch.int32, device=dev)
    _lib.check(lib.mseg_eval_pair_counts(true_lab.data_ptr(), pred_lab.data_ptr(), H, W, nt, npd, at.data_ptr(),
                                         ap.data_ptr(), inter.data_ptr(), _stream()), "eval_pair_counts")
    return at.cpu().numpy().astype(np.int64), ap.cpu().numpy().astype(np.int64), inter.cpu().numpy().astype(np.int64)


def _aji_from_counts(area_t, area_p, inter):
    """stats_utils.py:134-178 on the integer counts (ids contiguous 1..n)."""
    nt, npd = len(area_t) - 1, len(area_p) - 1
    pairwise_inter = inter[1:, 1:].astype(np.float64)
    pairwise_union = (area_t[1:, None] + area_p[None, 1:]).astype(np.float64) - pairwise_inter
    pairwise_union[pairwise_inter == 0] = 0.0          # the reference fills overlapping pairs only
    pairwise_iou = pairwise_inter / (pairwise_union + 1.0e-6)
    paired_true, paired_pred = linear_sum_assignment(-pairwise_iou)
    paired_iou = pairwise_iou[paired_true, paired_pred]
    paired_true = paired_true[paired_iou > 0.0]
    paired_pred = paired_pred[paired_iou > 0.0]
    overall_inter = pairwise_inter[paired_true, paired_pred].sum()
    overall_union = pairwise_union[paired_true, paired_pred].sum()
    overall_union += area_t[1:][np.setdiff1d(np.arange(nt), paired_true)].sum()
    overall_union += area_p[1:][np.setdiff1d(np.arange(npd), paired_pred)].sum()
    return overall_inter / overall_union


def get_fast_aji_plus(true, pred):
    """AJI+ of two label images with contiguous ids 1..n (what skimage.measure.label returns); numpy arrays or int32
    CUDA tensors.  Same value as the reference function."""
    dev = true.device if isinstance(true, torch.Tensor) and true.is_cuda else _device()

    def prep(a):
        if isinstance(a, torch.Tensor):
            return a.to(device=dev, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)
    t, p = prep(true), prep(pred)
    nt, npd = int(t.max().item()), int(p.max().item())
    return _aji_from_counts(*pair_counts_device(t, p, nt, npd))


def aji_plus_masks(prediction, ground_truth, border_width=10):
    """One test image of EvalWorker.calc_scores (eval.py:248-256): border correction of both masks, relabel, AJI+
    (0 when nothing of the prediction survives the border correction)."""
    p_lab, n_p = relabel_device(prediction, border_width)
    if n_p == 0:
        return 0
    g_lab, n_g = relabel_device(ground_truth, border_width)
    return _aji_from_counts(*pair_counts_device(g_lab, p_lab, n_g, n_p))


# ---- the other HoVer-Net metrics of the reference module (stats_utils.py:16-95, 183-437) -------------------------------------
# EvalWorker only calls get_fast_aji_plus; these share its integer statistics (areas + pairwise intersections from
# mseg_eval_pair_counts), so each is a short formula on the small (instances x instances) matrices.  Like the reference
# functions they expect contiguous ids 1..n (call remap_label first).

def _prep_pair(true, pred):
    dev = true.device if isinstance(true, torch.Tensor) and true.is_cuda else _device()

    def prep(a):
        if isinstance(a, torch.Tensor):
            return a.to(device=dev, dtype=torch.int32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(dev)
    t, p = prep(true), prep(pred)
    nt, npd = int(t.max().item()), int(p.max().item())
    return pair_counts_device(t, p, nt, npd)


def get_fast_aji(true, pred):
    """AJI as distributed by MoNuSeg (reference :16-95): every ground-truth instance is paired with the prediction of
    highest IoU, predictions may be reused."""
    area_t, area_p, inter = _prep_pair(true, pred)
    nt, npd = len(area_t) - 1, len(area_p) - 1
    pairwise_inter = inter[1:, 1:].astype(np.float64)
    pairwise_union = (area_t[1:, None] + area_p[None, 1:]).astype(np.float64) - pairwise_inter
    pairwise_union[pairwise_inter == 0] = 0.0
    pairwise_iou = pairwise_inter / (pairwise_union + 1.0e-6)
    best = np.argmax(pairwise_iou, axis=1)            # raises on an empty prediction, like the reference
    best_iou = np.max(pairwise_iou, axis=1)
    paired_true = np.nonzero(best_iou > 0.0)[0]
    paired_pred = best[paired_true]
    overall_inter = pairwise_inter[paired_true, paired_pred].sum()
    overall_union = pairwise_union[paired_true, paired_pred].sum()
    overall_union += area_t[1:][np.setdiff1d(np.arange(nt), paired_true)].sum()
    overall_union += area_p[1:][np.setdiff1d(np.arange(npd), paired_pred)].sum()
    return overall_inter / overall_union


def get_fast_pq(true, pred, match_iou=0.5):
    """Panoptic quality (reference :183-285).  Returns [dq, sq, pq], [paired_true, paired_pred, unpaired_true,
    unpaired_pred] (instance ids)."""
    assert match_iou >= 0.0, "Cant' be negative"
    area_t, area_p, inter = _prep_pair(true, pred)
    nt, npd = len(area_t) - 1, len(area_p) - 1
    pairwise_inter = inter[1:, 1:].astype(np.float64)
    total = (area_t[1:, None] + area_p[None, 1:]).astype(np.float64)
    pairwise_iou = np.where(pairwise_inter > 0, pairwise_inter / (total - pairwise_inter), 0.0)
    if match_iou >= 0.5:                              # IoU > 0.5 pairs are unique and maximal
        pairwise_iou[pairwise_iou <= match_iou] = 0.0
        paired_true, paired_pred = np.nonzero(pairwise_iou)
        paired_iou = pairwise_iou[paired_true, paired_pred]
        paired_true = paired_true + 1
        paired_pred = paired_pred + 1
    else:                                             # maximal unique pairing, then threshold
        rows, cols = linear_sum_assignment(-pairwise_iou)
        paired_iou = pairwise_iou[rows, cols]
        keep = paired_iou > match_iou
        paired_true = list(rows[keep] + 1)
        paired_pred = list(cols[keep] + 1)
        paired_iou = paired_iou[keep]
    unpaired_true = [i for i in range(1, nt + 1) if i not in paired_true]
    unpaired_pred = [i for i in range(1, npd + 1) if i not in paired_pred]
    tp, fp, fn = len(paired_true), len(unpaired_pred), len(unpaired_true)
    dq = tp / (tp + 0.5 * fp + 0.5 * fn)
    sq = paired_iou.sum() / (tp + 1.0e-6)
    return [dq, sq, dq * sq], [paired_true, paired_pred, unpaired_true, unpaired_pred]


def get_fast_dice_2(true, pred):
    """Ensemble dice (reference :288-325): over all overlapping (true, pred) pairs."""
    area_t, area_p, inter = _prep_pair(true, pred)
    pairwise_inter = inter[1:, 1:]
    overlapping = pairwise_inter > 0
    overall_total = ((area_t[1:, None] + area_p[None, 1:]) * overlapping).sum()
    overall_inter = pairwise_inter.sum()
    return 2 * overall_inter / overall_total


def get_dice_2(true, pred):
    """Ensemble Dice as used in the Computational Precision Medicine Challenge (reference :341-362): the same number as
    get_fast_dice_2."""
    return get_fast_dice_2(true, pred)


def get_dice_1(true, pred):
    """Traditional dice of the binarised masks (reference :328-338)."""
    area_t, area_p, inter = _prep_pair(true, pred)
    return 2.0 * inter[1:, 1:].sum() / (area_t[1:].sum() + area_p[1:].sum())


def remap_label(pred, by_size=False):
    """Rename the instance ids to 1..n (order preserved, or bigger instances first with by_size), reference :365-395.
    A lookup-table pass on the device (histogram + gather through torch: no arithmetic worth a kernel)."""
    dev = _device()
    a = np.ascontiguousarray(pred)
    if a.min(initial=0) < 0:
        raise RuntimeError("instance ids must be non-negative")
    t = torch.from_numpy(a.astype(np.int64)).to(dev)
    counts = torch.bincount(t.flatten())
    ids = torch.nonzero(counts[1:] > 0).flatten() + 1
    if ids.numel() == 0:
        return pred                                  # no label, like the reference
    if by_size:
        order = torch.sort(counts[ids], descending=True, stable=True).indices
        ids = ids[order]
    lut = torch.zeros(counts.numel(), dtype=torch.int32, device=dev)
    lut[ids] = torch.arange(1, ids.numel() + 1, dtype=torch.int32, device=dev)
    return lut[t].cpu().numpy()


def pair_coordinates(setA, setB, radius):
    """Optimal unique pairing of two point sets within `radius` (reference :398-437; point-set arithmetic on the host, as
    in the reference: scipy's cdist + linear_sum_assignment).  Returns pairing (K, 2), unpairedA, unpairedB."""
    from scipy.spatial.distance import cdist
    cost = cdist(setA, setB, metric='euclidean')
    ia, ib = linear_sum_assignment(cost)
    ok = cost[ia, ib] <= radius
    pa, pb = ia[ok], ib[ok]
    pairing = np.concatenate([pa[:, None], pb[:, None]], axis=-1)
    return pairing, np.delete(np.arange(setA.shape[0]), pa), np.delete(np.arange(setB.shape[0]), pb)
