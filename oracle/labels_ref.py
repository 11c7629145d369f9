"""TEST INFRASTRUCTURE ONLY — numpy restatement of the label creation of the boundary method (first
part): boundary_label / border_label of src/training/train_data_representations.py (:75-99, :102-125).  Pinned by
tests/golden/labels_boundary.npz (tools/gen_golden_labels.py runs the real reference functions).

The reference loops over the instances (binary_dilation(nucleus, 3x3) ^ nucleus, OR-ed); per pixel that is:
  boundary(p) = some 8-neighbour q of p carries an instance id different from label(p)   (outside the image: nothing)
  outer(p)    = label(p) == 0 and some 8-neighbour is foreground
  boundary_label = 2 where boundary, else 1 where label > 0, else 0
  border_label   = 2 where boundary XOR outer (= foreground pixels that touch ANOTHER instance), else 1 where label > 0
"""
import numpy as np


def _neighbour_flags(label):
    lab = np.asarray(label).astype(np.int64)
    H, W = lab.shape
    pad = np.zeros((H + 2, W + 2), np.int64)
    pad[1:-1, 1:-1] = lab
    other = np.zeros((H, W), bool)      # a neighbour with a different, non-zero id
    anyfg = np.zeros((H, W), bool)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dy == 1 and dx == 1:
                continue
            q = pad[dy:dy + H, dx:dx + W]
            other |= (q > 0) & (q != lab)
            anyfg |= q > 0
    return lab, other, anyfg


def boundary_label(label):
    lab, other, _ = _neighbour_flags(label)
    return np.maximum(lab > 0, 2 * other).astype(np.uint8)


def border_label(label):
    lab, other, anyfg = _neighbour_flags(label)
    outer = (lab == 0) & anyfg
    return np.maximum(lab > 0, 2 * (other ^ outer)).astype(np.uint8)


# ---- distance labels (second part) -----------------------------------------------------------------
# Restatement of bottom_hat_closing (:40-72) and distance_label (:261-361) of the same reference file with numpy +
# scipy.ndimage only (scikit-image is not in the GPU box's interpreter: measure.label -> ndimage.label with the full
# 3x3 structure, regionprops -> the moment formulas below).  Pinned by tests/golden/labels_distance.npz
# (tools/gen_golden_distlabel.py runs the real reference functions).

def _disk3():
    r = np.arange(-3, 4)
    return (r[:, None] ** 2 + r[None, :] ** 2) <= 9          # skimage.morphology.disk(3): 29 pixels


def _minor_axis_length(ys, xs):
    """regionprops.minor_axis_length: 4 * sqrt(smaller eigenvalue of the normalised second central moments)"""
    y = ys.astype(np.float64) - ys.mean()
    x = xs.astype(np.float64) - xs.mean()
    a, c, b = (y * y).mean(), (x * x).mean(), (x * y).mean()
    l2 = 0.5 * (a + c) - 0.5 * np.sqrt(4 * b * b + (a - c) ** 2)
    return 4.0 * np.sqrt(max(l2, 0.0))


def bottom_hat_closing(label):
    """-> (gap components int32 (ids in scipy's order), gap weights float32: 1, or 0.8 on the rim of wide gaps)"""
    from scipy import ndimage as ndi
    lab = np.asarray(label)
    se = _disk3()
    filled = np.zeros(lab.shape, bool)
    for k in np.unique(lab):
        if k > 0:
            filled |= ndi.binary_closing(lab == k, se)        # per cell, so that touching cells do not fuse artefacts
    gaps = ndi.binary_closing(filled, se) & ~filled
    comp, n = ndi.label(gaps, structure=np.ones((3, 3), bool))
    weight = gaps.astype(np.float32)
    cross = ndi.generate_binary_structure(2, 1)
    for g in range(1, n + 1):
        one = comp == g
        ys, xs = np.nonzero(one)
        if _minor_axis_length(ys, xs) >= 3:
            weight[one & ~ndi.binary_erosion(one, cross)] = 0.8
    return comp.astype(np.int32), weight


def distance_label(label, search_radius):
    """-> (cell distances float32, neighbour distances float32), both (H, W)"""
    from scipy import ndimage as ndi
    lab = np.asarray(label).astype(np.int64)
    H, W = lab.shape
    sr = int(search_radius)
    cell = np.zeros((H, W), np.float64)
    nb = np.zeros((H, W), np.float64)
    for k in np.unique(lab):
        if k == 0:
            continue
        ys, xs = np.nonzero(lab == k)
        cy, cx = np.round(ys.mean()), np.round(xs.mean())     # regionprops centroid, rounded half-to-even
        y0, y1 = int(max(cy - sr, 0)), int(min(cy + sr, H))
        x0, x1 = int(max(cx - sr, 0)), int(min(cx + sr, W))
        win = lab[y0:y1, x0:x1]
        own = win == k
        d = ndi.distance_transform_edt(own)                   # to the nearest pixel of the window that is not this cell
        dmax = d.max() if d.size else 0.0
        if not dmax > 0:
            continue
        cell[y0:y1, x0:x1] += d / dmax
        others = (win != 0) & ~own
        if not others.any():
            continue
        dn = ndi.distance_transform_edt(~others) * own        # to the nearest pixel of ANOTHER cell inside the window
        if dn.max() > 0:
            frac = np.clip(dn / min(dmax + 3, dn.max()), 0, 1)
        else:
            frac = 1
        nb[y0:y1, x0:x1] += (1 - frac) * own
    comp, weight = bottom_hat_closing(lab)
    for g in range(1, comp.max() + 1):
        one = comp == g
        area = int(one.sum())
        ring = ndi.binary_dilation(one, np.ones((3, 3), bool)) & ~one
        th = 5 if area <= 20 else 8 if area <= 30 else 10 if area <= 50 else 20
        if nb[ring].sum() < th:                               # nothing but background around it: an artefact
            weight[one] = 0
    nb = np.maximum(nb, weight.astype(np.float64))
    nb = np.maximum(nb, (border_label(lab) == 2).astype(np.float64))
    nb = np.clip(1 / np.sqrt(0.65 + 0.5 * np.exp(-11 * (nb - 0.75))) - 0.19, 0, 1)
    nb = ndi.grey_closing(nb, size=(3, 3))
    return cell.astype(np.float32), nb.astype(np.float32)


def major_axis_lengths(label):
    """regionprops(label)[i].major_axis_length in ascending id order (the label creation takes ceil(max) as max_mal,
    src/training/train.py:73-78): 4 * sqrt(larger eigenvalue of the normalised second central moments)"""
    lab = np.asarray(label)
    out = []
    for k in np.unique(lab):
        if k == 0:
            continue
        ys, xs = np.nonzero(lab == k)
        y = ys.astype(np.float64) - ys.mean()
        x = xs.astype(np.float64) - xs.mean()
        a, c, b = (y * y).mean(), (x * x).mean(), (x * y).mean()
        out.append(4.0 * np.sqrt(0.5 * (a + c) + 0.5 * np.sqrt(4 * b * b + (a - c) ** 2)))
    return np.array(out, np.float64)


def cell_distance_label(label, search_radius, apply_clipping=False, clip_val=5):
    """cell_distance_label (:219-258): the per-cell windowed distance transform, normalised per cell or clipped"""
    from scipy import ndimage as ndi
    lab = np.asarray(label).astype(np.int64)
    H, W = lab.shape
    sr = int(search_radius)
    out = np.zeros((H, W), np.float64)
    for k in np.unique(lab):
        if k == 0:
            continue
        ys, xs = np.nonzero(lab == k)
        cy, cx = np.round(ys.mean()), np.round(xs.mean())
        y0, y1 = int(max(cy - sr, 0)), int(min(cy + sr, H))
        x0, x1 = int(max(cx - sr, 0)), int(min(cx + sr, W))
        d = ndi.distance_transform_edt(lab[y0:y1, x0:x1] == k)
        if d.size and d.max() > 0 and not apply_clipping:
            d = d / d.max()
        out[y0:y1, x0:x1] += d
    if apply_clipping:
        out = np.clip(out, 0, clip_val) / clip_val
    return out.astype(np.float32)


def j4_label(label, k_neighbors=2, se_radius=4):
    """j4_label (:157-216): 0 background, 1 cell, 2 touching, 3 gap"""
    from scipy import ndimage as ndi
    lab = np.asarray(label).astype(np.int64)
    H, W = lab.shape
    fg = lab > 0
    r = np.arange(-se_radius, se_radius + 1)
    se = (r[:, None] ** 2 + r[None, :] ** 2) <= se_radius ** 2
    gap = (ndi.binary_closing(fg, se) ^ fg) & ~fg
    k = k_neighbors
    pad = np.zeros((H + 2 * k, W + 2 * k), np.int64)
    pad[k:k + H, k:k + W] = lab
    other = np.zeros((H, W), bool)                    # another instance id somewhere in the (2k+1)^2 window
    for dy in range(2 * k + 1):
        for dx in range(2 * k + 1):
            q = pad[dy:dy + H, dx:dx + W]
            other |= (q > 0) & (q != lab)
    out = np.zeros((H, W), np.uint8)
    out[fg] = 1
    out[fg & other] = 2
    out[gap] = 3
    return out
