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
