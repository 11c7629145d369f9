"""Training-label creation, the part that is a pure neighbourhood rule (first part).

Mirror of ``boundary_label`` (:75-99), ``border_label`` (:102-125) and the dispatcher ``get_label`` (:12-38) of
``src/training/train_data_representations.py``.  The reference builds both labels by looping over the instances
(``binary_dilation(nucleus, 3x3) ^ nucleus``); on the MI355X they are ONE streaming kernel over the mask batch
(csrc/augment.hip: ``mseg_label_boundary``), exact.  The distance labels (``distance_label``: per-cell Euclidean distance
transforms, bottom-hat gap filling, grey closing) are not part of this build yet — ``get_label`` raises for them, as it
does for the reference's experimental label types.  No CPU fallback (oracle/labels_ref.py is the CPU checker of the tests).
"""
import numpy as np
import torch

from .. import _lib


def _label_device(label, mode):
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd label creation needs the MI355X HIP path (no CPU fallback)")
    lib = _lib.load()
    a = np.ascontiguousarray(label)
    if a.ndim != 2:
        raise RuntimeError("expected a 2-D intensity-coded instance mask")
    if a.min(initial=0) < 0 or a.max(initial=0) > 65535:
        raise RuntimeError("instance ids must fit uint16")
    dev = torch.device("cuda", torch.cuda.current_device())
    m = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
    out = torch.empty(a.shape, dtype=torch.uint8, device=dev)
    _lib.check(lib.mseg_label_boundary(m.data_ptr(), 1, a.shape[0], a.shape[1], mode, out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "label_boundary")
    return out.cpu().numpy()


def boundary_label(label):
    """ Boundary label image: 0 background, 1 cell interior, 2 boundary (the 3x3 dilation ring of every instance). """
    return _label_device(label, 0)


def border_label(label):
    """ Border label image: 0 background, 1 cell, 2 borders in-between touching cells. """
    return _label_device(label, 1)


def get_label(mask, label_type, max_mal):
    """ Training data representation of an instance mask (reference train_data_representations.py:12-38). """
    if label_type == 'boundary':
        return boundary_label(mask)
    if label_type == 'border':
        return border_label(mask)
    if label_type in ('adapted_border', 'j4', 'cell_dist', 'cell_dist_clipped', 'distance'):
        raise RuntimeError(f"label type '{label_type}' is not part of the MI355X build yet (SURVEY.md §8f n2): create it "
                           "with the reference's train_data_representations.py")
    raise Exception('Label type not known')
