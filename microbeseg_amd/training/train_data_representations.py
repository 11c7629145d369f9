"""Training-label creation on the MI355X.

Mirror of ``boundary_label`` (:75-99), ``border_label`` (:102-125), ``distance_label`` (:261-361) and the dispatcher
``get_label`` (:12-38) of ``src/training/train_data_representations.py``.  The reference builds every label by looping
over the instances in Python (``binary_dilation(nucleus, 3x3) ^ nucleus``; per-cell crops through scipy's Euclidean
distance transform; two binary closings per cell over the whole image).  Here the boundary / border labels are ONE
streaming kernel over the mask batch (csrc/augment.hip: ``mseg_label_boundary``) and the distance labels a fixed sequence
of passes over the pixel batch (csrc/labels.hip: ``mseg_label_distance``): exact integer distances by a row sweep over
run ends, closings with the radius-3 disk, union-find gap components, per-gap moments and rim sums by atomics.
``get_label`` also serves the reference's ``cell_dist`` / ``cell_dist_clipped`` (``cell_distance_label``) and ``j4`` types and
raises for ``adapted_border`` (OpenCV Canny edges), which microbeSEG's training never requests.  No CPU fallback (oracle/labels_ref.py is the CPU checker of the tests).
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


def distance_label_batch(masks, search_radius):
    """ Cell and neighbor distance labels of a batch of equally sized instance masks (N, H, W) in ONE call.
    Returns (cell distances, neighbor distances), float32 numpy arrays (N, H, W). """
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd label creation needs the MI355X HIP path (no CPU fallback)")
    lib = _lib.load()
    a = np.ascontiguousarray(masks)
    if a.ndim != 3:
        raise RuntimeError("expected a batch (N, H, W) of intensity-coded instance masks")
    if a.size == 0:
        raise RuntimeError("empty mask batch")
    if a.min() < 0 or a.max() > 65535:
        raise RuntimeError("instance ids must fit uint16")
    if int(search_radius) <= 0:
        raise RuntimeError("search_radius must be positive")
    N, H, W = a.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    st = torch.cuda.current_stream().cuda_stream
    cell = np.empty((N, H, W), np.float32)
    nb = np.empty((N, H, W), np.float32)
    step = 64                                     # images per launch sequence: bounds the per-cell tables (3 MiB / image)
    for i in range(0, N, step):
        n = min(step, N - i)
        need = lib.mseg_label_distance_workspace_bytes(n, H, W)
        if need == 0:
            raise RuntimeError(f"unsupported mask shape {H}x{W}")
        m = torch.from_numpy(a[i:i + n].astype(np.uint16).view(np.int16)).to(dev)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        c = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        d = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        _lib.check(lib.mseg_label_distance(m.data_ptr(), n, H, W, int(search_radius), c.data_ptr(), d.data_ptr(),
                                           ws.data_ptr(), need, st), "label_distance")
        cell[i:i + n] = c.cpu().numpy()
        nb[i:i + n] = d.cpu().numpy()
    return cell, nb


def distance_label(label, search_radius):
    """ Cell and neighbor distance label creation (Euclidean distance).

    :param label: Intensity-coded instance segmentation label image.
    :param search_radius: half size of the window around each cell's centroid in which its distances are computed.
    :return: Cell distance label image, neighbor distance label image (float32).
    """
    a = np.asarray(label)
    if a.ndim != 2:
        raise RuntimeError("expected a 2-D intensity-coded instance mask")
    cell, nb = distance_label_batch(a[None], search_radius)
    return cell[0], nb[0]


def bottom_hat_closing(label):
    """ Bottom-hat-transform based grayscale closing (reference :40-72).

    :param label: Intensity coded label image.
    :return: closed label (only closed regions, numbered like ``measure.label``), closed label (only closed regions,
        1.0, or 0.8 on the rim of gaps with a minor axis length of at least 3), float32.
    """
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd label creation needs the MI355X HIP path (no CPU fallback)")
    lib = _lib.load()
    a = np.ascontiguousarray(label)
    if a.ndim != 2:
        raise RuntimeError("expected a 2-D intensity-coded instance mask")
    if a.min(initial=0) < 0 or a.max(initial=0) > 65535:
        raise RuntimeError("instance ids must fit uint16")
    dev = torch.device("cuda", torch.cuda.current_device())
    H, W = a.shape
    m = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
    root = torch.empty((H, W), dtype=torch.int32, device=dev)
    corr = torch.empty((H, W), dtype=torch.float32, device=dev)
    need = lib.mseg_label_distance_workspace_bytes(1, H, W)
    if need == 0:
        raise RuntimeError("unsupported mask shape")
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.mseg_label_bottom_hat(m.data_ptr(), 1, H, W, root.data_ptr(), corr.data_ptr(), ws.data_ptr(), need,
                                         torch.cuda.current_stream().cuda_stream), "label_bottom_hat")
    # measure.label numbers the components by their first pixel in raster order = the rank of the component's root
    roots, inverse = torch.unique(root, sorted=True, return_inverse=True)      # roots[0] == -1 iff there is background
    ids = inverse
    if roots.numel() and roots[0].item() >= 0:
        ids = inverse + 1                                                      # (no pixel outside the gaps at all)
    return ids.to(torch.int64).cpu().numpy(), corr.cpu().numpy()


def cell_distance_label(label, search_radius, apply_clipping=False, clip_val=5):
    """ Cell distance label creation (Euclidean distance), reference :219-258: the per-cell normalised distance
    transform alone, or (apply_clipping) min(distance, clip_val) / clip_val.  float32 (H, W). """
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd label creation needs the MI355X HIP path (no CPU fallback)")
    lib = _lib.load()
    a = np.ascontiguousarray(label)
    if a.ndim != 2:
        raise RuntimeError("expected a 2-D intensity-coded instance mask")
    if a.min(initial=0) < 0 or a.max(initial=0) > 65535:
        raise RuntimeError("instance ids must fit uint16")
    if int(search_radius) <= 0 or (apply_clipping and clip_val <= 0):
        raise RuntimeError("search_radius and clip_val must be positive")
    H, W = a.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    need = lib.mseg_label_distance_workspace_bytes(1, H, W)
    if need == 0:
        raise RuntimeError(f"unsupported mask shape {H}x{W}")
    m = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = torch.empty((H, W), dtype=torch.float32, device=dev)
    _lib.check(lib.mseg_label_cell_distance(m.data_ptr(), 1, H, W, int(search_radius),
                                            float(clip_val) if apply_clipping else 0.0, out.data_ptr(), ws.data_ptr(), need,
                                            torch.cuda.current_stream().cuda_stream), "label_cell_distance")
    return out.cpu().numpy()


def j4_label(label, k_neighbors=2, se_radius=4):
    """ Pena label creation for the J4 method (0 background, 1 cell, 2 touching, 3 gap), reference :157-216. """
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
    tmp = torch.empty(a.shape, dtype=torch.uint8, device=dev)
    out = torch.empty(a.shape, dtype=torch.uint8, device=dev)
    _lib.check(lib.mseg_label_j4(m.data_ptr(), 1, a.shape[0], a.shape[1], int(k_neighbors), int(se_radius), tmp.data_ptr(),
                                 out.data_ptr(), torch.cuda.current_stream().cuda_stream), "label_j4")
    return out.cpu().numpy()


def max_major_axis_length(mask):
    """ Largest ``regionprops(mask)[i].major_axis_length`` of an instance mask (what create_labels turns into max_mal,
    reference src/training/train.py:73-78); 0.0 for an empty mask. """
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd label creation needs the MI355X HIP path (no CPU fallback)")
    lib = _lib.load()
    a = np.ascontiguousarray(mask)
    if a.ndim != 2:
        raise RuntimeError("expected a 2-D intensity-coded instance mask")
    if a.min(initial=0) < 0 or a.max(initial=0) > 65535:
        raise RuntimeError("instance ids must fit uint16")
    dev = torch.device("cuda", torch.cuda.current_device())
    m = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
    need = lib.mseg_label_major_axis_workspace_bytes(1)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    _lib.check(lib.mseg_label_max_major_axis(m.data_ptr(), 1, a.shape[0], a.shape[1], out.data_ptr(), ws.data_ptr(), need,
                                             torch.cuda.current_stream().cuda_stream), "label_max_major_axis")
    return float(out.item())


def get_label(mask, label_type, max_mal):
    """ Training data representation of an instance mask (reference train_data_representations.py:12-38). """
    if label_type == 'boundary':
        return boundary_label(mask)
    if label_type == 'border':
        return border_label(mask)
    if label_type == 'distance':
        return distance_label(mask, search_radius=int(np.ceil(0.75 * max_mal)))
    if label_type == 'cell_dist':
        return cell_distance_label(mask, search_radius=int(np.ceil(0.75 * max_mal)))
    if label_type == 'cell_dist_clipped':
        return cell_distance_label(mask, search_radius=int(np.ceil(0.75 * max_mal)), apply_clipping=True)
    if label_type == 'j4':
        return j4_label(mask)
    if label_type in ('adapted_border',):
        raise RuntimeError(f"label type '{label_type}' is not part of the MI355X build yet (SURVEY.md §8f n2): create it "
                           "with the reference's train_data_representations.py")
    raise Exception('Label type not known')
