"""Prediction -> instance labels on the MI355X (gaussian smoothing, thresholds, 8-connected seed labelling, small-seed
removal, marker-based watershed), bit-identical to the reference CPU chain.

Mirror of ``src/inference/postprocessing.py`` (reference): ``distance_postprocessing`` (:7-59) and
``boundary_postprocessing`` (:62-90) with the same signatures, argument order (seed threshold before cell threshold;
callers pass keywords) and return type (``np.uint16`` array of shape (H, W)).  Inputs may be numpy arrays — (H, W, 1)
as every reference caller passes (infer.py:358-359) or (H, W) — or CUDA tensors that are already on the device
(the inference driver keeps predictions in HBM).  There is no scipy/scikit-image fallback: without libmseg_hip the
call raises.
"""
import numpy as np
import torch

from .. import _lib

_ws_cache = {}
_slot = [0]


class workspace_slot:
    """``with workspace_slot(i):`` the post-processing calls inside use workspace i (default 0) — for callers that keep
    several calls in flight on different streams; a slot must not be reused before its previous call has finished"""

    def __init__(self, slot):
        self.slot = int(slot)

    def __enter__(self):
        self.prev, _slot[0] = _slot[0], self.slot

    def __exit__(self, *exc):
        _slot[0] = self.prev
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _device(t=None):
    if t is not None and isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd post-processing needs the MI355X HIP path (no CPU fallback); "
                           "oracle/postproc_ref.py is the CPU checker used by the tests")
    return torch.device("cuda", torch.cuda.current_device())


def _workspace(dev, H, W):
    lib = _lib.load()
    need = lib.mseg_postproc_workspace_bytes(H, W)
    if need == 0:
        raise RuntimeError(f"unsupported frame size {H}x{W}")
    # one workspace per (device, slot): calls that may be in flight together on different streams (frames of a stack,
    # infer_stack) name different slots and so never share scratch memory
    key = (str(dev), _slot[0])
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    return ws, need


def _as_device_2d(a, dev):
    """-> (contiguous float32 CUDA tensor (H, W) or (H, W, C), original ndim)"""
    if isinstance(a, torch.Tensor):
        t = a.detach().to(device=dev, dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    return t.contiguous(), t.dim()


def distance_postprocessing_device(border, cell, th_seed, th_cell, col_major_ids=True):
    """Device-resident variant: border / cell float32 CUDA tensors (H, W) -> (labels uint16 CUDA tensor (H, W),
    n_instances int32 tensor, status int32 tensor [bit0: exact serial flood was used, bit1: age-0 tie detected])."""
    lib = _lib.load()
    H, W = cell.shape
    dev = cell.device
    ws, need = _workspace(dev, H, W)
    # int16 storage for uint16 labels (torch has no uint16 arithmetic; reinterpret on the host side)
    labels = torch.empty((H, W), dtype=torch.int16, device=dev)
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(lib.mseg_distance_postprocess(border.data_ptr(), cell.data_ptr(), H, W, float(th_cell), float(th_seed),
                                             1 if col_major_ids else 0, labels.data_ptr(), info[0:].data_ptr(),
                                             info[1:].data_ptr(), ws.data_ptr(), need, _stream()),
               "distance_postprocess")
    return labels, info[0], info[1]


def distance_postprocessing_sweep_device(border, cell, ths, col_major_ids=True):
    """Threshold sweep of the evaluation (reference: src/evaluation/eval.py:127-131,397-409 calls distance_postprocessing
    once per (th_cell, th_seed) pair on the same prediction).  border / cell: float32 CUDA tensors (H, W); ths: sequence
    of (th_cell, th_seed).  Returns (labels int16-view-of-uint16 CUDA tensor (n, H, W), n_instances int32 (n,),
    status int32 (n,)); labels[i] equals distance_postprocessing_device(border, cell, th_seed_i, th_cell_i)."""
    import ctypes as C
    lib = _lib.load()
    H, W = cell.shape
    dev = cell.device
    nth = len(ths)
    if nth == 0:
        raise RuntimeError("empty threshold list")
    ws, need = _workspace(dev, H, W)
    labels = torch.empty((nth, H, W), dtype=torch.int16, device=dev)
    info = torch.zeros((2, nth), dtype=torch.int32, device=dev)
    tc = (C.c_float * nth)(*[float(t[0]) for t in ths])
    ts = (C.c_float * nth)(*[float(t[1]) for t in ths])
    _lib.check(lib.mseg_distance_postprocess_sweep(border.data_ptr(), cell.data_ptr(), H, W, tc, ts, nth,
                                                   1 if col_major_ids else 0, labels.data_ptr(), info[0].data_ptr(),
                                                   info[1].data_ptr(), ws.data_ptr(), need, _stream()),
               "distance_postprocess_sweep")
    return labels, info[0], info[1]


def distance_postprocessing(border_prediction, cell_prediction, th_seed, th_cell):
    """ Post-processing for distance label (cell + neighbor) prediction (reference postprocessing.py:7).

    :param border_prediction: Neighbor distance prediction, (H, W, 1) or (H, W) float32.
    :param cell_prediction: Cell distance prediction, same shape.
    :param th_seed: Threshold for seed/marker extraction.
    :param th_cell: Threshold for cell size.
    :return: Instance segmentation mask (np.uint16, (H, W)).
    """
    dev = _device(cell_prediction)
    cell, nd = _as_device_2d(cell_prediction, dev)
    border, _ = _as_device_2d(border_prediction, dev)
    # (H, W, 1) inputs get column-major instance ids, 2-D inputs raster ids — exactly what skimage.measure.label does
    # to the reference for these two ranks
    col_major = nd == 3
    if nd == 3:
        if cell.shape[2] != 1:
            raise RuntimeError("expected (H, W, 1) predictions")
        cell, border = cell[..., 0].contiguous(), border[..., 0].contiguous()
    labels, _, _ = distance_postprocessing_device(border, cell, th_seed, th_cell, col_major)
    return labels.cpu().numpy().view(np.uint16)


def boundary_postprocessing_device(probs_hwc):
    lib = _lib.load()
    H, W, Cc = probs_hwc.shape
    if Cc != 3:
        raise RuntimeError("expected (H, W, 3) softmax probabilities")
    dev = probs_hwc.device
    ws, need = _workspace(dev, H, W)
    labels = torch.empty((H, W), dtype=torch.int16, device=dev)
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(lib.mseg_boundary_postprocess(probs_hwc.data_ptr(), H, W, labels.data_ptr(), info[0:].data_ptr(),
                                             info[1:].data_ptr(), ws.data_ptr(), need, _stream()),
               "boundary_postprocess")
    return labels, info[0], info[1]


def boundary_postprocessing_batch_device(probs_list, first_slot=0):
    """The boundary method's post-processing of up to 8 frames at once: per frame the stages before the flood on its own
    workspace (slots first_slot .. first_slot + B - 1), ONE launch for the floods of all frames (one workgroup per frame:
    the flood is a single wavefront busy for ~45 ms per 2048 x 2048 frame), then the last stage per frame.  Returns a list
    of (labels, n_instances, status) device tensors, bit-identical to boundary_postprocessing_device frame by frame."""
    import ctypes as C
    lib = _lib.load()
    B = len(probs_list)
    if not 1 <= B <= 8:
        raise RuntimeError("1 to 8 frames per batch")
    H, W, Cc = probs_list[0].shape
    dev = probs_list[0].device
    wss, need = [], 0
    for i, pr in enumerate(probs_list):
        if tuple(pr.shape) != (H, W, 3):
            raise RuntimeError("expected (H, W, 3) softmax probabilities of one size")
        with workspace_slot(first_slot + i):
            ws, need = _workspace(dev, H, W)
        wss.append(ws)
        _lib.check(lib.mseg_boundary_postprocess_pre(pr.data_ptr(), H, W, ws.data_ptr(), need, _stream()),
                   "boundary_postprocess_pre")
    ptrs = (C.c_void_p * B)(*[ws.data_ptr() for ws in wss])
    _lib.check(lib.mseg_boundary_flood_batch(ptrs, B, H, W, _stream()), "boundary_flood_batch")
    out = []
    for ws in wss:
        labels = torch.empty((H, W), dtype=torch.int16, device=dev)
        info = torch.zeros(2, dtype=torch.int32, device=dev)
        _lib.check(lib.mseg_boundary_postprocess_post(H, W, labels.data_ptr(), info[0:].data_ptr(), info[1:].data_ptr(),
                                                      ws.data_ptr(), need, _stream()), "boundary_postprocess_post")
        out.append((labels, info[0], info[1]))
    return out


def boundary_postprocessing(prediction):
    """ Post-processing for boundary label prediction (reference postprocessing.py:62).

    :param prediction: softmax probabilities (H, W, 3): background, cell interior, boundary.
    :return: Instance segmentation mask (np.uint16, (H, W)).
    """
    dev = _device(prediction)
    probs, _ = _as_device_2d(prediction, dev)
    labels, _, _ = boundary_postprocessing_device(probs)
    return labels.cpu().numpy().view(np.uint16)
