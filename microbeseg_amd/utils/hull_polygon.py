"""Instance label image -> per-instance contour polygons, on the MI355X.

Drop-in for ``src/utils/hull_polygon.py`` of the reference: ``get_indices_pandas(data, background_id=0)`` (:8-41) and
``cv2_countour(mask_idx)`` (:44-89) keep their names, arguments and return types — the reference's upload loop
(``src/inference/infer.py``:274-287) works unchanged on top of them — but that loop calls OpenCV once per instance from
Python; the batch entry point ``label_polygons(labels)`` traces every instance of a frame in two launches
(``csrc/polygons.hip``) and is what ``InferWorker.polygon_rois`` uses.

A polygon is a ``(2, N)`` integer array ``[rows; cols]`` in image coordinates, listing the pixels of the instance's outer
border in the order ``cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_NONE)`` lists them.  Holes are ignored, exactly as the
reference keeps only the contour that covers the others.  Parity with OpenCV is UNPINNED (OpenCV / shapely are installed
in neither interpreter of the build container): the kernel follows ``oracle/contour_ref.py``, a restatement of the
published border-following algorithm, bit for bit (tests/test_polygons.py).

Deviation, documented: an id with several 8-connected components (cannot come out of the watershed; a wrapped uint16 id
could) yields one polygon per component in raster order of their first pixels; the reference's shapely logic returns an
implementation-defined subset there.
"""
import numpy as np
import torch

from .. import _lib


def _device(t=None):
    if isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("microbeseg_amd polygon tracing needs the MI355X HIP path (no CPU fallback); "
                           "oracle/contour_ref.py is the CPU checker used by the tests")
    return torch.device("cuda", torch.cuda.current_device())


def label_polygons_device(labels):
    """labels: int16 / uint16-as-int16 CUDA tensor (H, W), 0 = background.
    Returns (ids int64 [P], first_px int64 [P], offsets int64 [P + 1], points int32 CUDA tensor [total, 2] of (row, col)),
    polygons ordered by id, then by the raster index of their first pixel."""
    lib = _lib.load()
    if labels.dim() != 2 or labels.dtype != torch.int16 or not labels.is_cuda:
        raise RuntimeError("expected a 2-D int16 (uint16 bit pattern) CUDA tensor")
    labels = labels.contiguous()
    H, W = labels.shape
    dev = labels.device
    stream = torch.cuda.current_stream().cuda_stream
    capacity = max(4096, H * W // 64)
    while True:
        cand = torch.empty((3, capacity), dtype=torch.int32, device=dev)
        n_cand = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mseg_polygons_find(labels.data_ptr(), H, W, cand[0].data_ptr(), cand[1].data_ptr(),
                                          cand[2].data_ptr(), capacity, n_cand.data_ptr(), stream), "polygons_find")
        n = int(n_cand.item())                       # a few thousand records: the one host round trip of the count pass
        if n <= capacity:
            break
        capacity = n
    px, ids, length = (t.to(torch.int64) for t in cand[:, :n].cpu())
    keep = length > 0
    px, ids, length = px[keep], ids[keep] & 0xFFFF, length[keep]
    order = torch.argsort(ids * (H * W) + px)        # by id, then first pixel
    px, ids, length = px[order], ids[order], length[order]
    offsets = torch.zeros(len(px) + 1, dtype=torch.int64)
    torch.cumsum(length, 0, out=offsets[1:])
    total = int(offsets[-1])
    points = torch.empty((max(total, 1), 2), dtype=torch.int32, device=dev)
    if len(px):
        start = px.to(torch.int32).to(dev)
        off_dev = offsets.to(dev)
        _lib.check(lib.mseg_polygons_trace(labels.data_ptr(), H, W, start.data_ptr(), off_dev.data_ptr(), len(px),
                                           points.data_ptr(), stream), "polygons_trace")
    return ids, px, offsets, points[:total]


def label_polygons(labels, background_id=0):
    """``{instance id: [(2, N) array [rows; cols], ...]}`` for a label image (numpy uint16 / integer array or CUDA tensor):
    the batch form of ``get_indices_pandas`` + ``cv2_countour`` per instance."""
    if background_id != 0:
        raise RuntimeError("only background_id = 0 is supported (the reference never passes another value)")
    dev = _device(labels)
    if isinstance(labels, torch.Tensor):
        t = labels if labels.dtype == torch.int16 else labels.to(torch.int32).to(torch.int16)
    else:
        a = np.ascontiguousarray(np.squeeze(labels))
        if a.ndim != 2:
            raise RuntimeError("expected a 2-D label image")
        if a.min(initial=0) < 0 or a.max(initial=0) > 65535:
            raise RuntimeError("label ids must fit uint16 (the reference's masks are uint16)")
        t = torch.from_numpy(a.astype(np.uint16).view(np.int16))
    t = t.to(dev)
    ids, _, offsets, points = label_polygons_device(t)
    pts = points.cpu().numpy().astype(np.int64)
    out = {}
    for k, i in enumerate(ids.tolist()):
        out.setdefault(i, []).append(pts[offsets[k]:offsets[k + 1]].T.reshape(2, -1))
    return out


def get_indices_pandas(data, background_id=0):
    """ Positions of every mask id within the array: pandas Series indexed by mask id whose values are the
    ``np.unravel_index`` tuples of that id's pixels (raster order) — the reference's return format. """
    import pandas as pd
    data = np.asarray(data)
    flat = data.ravel()
    where = np.flatnonzero(flat != background_id)
    ids = flat[where]
    order = np.argsort(ids, kind="stable")
    where, ids = where[order], ids[order]
    cuts = np.flatnonzero(np.diff(ids)) + 1
    groups = np.split(where, cuts) if len(where) else []
    series = pd.Series([np.unravel_index(g, data.shape) for g in groups],
                       index=pd.Index([ids_g[0] for ids_g in np.split(ids, cuts)] if len(where) else [], name="mask_id"),
                       dtype=object)
    return series


def cv2_countour(mask_idx):
    """ Contour polygon(s) of ONE instance given its pixel positions ``(rows, cols)`` (the values of
    ``get_indices_pandas``).  Returns a list of (2, N) arrays like the reference; traced on the device. """
    mask_idx = np.array(mask_idx)
    if mask_idx.ndim != 2 or mask_idx.shape[0] < 2:
        raise AssertionError("expected (rows, cols) index arrays")
    rows, cols = mask_idx[-2], mask_idx[-1]              # (z, rows, cols) of an (H, W, 1)-style index is accepted too
    lo = np.array([rows.min(), cols.min()])
    box = np.zeros((rows.max() - lo[0] + 3, cols.max() - lo[1] + 3), dtype=np.uint16)
    box[rows - lo[0] + 1, cols - lo[1] + 1] = 1
    return [p + lo.reshape(2, 1) - 1 for p in label_polygons(box).get(1, [])]


def points_string(polygon):
    """ (2, N) [rows; cols] -> ``"x,y x,y ... "``, the points attribute of an OMERO polygon ROI (infer.py:283-286). """
    return "".join("{},{} ".format(c, r) for r, c in zip(polygon[0].tolist(), polygon[1].tolist()))
