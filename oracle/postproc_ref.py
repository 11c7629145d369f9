"""ORACLE (test infrastructure only): ctypes front-end of oracle/postproc_ref.c — the CPU restatement of
src/inference/postprocessing.py (reference) used as the checker in tests/, smoke() and bench.py's cpu_baseline."""
import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SO = _HERE / "_build" / "liboracle_postproc.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            subprocess.run(["bash", str(_HERE / "build.sh")], check=True)
        _lib = C.CDLL(str(_SO))
        _lib.ref_label8.restype = C.c_int
        _lib.ref_distance_postprocessing.restype = C.c_int
        _lib.ref_distance_postprocessing.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                                     C.c_int, C.c_void_p, C.c_void_p]
        _lib.ref_boundary_postprocessing.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.squeeze(a), dtype=np.float32)


def gaussian05(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().ref_gaussian05(x.ctypes.data_as(C.c_void_p), C.c_int(x.shape[0]), C.c_int(x.shape[1]),
                         out.ctypes.data_as(C.c_void_p))
    return out


def label8(binary):
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    out = np.empty(b.shape, np.int32)
    n = lib().ref_label8(b.ctypes.data_as(C.c_void_p), C.c_int(b.shape[0]), C.c_int(b.shape[1]),
                         out.ctypes.data_as(C.c_void_p))
    return out, n


def watershed(image, markers, mask):
    img = np.ascontiguousarray(image, dtype=np.float64)
    mk = np.ascontiguousarray(markers, dtype=np.int32)
    ms = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(img.shape, np.int32)
    lib().ref_watershed(img.ctypes.data_as(C.c_void_p), mk.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p),
                        C.c_int(img.shape[0]), C.c_int(img.shape[1]), out.ctypes.data_as(C.c_void_p))
    return out


def distance_postprocessing(border_prediction, cell_prediction, th_seed, th_cell, return_margin=False):
    """Same argument order as the reference (postprocessing.py:7).  (H,W,1) inputs -> column-major instance ids,
    (H,W) inputs -> raster ids."""
    col_major = int(np.ndim(cell_prediction) == 3)
    b, c = _f32(border_prediction), _f32(cell_prediction)
    out = np.empty(c.shape, np.uint16)
    margin = C.c_float(0)
    lib().ref_distance_postprocessing(b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), c.shape[0],
                                      c.shape[1], float(th_seed), float(th_cell), col_major,
                                      out.ctypes.data_as(C.c_void_p), C.cast(C.byref(margin), C.c_void_p))
    return (out, margin.value) if return_margin else out


def boundary_postprocessing(prediction):
    p = np.ascontiguousarray(prediction, dtype=np.float32)
    out = np.empty(p.shape[:2], np.uint16)
    lib().ref_boundary_postprocessing(p.ctypes.data_as(C.c_void_p), C.c_int(p.shape[0]), C.c_int(p.shape[1]),
                                      out.ctypes.data_as(C.c_void_p))
    return out
