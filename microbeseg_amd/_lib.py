"""ctypes binding of libmseg_hip.so (C ABI declared in include/mseg_hip.h).

The library is the product: there is no PyTorch / CPU fallback behind these calls.  If the shared object is
missing or does not load, importing the compute path raises (``MsegLibraryError``) instead of silently degrading.
"""
import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmseg_hip.so"

ACT = {"none": 0, "relu": 1, "leakyrelu": 2, "elu": 3, "mish": 4}
NORM = {"bn": 0, "gn": 1, "in": 2}
MODE_CONV, MODE_TCONV = 0, 1
EPI_PLAIN, EPI_SCATTER2X2 = 0, 1
MORDER_LINEAR, MORDER_PARITY = 0, 1
ST_F32, ST_BF16 = 0, 1      # tensor storage in HBM (MsegSrc.dtype, MsegIgemm.dst_dtype, `st` arguments)


class MsegLibraryError(RuntimeError):
    pass


class MsegSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int32), ("act", C.c_int32), ("ss", C.c_int32), ("dtype", C.c_int32)]


class MsegIgemm(C.Structure):
    _fields_ = [("src", MsegSrc * 2), ("w", C.c_void_p), ("bias", C.c_void_p), ("dst0", C.c_void_p),
                ("dst1", C.c_void_p),
                ("nsrc", C.c_int32), ("Cin", C.c_int32), ("Kpad", C.c_int32), ("Npad", C.c_int32),
                ("NB", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("mode", C.c_int32), ("morder", C.c_int32),
                ("Ngemm", C.c_int32), ("epi", C.c_int32), ("split", C.c_int32), ("ld0", C.c_int32),
                ("ld1", C.c_int32), ("acc0", C.c_int32), ("acc1", C.c_int32), ("Cq", C.c_int32),
                ("precision", C.c_int32), ("dst_dtype", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("stats", C.c_void_p), ("stats_act", C.c_int32), ("reserved0", C.c_int32)]


class MsegWgrad(C.Structure):
    _fields_ = [("P", MsegSrc), ("Q", MsegSrc * 2), ("ws", C.c_void_p), ("dst", C.c_void_p),
                ("nq", C.c_int32), ("Nch", C.c_int32), ("Nch_store", C.c_int32),
                ("NB", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32), ("Hq", C.c_int32), ("Wq", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("splits", C.c_int32), ("phase", C.c_int32), ("precision", C.c_int32), ("reserved", C.c_int32)]


class MsegKernelInfo(C.Structure):
    _fields_ = [("name", C.c_char * 120), ("precision", C.c_int32), ("launches", C.c_int32), ("grid", C.c_uint32),
                ("block", C.c_uint32), ("workspace", C.c_size_t), ("stats_rows", C.c_int32), ("reserved0", C.c_int32)]


class MsegPackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst16", C.c_void_p),
                ("T", C.c_int32), ("R", C.c_int32), ("Rpad", C.c_int32), ("C", C.c_int32), ("Cpad", C.c_int32),
                ("st", C.c_int32), ("sr", C.c_int32), ("sc", C.c_int32),
                ("first_block", C.c_uint32), ("reserved", C.c_uint32)]


class MsegRangerJob(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("slow", C.c_void_p),
                ("n", C.c_uint64), ("rows", C.c_int32), ("step_lr", C.c_float), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_D = C.c_double
_SZ = C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/mseg_hip.h
SIGNATURES = {
    "mseg_igemm": (_I, [C.POINTER(MsegIgemm), _P]),
    "mseg_igemm_workspace_bytes": (_SZ, [C.POINTER(MsegIgemm)]),
    "mseg_igemm_query": (_I, [C.POINTER(MsegIgemm), C.POINTER(MsegKernelInfo)]),
    "mseg_last_kernel": (C.c_char_p, []),
    "mseg_igemm_set_persistent": (_I, [_I]),
    "mseg_igemm_set_wide_tiles": (_I, [_I]),
    "mseg_igemm_set_p8": (_I, [_I]),
    "mseg_f32_to_bf16": (_I, [_P, _P, _SZ, _P]),
    "mseg_first_conv_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "mseg_frame_minmax": (_I, [_P, _I, _SZ, _P, _P]),
    "mseg_first_conv_fwd_raw": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    "mseg_frame_normalize": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mseg_first_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "mseg_first_wgrad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mseg_wgrad_workspace_bytes": (_SZ, [C.POINTER(MsegWgrad)]),
    "mseg_wgrad": (_I, [C.POINTER(MsegWgrad), _P]),
    "mseg_wgrad_query": (_I, [C.POINTER(MsegWgrad), C.POINTER(MsegKernelInfo)]),
    "mseg_pack_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mseg_pack_job_blocks": (C.c_uint, [_I, _I, _I]),
    "mseg_pack_weights_multi": (_I, [_P, _I, C.c_uint, _P]),
    "mseg_norm_workspace_bytes": (_SZ, [_I, _I, _I]),
    "mseg_norm_set_tails": (_I, [_I]),
    "mseg_norm_set_finish": (_I, [_I]),
    "mseg_norm_stats_from_conv": (_I, [_P, _I, _I, C.c_longlong, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "mseg_norm_stats": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mseg_activation": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "mseg_bn_eval_coeffs": (_I, [_P, _P, _P, _P, _F, _I, _P, _P, _P]),
    "mseg_norm_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mseg_maxpool2x2_fwd": (_I, [C.POINTER(MsegSrc), _I, _I, _I, _P, _P]),
    "mseg_maxpool2x2_bwd": (_I, [C.POINTER(MsegSrc), _I, _I, _I, _P, _P, _I, _P]),
    "mseg_head_fwd": (_I, [C.POINTER(MsegSrc), _I, _I, _P, _P, _I, _P, _P]),
    "mseg_head_bwd_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "mseg_head_bwd": (_I, [C.POINTER(MsegSrc), _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P]),
    "mseg_softmax3_hwc": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mseg_loss_workspace_bytes": (_SZ, [_SZ]),
    "mseg_regression_loss": (_I, [_P, _P, _SZ, _I, _P, _P, _P]),
    "mseg_regression_loss_bwd": (_I, [_P, _P, _SZ, _I, _P, _P, _P]),
    "mseg_ce_dice_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "mseg_ce_dice_bwd": (_I, [_P, _P, _I, _I, _I, _P, C.c_double, C.c_double, _P, _P, _P]),
    "mseg_adam_amsgrad_step": (_I, [_P, _P, _P, _P, _P, _SZ, _D, _D, _D, _D, _I, _P]),
    "mseg_adam_amsgrad_step_dev": (_I, [_P, _P, _P, _P, _P, _SZ, _P, _D, _D, _D, _P]),
    "mseg_ranger_step": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _D, _D, _D, _D, _I, _I, _I, _D, _P]),
    "mseg_ranger_step_multi": (_I, [_P, _I, _D, _D, _D, _D, _P]),
    "mseg_postproc_workspace_bytes": (_SZ, [_I, _I]),
    "mseg_postproc_tuning": (_I, [_I, _I, _I]),
    "mseg_postproc_set_const_stream": (_I, [_I]),
    "mseg_distance_postprocess": (_I, [_P, _P, _I, _I, _F, _F, _I, _P, _P, _P, _P, _SZ, _P]),
    "mseg_boundary_postprocess": (_I, [_P, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "mseg_boundary_postprocess_pre": (_I, [_P, _I, _I, _P, _SZ, _P]),
    "mseg_boundary_flood_batch": (_I, [C.POINTER(C.c_void_p), _I, _I, _I, _P]),
    "mseg_boundary_postprocess_post": (_I, [_I, _I, _P, _P, _P, _P, _SZ, _P]),
    "mseg_distance_postprocess_sweep": (_I, [_P, _P, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _I, _P, _P,
                                             _P, _P, _SZ, _P]),
    "mseg_aug_u16_to_f32": (_I, [_P, _P, _SZ, _P]),
    "mseg_aug_flip": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "mseg_aug_affine": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "mseg_aug_blur": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "mseg_aug_stats": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "mseg_aug_contrast_params": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "mseg_aug_contrast": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "mseg_aug_clahe_workspace_bytes": (_SZ, [_I]),
    "mseg_aug_clahe": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mseg_aug_noise_normalize": (_I, [_P, _P, _I, _I, _I, _P, _P, C.c_uint32, _F, _F, _P]),
    "mseg_label_boundary": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "mseg_label_distance_workspace_bytes": (_SZ, [_I, _I, _I]),
    "mseg_label_distance": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "mseg_label_bottom_hat": (_I, [_P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "mseg_label_cell_distance": (_I, [_P, _I, _I, _I, _I, _F, _P, _P, _SZ, _P]),
    "mseg_label_j4": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "mseg_label_major_axis_workspace_bytes": (_SZ, [_I]),
    "mseg_label_max_major_axis": (_I, [_P, _I, _I, _I, _P, _P, _SZ, _P]),
    "mseg_eval_workspace_bytes": (_SZ, [_I, _I]),
    "mseg_eval_relabel": (_I, [_P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "mseg_eval_pair_counts": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "mseg_polygons_find": (_I, [_P, _I, _I, _P, _P, _P, _I, _P, _P]),
    "mseg_polygons_trace": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "mseg_version": (_I, []),
    "mseg_strerror": (C.c_char_p, [_I]),
    "mseg_last_hip_error": (_I, []),
}

_lib = None


def load():
    """Load libmseg_hip.so (once) and attach prototypes.  Raises MsegLibraryError if it is absent/unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the copy that is resident before this library is
    # opened (same soname: the first one loaded wins, and a second runtime would not see torch's device/streams).
    import torch  # noqa: F401
    path = os.environ.get("MSEG_HIP_LIB", str(LIB_PATH))
    if not os.path.exists(path):
        raise MsegLibraryError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the compute path.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise MsegLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MsegLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        lib = load()
        msg = lib.mseg_strerror(code).decode()
        hip = lib.mseg_last_hip_error()
        # RuntimeError is load-bearing in the reference: OOM ladder train.py:276-297, zero mask infer.py:354-356
        raise RuntimeError(f"libmseg_hip {what}: {msg} (code {code}, hipError {hip})")
