"""Shared helpers for the parity tests (fixtures -> tensors, variant table)."""
import pathlib

import numpy as np
import torch

GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"

# name -> (unet_type, act, norm, filters, ch_out, label_type)   [same table as tools/gen_golden_unet.py]
VARIANTS = {
    "DU_bn_relu_8_16": ("DU", "relu", "bn", (8, 16), 1, "distance"),
    "DU_bn_mish_8_16": ("DU", "mish", "bn", (8, 16), 1, "distance"),
    "U_gn_relu_8_16": ("U", "relu", "gn", (8, 16), 3, "boundary"),
    "U_in_elu_8_16": ("U", "elu", "in", (8, 16), 3, "boundary"),
    "DU_gn_leakyrelu_8_32": ("DU", "leakyrelu", "gn", (8, 32), 1, "distance"),
    "U_bn_mish_8_32": ("U", "mish", "bn", (8, 32), 3, "boundary"),
    "DU_bn_leakyrelu_max_8_32": ("DU", "leakyrelu", "bn", (8, 32), 1, "distance", "max"),
}


def variant(name):
    """-> (unet_type, act, norm, filters, ch_out, label_type, pool_method)"""
    v = VARIANTS[name]
    return v if len(v) == 7 else v + ("conv",)


def load_npz(name):
    with np.load(GOLDEN / name) as f:
        return {k: f[k] for k in f.files}


def state_from(fx, prefix="w/"):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in fx.items() if k.startswith(prefix)}


def rel_err(a, b, floor=0.0):
    """max |a-b| / max(|b|) — the 'relative fp32' measure used for the 1e-4 parity bar (BASELINE.json north_star).
    ``floor`` bounds the denominator from below (gradients that are analytically zero, e.g. a conv bias in front of
    InstanceNorm, are pure rounding noise in both implementations)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    denom = max(b.abs().max().item(), floor)
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def grad_floor(fx, prefix="g/"):
    """1e-3 x the largest gradient magnitude of the fixture: the scale below which a gradient counts as zero."""
    return 1e-3 * max(float(np.abs(v).max()) for k, v in fx.items() if k.startswith(prefix))
