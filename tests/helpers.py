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


def bf16_rule(kind, xs, ws):
    """which of (forward, data gradient, weight gradient) of a layer run with bf16 operands in bf16 mode: the LIBRARY's
    answer for the three launches of that layer (engine.igemm_query / wgrad_query = mseg_igemm_query / mseg_wgrad_query;
    concat inputs are asked as one source, which the dispatch treats alike; first layer: VALU kernels, fp32)"""
    from microbeseg_amd import engine as E
    from microbeseg_amd._lib import ST_F32
    N, cin, H, W = xs

    def src(c):
        return E._dummy_src(c, ST_F32)

    def ig(c_in, ngemm, hi, wi, ho, wo, k, stride, pad, mode, **kw):
        npad, kpad = E._dummy_pack(ngemm, c_in)
        return E.igemm_query([src(c_in)], kpad, npad, N, hi, wi, ho, wo, k, k, stride, pad, mode, ngemm, precision="bf16",
                             **kw) is not None

    def wg(pc, qc, hp, wp, hq, wq, k, stride, pad):
        return E.wgrad_query(src(pc), [src(qc)], N, hp, wp, hq, wq, k, k, stride, pad, precision="bf16") is not None
    if kind == "up":                                      # ConvTranspose2d 2x2 stride 2, weight (Cin, Cout, 2, 2)
        co = ws[1]
        return (ig(cin, 4 * co, H, W, H, W, 1, 1, 0, E.MODE_CONV, ld0=co, epi=E.EPI_SCATTER2X2, Cq=co),
                ig(co, cin, 2 * H, 2 * W, H, W, 2, 2, 0, E.MODE_CONV, ld0=cin, bias=False),
                wg(cin, co, H, W, 2 * H, 2 * W, 2, 2, 0))
    cout = ws[0]
    if cin <= 4:
        return (False, False, False)
    if kind == "pool":                                    # 3x3 stride 2
        ho, wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        return (ig(cin, cout, H, W, ho, wo, 3, 2, 1, E.MODE_CONV, ld0=cout),
                ig(cout, cin, ho, wo, H, W, 3, 2, 1, E.MODE_TCONV, ld0=cin, morder=E.MORDER_PARITY, bias=False),
                wg(cout, cin, ho, wo, H, W, 3, 2, 1))
    return (ig(cin, cout, H, W, H, W, 3, 1, 1, E.MODE_CONV, ld0=cout),
            ig(cout, cin, H, W, H, W, 3, 1, 1, E.MODE_TCONV, ld0=cin, bias=False),
            wg(cout, cin, H, W, H, W, 3, 1, 1))


class NodeTrace:
    """records every engine Node of a forward pass (execution order) by wrapping engine.norm_stats"""

    def __enter__(self):
        from microbeseg_amd import engine
        self.engine, self.nodes, self.orig = engine, [], engine.norm_stats

        def wrapped(node, *a, **k):
            self.nodes.append(node)
            return self.orig(node, *a, **k)
        engine.norm_stats = wrapped
        return self

    def __exit__(self, *exc):
        self.engine.norm_stats = self.orig
        return False

    def relu_masks(self):
        """the HIP path's ReLU decisions, NCHW bool on the CPU, one per activated layer in execution order"""
        return [(n.z > 0).permute(0, 3, 1, 2).contiguous().cpu() for n in self.nodes if n.act == 1]   # _lib.ACT["relu"]


def check_relu_flips(masks, trace, max_frac=1e-4, tol=1e-4):
    """ReLU decisions that differ between the HIP path and the fp32 oracle: each must sit at a pre-activation within
    the forward tolerance of 0 (|z_ref| <= tol * max|z| of its layer), and there must be few of them."""
    total = flips = 0
    for m, z in zip(masks, trace):
        d = m != (z > 0)
        n = int(d.sum())
        total += d.numel()
        flips += n
        if n:
            assert z[d].abs().max().item() <= tol * z.abs().max().item(), "a ReLU decision flipped away from z = 0"
            assert n <= max(4, max_frac * d.numel()), f"{n} flipped ReLU decisions in a layer of {d.numel()}"
    return flips, total
