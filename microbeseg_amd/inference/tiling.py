"""Sliding-window (tiled) inference — an EXTENSION: the reference stores ``sliding_window`` and never reads it
(src/inference/infer.py:60,76; GUI checkbox hidden), and tells the user "Image too big to pad. Use sliding windows"
for frames beyond 8192 px (src/utils/utils.py:154-155) without offering any.

A frame is cut into tiles of ``tile`` x ``tile`` output pixels, each computed from an input window that extends ``halo``
pixels beyond the tile on every side that is not a frame border.  For a BatchNorm model in eval mode every layer is a
local operator (per-channel affine after the activation), so with a halo wider than the receptive-field radius of the
network — 107 px for the 5-level nets (per level two 3x3 convs and a stride-2 3x3 conv down, two 3x3 convs up) — the
stitched prediction EQUALS whole-frame inference: frame borders coincide with tile-window borders (same zero padding),
cut edges are further from every kept pixel than the network can see.  HALO = 128 also keeps every window aligned to the
16-px grid of the deepest level.  GroupNorm / InstanceNorm statistics are per whole frame: tiling such a
model would change its output, so it is refused.
"""
import torch
import torch.nn as nn

HALO = 128          # > receptive-field radius (107) of the deepest supported net, multiple of 2 ** (levels - 1)


def pad_to_grid(img, pad_val, grid=16):
    """top / left padding (the reference pads top and left, utils.py:161) up to the next multiple of ``grid``"""
    import numpy as np
    pads = [(-img.shape[0]) % grid, (-img.shape[1]) % grid]
    if pads[0] or pads[1]:
        img = np.pad(img, ((pads[0], 0), (pads[1], 0)), mode='constant', constant_values=pad_val)
    return img, pads


def check_tileable(net):
    bare = net.module if hasattr(net, "module") else net
    for m in bare.modules():
        if isinstance(m, (nn.GroupNorm, nn.InstanceNorm2d)):
            raise RuntimeError("sliding-window inference needs a BatchNorm model: GroupNorm / InstanceNorm statistics "
                               "span the whole frame, tiling would change the prediction")
    if bare.training:
        raise RuntimeError("sliding-window inference needs eval mode (running BatchNorm statistics)")


def tiled_forward(net, x, tile=2048, halo=HALO):
    """x: (N, 1, H, W) CUDA tensor with H, W multiples of 16.  Returns what ``net(x)`` returns (tensor or tuple)."""
    check_tileable(net)
    if tile % 16 or halo % 16 or tile <= 0:
        raise ValueError("tile and halo must be positive multiples of 16")
    N, _, H, W = x.shape
    outs = None
    for y0 in range(0, H, tile):
        y1 = min(y0 + tile, H)
        ys, ye = max(y0 - halo, 0), min(y1 + halo, H)
        for x0 in range(0, W, tile):
            x1 = min(x0 + tile, W)
            xs, xe = max(x0 - halo, 0), min(x1 + halo, W)
            pred = net(x[:, :, ys:ye, xs:xe].contiguous())
            single = not isinstance(pred, tuple)
            pred = (pred,) if single else pred
            if outs is None:
                outs = tuple(torch.empty((N, p.shape[1], H, W), dtype=p.dtype, device=p.device) for p in pred)
            for o, p in zip(outs, pred):
                o[:, :, y0:y1, x0:x1] = p[:, :, y0 - ys:y1 - ys, x0 - xs:x1 - xs]
            del pred
    return outs[0] if single else outs
