#!/usr/bin/env python3
"""Generate tests/golden/unet_*.npz from the REAL reference (imports /root/reference by path; build container only).

Run:  cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_unet.py
Nothing from the reference is copied: the fixtures hold inputs, weights drawn from numpy PCG64, and the numbers the
reference modules (src/utils/unets.py, src/training/losses.py, src/training/ranger2020.py) produce for them.
"""
import contextlib
import io
import pathlib
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from src.utils.unets import build_unet  # noqa: E402
from src.training.losses import get_loss  # noqa: E402
from src.training.ranger2020 import Ranger  # noqa: E402

OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
torch.set_num_threads(4)

VARIANTS = [
    # name, unet_type, act, norm, filters, ch_out, label_type
    ("DU_bn_relu_8_16", "DU", "relu", "bn", (8, 16), 1, "distance"),
    ("DU_bn_mish_8_16", "DU", "mish", "bn", (8, 16), 1, "distance"),
    ("U_gn_relu_8_16", "U", "relu", "gn", (8, 16), 3, "boundary"),
    ("U_in_elu_8_16", "U", "elu", "in", (8, 16), 3, "boundary"),
    ("DU_gn_leakyrelu_8_32", "DU", "leakyrelu", "gn", (8, 32), 1, "distance"),
    ("U_bn_mish_8_32", "U", "mish", "bn", (8, 32), 3, "boundary"),
    ("DU_bn_leakyrelu_max_8_32", "DU", "leakyrelu", "bn", (8, 32), 1, "distance", "max"),
]


def seeded_state(net, rng):
    """Replace every tensor of the state dict by PCG64 draws (so affine params / running stats are non-trivial)."""
    sd = net.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            new[k] = torch.tensor(3, dtype=torch.int64)
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy(rng.uniform(0.5, 1.5, v.shape).astype(np.float32))
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy(rng.normal(0, 0.2, v.shape).astype(np.float32))
        elif v.dim() == 1 and (".conv.2." in k or ".conv.5." in k or ".conv_pool.2." in k or ".norm." in k):
            if k.endswith("weight"):
                new[k] = torch.from_numpy(rng.uniform(0.7, 1.3, v.shape).astype(np.float32))
            else:
                new[k] = torch.from_numpy(rng.normal(0, 0.1, v.shape).astype(np.float32))
        else:
            fan = max(1, int(np.prod(v.shape[1:])))
            new[k] = torch.from_numpy((rng.normal(0, 1.0, v.shape) / np.sqrt(fan)).astype(np.float32))
    net.load_state_dict(new)
    return {k: v.clone() for k, v in new.items()}


def make_batch(rng, n, size, label_type):
    x = torch.from_numpy(rng.uniform(-1, 1, (n, 1, size, size)).astype(np.float32))
    if label_type == "distance":
        a = torch.from_numpy(rng.uniform(0, 1, (n, 1, size, size)).astype(np.float32))
        b = torch.from_numpy(rng.uniform(0, 1, (n, 1, size, size)).astype(np.float32))
        # push some residuals beyond |d| > 1 so both SmoothL1 branches are exercised
        a[:, :, :4] += 3.0
        return x, a, b
    y = torch.from_numpy(rng.integers(0, 3, (n, size, size)).astype(np.int64))
    return x, y, None


def loss_of(net, crit, batch, label_type):
    x, l1, l2 = batch
    if label_type == "distance":
        border, cell = net(x)
        return crit["border"](border, l1) + crit["cell"](cell, l2), (border, cell)
    out = net(x)
    return crit(out, l1), (out,)


def fwd_bwd_fixture(name, ut, act, norm, filters, ch_out, label_type, pool="conv", seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    net = build_unet(ut, act, pool, norm, "cpu", 1, ch_out=ch_out, filters=filters)
    sd0 = seeded_state(net, rng)
    batch = make_batch(rng, 2, 32, label_type)
    crit = get_loss("smooth_l1" if label_type == "distance" else "ce_dice", label_type)
    out = {f"w/{k}": v.numpy() for k, v in sd0.items()}
    out["x"] = batch[0].numpy()
    out["label1"] = batch[1].numpy()
    if batch[2] is not None:
        out["label2"] = batch[2].numpy()
    # eval-mode forward
    net.eval()
    with torch.no_grad():
        _, outs = loss_of(net, crit, batch, label_type)
    for i, o in enumerate(outs):
        out[f"eval_out{i}"] = o.numpy()
    # train-mode forward + backward
    net.train()
    loss, outs = loss_of(net, crit, batch, label_type)
    loss.backward()
    for i, o in enumerate(outs):
        out[f"train_out{i}"] = o.detach().numpy()
    out["loss"] = np.float32(loss.item())
    for k, p in net.named_parameters():
        out[f"g/{k}"] = p.grad.numpy()
    sd1 = net.state_dict()
    for k, v in sd1.items():
        if "running_" in k or "num_batches" in k:
            out[f"after/{k}"] = v.numpy()
    np.savez_compressed(OUT / f"unet_{name}.npz", **out)
    print(name, "loss", loss.item(), "params", sum(p.numel() for p in net.parameters()))


def trajectory_fixture(name, ut, act, norm, filters, ch_out, label_type, opt_name, steps, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    net = build_unet(ut, act, "conv", norm, "cpu", 1, ch_out=ch_out, filters=filters)
    sd0 = seeded_state(net, rng)
    batches = [make_batch(rng, 2, 32, label_type) for _ in range(2)]
    crit = get_loss("smooth_l1" if label_type == "distance" else "ce_dice", label_type)
    if opt_name == "adam":   # train.py:380-385
        opt = torch.optim.Adam(net.parameters(), lr=8e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, amsgrad=True)
    else:                    # train.py:414-420
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger(net.parameters(), lr=6e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-6,
                         weight_decay=0, use_gc=True, gc_conv_only=False, gc_loc=True)
    net.train()
    losses = []
    for s in range(steps):
        opt.zero_grad()
        loss, _ = loss_of(net, crit, batches[s % 2], label_type)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out = {f"w/{k}": v.numpy() for k, v in sd0.items()}
    for i, b in enumerate(batches):
        out[f"x{i}"] = b[0].numpy()
        out[f"label1_{i}"] = b[1].numpy()
        if b[2] is not None:
            out[f"label2_{i}"] = b[2].numpy()
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in net.state_dict().items():
        out[f"final/{k}"] = v.numpy()
    np.savez_compressed(OUT / f"traj_{name}.npz", **out)
    print("traj", name, losses[0], "->", losses[-1])


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    for i, v in enumerate(VARIANTS):
        fwd_bwd_fixture(*v, seed=1000 + i)
    trajectory_fixture("adam_DU_bn_relu", "DU", "relu", "bn", (8, 16), 1, "distance", "adam", 8, 2001)
    trajectory_fixture("ranger_DU_bn_mish", "DU", "mish", "bn", (8, 16), 1, "distance", "ranger", 14, 2002)
    trajectory_fixture("adam_U_gn_relu", "U", "relu", "gn", (8, 16), 3, "boundary", "adam", 6, 2003)
