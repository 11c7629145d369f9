"""CPU: the oracle restatement (oracle/unet_ref.py) against golden vectors produced by the REAL reference modules
(tools/gen_golden_unet.py imports /root/reference/src/utils/unets.py, losses.py, ranger2020.py)."""
import numpy as np
import pytest
import torch

from helpers import VARIANTS, grad_floor, load_npz, rel_err, state_from, variant
from oracle import unet_ref

TOL = 1e-5  # same torch build, same ops -> expected bit-equal; tolerance only guards thread-count effects


def _loss(outs, fx, label_type, suffix=""):
    l1 = torch.from_numpy(fx["label1" + suffix])
    if label_type == "distance":
        l2 = torch.from_numpy(fx["label2" + suffix])
        return unet_ref.regression_loss(outs[0], l1) + unet_ref.regression_loss(outs[1], l2)
    return unet_ref.ce_dice(outs, l1)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_forward_backward_matches_reference(name):
    ut, act, norm, filters, ch_out, label_type, pool = variant(name)
    fx = load_npz(f"unet_{name}.npz")
    sd = state_from(fx)
    x = torch.from_numpy(fx["x"])
    with torch.no_grad():
        outs = unet_ref.unet_forward(sd, x, ut, act, norm, filters, pool_method=pool, training=False)
    outs = outs if isinstance(outs, tuple) else (outs,)
    for i, o in enumerate(outs):
        assert rel_err(o, fx[f"eval_out{i}"]) < TOL
    params = {k: v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()
              for k, v in sd.items()}
    outs = unet_ref.unet_forward(params, x, ut, act, norm, filters, pool_method=pool, training=True,
                                 update_running_stats=True)
    outs_t = outs if isinstance(outs, tuple) else (outs,)
    for i, o in enumerate(outs_t):
        assert rel_err(o.detach(), fx[f"train_out{i}"]) < TOL
    loss = _loss(outs, fx, label_type)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    loss.backward()
    floor = grad_floor(fx)
    for k, v in fx.items():
        if k.startswith("g/"):
            assert rel_err(params[k[2:]].grad, v, floor) < 1e-4, k
        if k.startswith("after/"):
            got = params[k[6:]]
            assert np.allclose(got.detach().numpy(), v, rtol=1e-5, atol=1e-6), k


def _run_traj(fxname, ut, act, norm, filters, label_type, make_opt, steps):
    fx = load_npz(fxname)
    sd = state_from(fx)
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    plist = [v for v in params.values() if v.requires_grad]
    opt = make_opt(plist)
    losses = []
    for s in range(steps):
        opt.zero_grad()
        b = s % 2
        outs = unet_ref.unet_forward(params, torch.from_numpy(fx[f"x{b}"]), ut, act, norm, filters, training=True,
                                     update_running_stats=True)
        loss = _loss(outs, fx, label_type, suffix=f"_{b}")
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, fx["losses"], rtol=2e-4, atol=1e-6)
    for k, v in fx.items():
        if k.startswith("final/") and not k.endswith("num_batches_tracked"):
            assert rel_err(params[k[6:]].detach(), v) < 2e-3, k


def test_adam_trajectory_matches_reference():
    _run_traj("traj_adam_DU_bn_relu.npz", "DU", "relu", "bn", (8, 16), "distance",
              lambda ps: torch.optim.Adam(ps, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True), 8)


def test_ranger_trajectory_matches_reference():
    # 14 steps: crosses the N_sma > 5 switch (step 6) and two lookahead syncs (k = 6)
    _run_traj("traj_ranger_DU_bn_mish.npz", "DU", "mish", "bn", (8, 16), "distance",
              lambda ps: unet_ref.RangerRef(ps, lr=6e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999),
                                            eps=1e-6), 14)


def test_adam_ce_dice_trajectory_matches_reference():
    _run_traj("traj_adam_U_gn_relu.npz", "U", "relu", "gn", (8, 16), "boundary",
              lambda ps: torch.optim.Adam(ps, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True), 6)
