"""GPU (MI355X): the full HIP U-Net / DU-Net (forward, fused losses, explicit backward, optimizer steps) against
  (a) golden vectors produced by the real reference modules (tests/golden/unet_*.npz, traj_*.npz), and
  (b) the CPU oracle (oracle/unet_ref.py) on larger seeded inputs.
Tolerance 1e-4 relative (BASELINE.json north_star)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from helpers import VARIANTS, grad_floor, load_npz, rel_err, state_from, variant

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _build(name, dev, sd):
    from microbeseg_amd.utils.unets import build_unet
    ut, act, norm, filters, ch_out, label_type, pool = variant(name)
    net = build_unet(ut, act, pool, norm, dev, 1, ch_out=ch_out, filters=filters)
    net.load_state_dict(sd)
    return net


def _loss(net, fx, label_type, dev, sfx=""):
    from microbeseg_amd.training.losses import get_loss
    x = torch.from_numpy(fx["x" + sfx[1:]] if sfx else fx["x"]).to(dev)
    l1 = torch.from_numpy(fx["label1" + sfx]).to(dev)
    if label_type == "distance":
        crit = get_loss("smooth_l1", "distance")
        l2 = torch.from_numpy(fx["label2" + sfx]).to(dev)
        border, cell = net(x)
        return crit["border"](border, l1) + crit["cell"](cell, l2), (border, cell)
    crit = get_loss("ce_dice", "boundary")
    out = net(x)
    return crit(out, l1), (out,)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_net_matches_reference_fixture(name, dev):
    label_type = variant(name)[5]
    fx = load_npz(f"unet_{name}.npz")
    net = _build(name, dev, state_from(fx))
    net.eval()
    with torch.no_grad():
        _, outs = _loss(net, fx, label_type, dev)
    for i, o in enumerate(outs):
        assert rel_err(o.cpu(), fx[f"eval_out{i}"]) < TOL, f"eval_out{i}"
    net.train()
    loss, outs = _loss(net, fx, label_type, dev)
    for i, o in enumerate(outs):
        assert rel_err(o.detach().cpu(), fx[f"train_out{i}"]) < TOL, f"train_out{i}"
    assert abs(loss.item() - float(fx["loss"])) < TOL * max(1.0, abs(float(fx["loss"])))
    loss.backward()
    floor = grad_floor(fx)
    params = dict(net.named_parameters())
    for k, v in fx.items():
        if k.startswith("g/"):
            assert params[k[2:]].grad is not None, k
            assert rel_err(params[k[2:]].grad.cpu(), v, floor) < 5 * TOL, k
    sd = net.state_dict()
    for k, v in fx.items():
        if k.startswith("after/"):
            assert np.allclose(sd[k[6:]].cpu().numpy(), v, rtol=1e-4, atol=1e-6), k


def _traj(fxname, name, make_opt, steps, dev, tol_w):
    label_type = variant(name)[5]
    fx = load_npz(fxname)
    net = _build(name, dev, state_from(fx))
    net.train()
    opt = make_opt(net.parameters())
    losses = []
    for s in range(steps):
        opt.zero_grad()
        loss, _ = _loss(net, fx, label_type, dev, sfx=f"_{s % 2}")
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, fx["losses"], rtol=1e-3, atol=1e-5), (losses, fx["losses"])
    sd = net.state_dict()
    for k, v in fx.items():
        if k.startswith("final/") and not k.endswith("num_batches_tracked"):
            assert rel_err(sd[k[6:]].cpu(), v) < tol_w, k


def test_adam_trajectory(dev):
    _traj("traj_adam_DU_bn_relu.npz", "DU_bn_relu_8_16",
          lambda ps: torch.optim.Adam(ps, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True), 8, dev,
          5e-3)


def test_ranger_trajectory(dev):
    from microbeseg_amd.training.ranger2020 import Ranger
    def mk(ps):
        with contextlib.redirect_stdout(io.StringIO()):
            return Ranger(ps, lr=6e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-6,
                          weight_decay=0, use_gc=True, gc_conv_only=False, gc_loc=True)
    _traj("traj_ranger_DU_bn_mish.npz", "DU_bn_mish_8_16", mk, 14, dev, 2e-2)


def test_adam_ce_dice_trajectory(dev):
    _traj("traj_adam_U_gn_relu.npz", "U_gn_relu_8_16",
          lambda ps: torch.optim.Adam(ps, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True), 6, dev,
          5e-3)


def _l2_rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("ut,act,norm,filters,size,batch", [
    ("DU", "mish", "bn", (16, 128), 64, 3),      # 4 levels, odd batch
    ("U", "mish", "gn", (32, 64), 96, 2),
    ("DU", "elu", "bn", (64, 128), 48, 2),       # 64-channel level 0 like the default net
    ("DU", "relu", "bn", (64, 128), 48, 2),      # ReLU: see the note on mask flips below
])
def test_net_matches_oracle_on_larger_inputs(ut, act, norm, filters, size, batch, dev):
    """Same seeded weights / inputs through the HIP net and the CPU oracle, forward + parameter gradients.

    ReLU note: with ~3e5 activations per layer a few pre-activations land within fp32 rounding of 0, where the HIP
    and CPU summation orders disagree on the sign; each such flip switches one element of dz on/off (verified with
    tools/diag_dz.py: gy and z agree to 1e-6, dz differs only at those elements).  With N(0,1) upstream gradients a
    single flip moves max|dW| by ~1e-2, so the ReLU variant is checked in relative L2 norm; the smooth activations
    (and the small ReLU fixtures above) are checked element-wise."""
    from microbeseg_amd.utils.unets import build_unet
    from oracle import unet_ref
    torch.manual_seed(1234)
    ch_out = 3 if ut == "U" else 1
    net = build_unet(ut, act, "conv", norm, dev, 1, ch_out=ch_out, filters=filters)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(batch, 1, size, size) * 2 - 1
    net.train()
    outs = net(x.to(dev))
    outs = outs if isinstance(outs, tuple) else (outs,)
    gos = [torch.randn(o.shape) for o in outs]
    torch.autograd.backward(outs, [g.to(dev) for g in gos])

    def oracle(dtype):
        params = {k: (v.clone().to(dtype).requires_grad_(True) if v.is_floating_point() and "running" not in k
                      else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
        ref = unet_ref.unet_forward(params, x.to(dtype), ut, act, norm, filters, training=True,
                                    update_running_stats=True)
        ref = ref if isinstance(ref, tuple) else (ref,)
        torch.autograd.backward(ref, [g.to(dtype) for g in gos])
        return params, ref

    p32, r32 = oracle(torch.float32)     # the reference arithmetic (torch CPU fp32)
    p64, r64 = oracle(torch.float64)     # ground truth
    for o, r in zip(outs, r32):
        assert rel_err(o.detach().cpu(), r.detach()) < TOL
    # Parameter gradients of a deep BatchNorm net are ill-conditioned: torch-CPU fp32 itself is ~1e-3 away from the
    # fp64 truth.  The HIP path must be as accurate as the reference arithmetic: per parameter within 4x of the
    # fp32 oracle's own error, or no worse than the fp32 oracle's worst parameter, or within 5*TOL.
    floor = 1e-3 * max(p.grad.abs().max().item() for p in p64.values() if getattr(p, "grad", None) is not None)
    if act == "relu":
        for k, p in net.named_parameters():
            if p64[k].grad.abs().max().item() > floor:
                assert _l2_rel(p.grad.cpu(), p64[k].grad) < 2e-2, k
        return
    e_ref_all = {k: rel_err(p32[k].grad, p64[k].grad, floor) for k, _ in net.named_parameters()}
    worst_ref = max(e_ref_all.values())
    for k, p in net.named_parameters():
        e_hip = rel_err(p.grad.cpu(), p64[k].grad, floor)
        assert e_hip < max(5 * TOL, 4 * e_ref_all[k], worst_ref), (k, e_hip, e_ref_all[k], worst_ref)
    for k, v in net.state_dict().items():
        if "running" in k:
            assert rel_err(v.cpu(), p32[k]) < 1e-4, k


def _bf16_rule(kind, xs, ws):
    """which of (forward, data gradient, weight gradient) of a layer the engine runs with bf16 operands: mirror of
    engine._bf16_launch / engine._wgrad_bf16_ok for the shapes of this test (first layer: VALU kernels, fp32)"""
    N, cin, H, W = xs

    def rows_ok(h, w, pix):                               # pixel blocks of the bf16 weight-gradient kernel
        th = pix // (8 if w % 8 == 0 else 4)
        return w % 4 == 0 and h * 2 >= ((h + th - 1) // th) * th
    if kind == "up":                                      # P = the layer's input (H x W)
        return (True, True, rows_ok(H, W, 32))
    if kind == "pool":                                    # P = dz (H/2 x W/2); data gradient: whole 128-row parity tiles
        return (True, (N * H * W) % 512 == 0, H % 2 == 0 and W % 2 == 0 and rows_ok(H // 2, W // 2, 32))
    if cin <= 4:
        return (False, False, False)
    return (True, True, rows_ok(H, W, 64))   # forward / data gradient: halo kernel, or the gather kernel where it does not tile


@pytest.mark.parametrize("ut,act,norm,filters,size,batch", [("DU", "elu", "bn", (64, 128), 64, 2),
                                                            ("U", "mish", "bn", (64, 128), 48, 3),
                                                            ("DU", "mish", "gn", (64, 128), 64, 2),     # per-sample tables
                                                            ("U", "elu", "in", (64, 128), 64, 2)])
def test_bf16_mode_matches_bf16_oracle(ut, act, norm, filters, size, batch, dev):
    """BASELINE configs[2] (bf16 forward / backward, fp32 accumulate and norm statistics): engine.set_precision('bf16')
    against the oracle with the SAME rounding points (oracle/unet_ref.py BF16_RULE: operands of the 3x3 stride-1
    convolutions rounded to bf16, everything else fp32).  bf16 has its own tolerance: an operand whose fp32 value differs in
    the last bits between the two implementations can round to a different bf16 neighbour (2^-8 relative), so the
    agreement is ~1e-3 on the outputs — a quarter of the distance between bf16 and fp32 arithmetic itself, which is also
    checked (the bf16 path must stay within 3e-2 of the fp32 reference arithmetic)."""
    from microbeseg_amd import engine
    from microbeseg_amd.utils.unets import build_unet
    from oracle import unet_ref
    torch.manual_seed(1234)
    ch_out = 3 if ut == "U" else 1
    net = build_unet(ut, act, "conv", norm, dev, 1, ch_out=ch_out, filters=filters)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(batch, 1, size, size) * 2 - 1
    net.train()
    engine.set_precision("bf16")
    try:
        outs = net(x.to(dev))
        outs = outs if isinstance(outs, tuple) else (outs,)
        gos = [torch.randn(o.shape) for o in outs]
        torch.autograd.backward(outs, [g.to(dev) for g in gos])
    finally:
        engine.set_precision("fp32")

    def oracle(rule):
        unet_ref.BF16_RULE = rule
        try:
            params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                      for k, v in sd.items()}
            ref = unet_ref.unet_forward(params, x, ut, act, norm, filters, training=True, update_running_stats=True)
            ref = ref if isinstance(ref, tuple) else (ref,)
            torch.autograd.backward(ref, gos)
        finally:
            unet_ref.BF16_RULE = None
        return params, ref

    p16, r16 = oracle(_bf16_rule)
    p32, r32 = oracle(None)
    for o, a, b in zip(outs, r16, r32):
        assert rel_err(o.detach().cpu(), a.detach()) < 5e-3
        assert 1e-4 < rel_err(o.detach().cpu(), b.detach()) < 3e-2        # really bf16, and a sane bf16
    floor = 1e-3 * max(p.grad.abs().max().item() for p in p32.values() if getattr(p, "grad", None) is not None)
    for k, p in net.named_parameters():
        if p32[k].grad.abs().max().item() <= floor:
            continue
        noise = _l2_rel(p16[k].grad, p32[k].grad)                          # what bf16 rounding itself does to this gradient
        assert _l2_rel(p.grad.cpu(), p16[k].grad) < max(2e-2, noise), k
        assert _l2_rel(p.grad.cpu(), p32[k].grad) < max(4e-2, 2 * noise), k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_train_step_equals_eager(precision, dev, request):
    """hipGraph replay of the training step (training/graph_step.py): same losses and bit-identical weights as the eager
    step on the same batches (the kernels and their order are the same; only the launches are pre-recorded)."""
    from microbeseg_amd import engine
    from microbeseg_amd.training.graph_step import GraphedTrainStep
    from microbeseg_amd.training.losses import get_loss
    from microbeseg_amd.utils.unets import build_unet
    request.addfinalizer(lambda: engine.set_precision("fp32"))
    engine.set_precision(precision)
    crit = get_loss("smooth_l1", "distance")
    g = torch.Generator().manual_seed(4)
    batches = [(torch.rand(4, 1, 64, 64, generator=g) * 2 - 1, torch.rand(4, 1, 64, 64, generator=g),
                torch.rand(4, 1, 64, 64, generator=g)) for _ in range(7)]
    batches = [tuple(t.to(dev) for t in b) for b in batches]
    odd = tuple(t[:3].contiguous() for t in batches[0])        # a smaller batch in between: falls back to eager

    def run(graphed):
        torch.manual_seed(11)
        net = build_unet("DU", "relu", "conv", "bn", dev, 1, ch_out=1, filters=(64, 128))
        opt = torch.optim.Adam(net.parameters(), lr=8e-4, betas=(0.9, 0.999), eps=1e-8, amsgrad=True, capturable=True)
        net.train()

        def step(x, lb, lc):
            opt.zero_grad(set_to_none=True)
            border, cell = net(x)
            loss = crit["border"](border, lb) + crit["cell"](cell, lc)
            loss.backward()
            opt.step()
            return loss
        fn = GraphedTrainStep(step, opt, warmup=2) if graphed else step
        losses = []
        for i, b in enumerate(batches):
            losses.append(float(fn(*b).detach()))
            if i == 4:
                losses.append(float(fn(*odd).detach()))
        return losses, {k: v.detach().clone() for k, v in net.state_dict().items()}

    l_e, sd_e = run(False)
    l_g, sd_g = run(True)
    assert l_e == l_g
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
