"""GPU box: HIP network vs the CPU oracle (fp32 and fp64) on non-square inputs and odd level sizes — exercises the kernel
selection rules (halo tile widths 64..4, pixel blocks 32..4, gather fall-backs, per-sample tables) away from the powers of
two of the benchmark.  python tools/sweep_shapes.py"""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd.utils.unets import build_unet
from oracle import unet_ref

CASES = [("DU", "elu", "bn", "conv", (8, 64), 1, 48, 80), ("U", "mish", "gn", "conv", (16, 64), 2, 96, 160),
         ("DU", "elu", "in", "conv", (8, 32), 3, 40, 72), ("DU", "relu", "bn", "conv", (16, 128), 2, 160, 96),
         ("U", "elu", "bn", "max", (8, 32), 2, 64, 48), ("DU", "leakyrelu", "gn", "conv", (8, 16), 5, 320, 320),
         ("DU", "relu", "bn", "conv", (64, 256), 2, 80, 80), ("U", "relu", "gn", "conv", (64, 128), 3, 40, 200)]


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    dev = torch.device("cuda:0")
    bad = 0
    for ut, act, norm, pool, filters, B, H, W in CASES:
        torch.manual_seed(H * 7 + W)
        ch_out = 3 if ut == "U" else 1
        net = build_unet(ut, act, pool, norm, dev, 1, ch_out=ch_out, filters=list(filters))
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        x = torch.rand(B, 1, H, W) * 2 - 1
        net.train()
        outs = net(x.to(dev))
        outs = outs if isinstance(outs, tuple) else (outs,)
        gos = [torch.randn(o.shape) for o in outs]
        torch.autograd.backward(outs, [g.to(dev) for g in gos])

        def oracle(dtype):
            params = {k: (v.clone().to(dtype).requires_grad_(True) if v.is_floating_point() and "running" not in k
                          else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
            ref = unet_ref.unet_forward(params, x.to(dtype), ut, act, norm, filters, training=True,
                                        update_running_stats=True, pool_method=pool) \
                if "pool_method" in unet_ref.unet_forward.__code__.co_varnames else \
                unet_ref.unet_forward(params, x.to(dtype), ut, act, norm, filters, training=True, update_running_stats=True)
            ref = ref if isinstance(ref, tuple) else (ref,)
            torch.autograd.backward(ref, [g.to(dtype) for g in gos])
            return params, ref
        p32, r32 = oracle(torch.float32)
        p64, r64 = oracle(torch.float64)
        eo = max(rel(o.detach().cpu(), r.detach()) for o, r in zip(outs, r64))
        eo32 = max(rel(r.detach(), q.detach()) for r, q in zip(r32, r64))
        worst, worst32, wk = 0.0, 0.0, ""
        for k, p in net.named_parameters():
            g64 = p64[k].grad
            if g64 is None or g64.norm().item() < 1e-12:
                continue
            e = rel(p.grad.detach().cpu(), g64)
            e32 = rel(p32[k].grad, g64)
            if e > worst:
                worst, wk = e, k
            worst32 = max(worst32, e32)
        ok = eo < max(1e-4, 4 * eo32) and worst < max(2e-3, 6 * worst32)
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {ut}-{act}-{norm}-{pool} {filters} {B}x{H}x{W}: out {eo:.2e} (cpu fp32 {eo32:.2e})  "
              f"grad L2 worst {worst:.2e} [{wk}] (cpu fp32 worst {worst32:.2e})")
    print("FAILED" if bad else "all ok")
    return bad


if __name__ == "__main__":
    sys.exit(main())
