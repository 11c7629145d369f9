"""Diagnostic (GPU box): per-node dz (dL/d conv output) of the HIP backward vs CPU autograd."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from microbeseg_amd.utils.unets import build_unet
from microbeseg_amd import engine
from oracle import unet_ref

ut, act, norm, filters, size, batch = sys.argv[1], sys.argv[2], sys.argv[3], (int(sys.argv[4]), int(sys.argv[5])), int(sys.argv[6]), int(sys.argv[7])
dev = torch.device("cuda:0")
torch.manual_seed(1234)
net = build_unet(ut, act, "conv", norm, dev, 1, ch_out=3 if ut == "U" else 1, filters=filters)
sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
x = torch.rand(batch, 1, size, size) * 2 - 1

hip_dz = []
orig = engine.norm_bwd
def patched(node, gy, *a, **k):
    gy_in = gy.clone()
    out = orig(node, gy, *a, **k)
    hip_dz.append((node, gy_in.cpu(), out.clone().cpu()))
    return out
engine.norm_bwd = patched
net.train()
outs = net(x.to(dev)); outs = outs if isinstance(outs, tuple) else (outs,)
gos = [torch.randn(o.shape) for o in outs]
torch.autograd.backward(outs, [g.to(dev) for g in gos])

zs = []
c2, ct = F.conv2d, F.conv_transpose2d
def conv2d(*a, **k):
    o = c2(*a, **k); o.retain_grad(); zs.append(o); return o
def convt(*a, **k):
    o = ct(*a, **k); o.retain_grad(); zs.append(o); return o
unet_ref.F.conv2d, unet_ref.F.conv_transpose2d = conv2d, convt
ys = []
bn0 = F.batch_norm
def bnp(*a, **k):
    o = bn0(*a, **k); o.retain_grad(); ys.append(o); return o
unet_ref.F.batch_norm = bnp
params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
ref = unet_ref.unet_forward(params, x, ut, act, norm, filters, training=True, update_running_stats=True)
ref = ref if isinstance(ref, tuple) else (ref,)
torch.autograd.backward(ref, gos)
# zs order: enc convs..., dec1 (up,c1,c2)*, head1, dec2..., head2 ; drop heads (1x1 convs with 1/3 out channels at the end of each decoder)
nlev = len(unet_ref._levels(filters))
per_dec = 3 * (nlev - 1) + 1
n_enc = len(zs) - per_dec * len(ref)
order = list(range(n_enc))
for d in range(len(ref)):
    base = n_enc + d * per_dec
    order += list(range(base, base + per_dec - 1))
ref_nodes = [zs[i] for i in order]
hip = list(reversed(hip_dz))
assert len(hip) == len(ref_nodes), (len(hip), len(ref_nodes))
def rel(a, b): return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)
for i, ((node, gy, dz), z) in enumerate(zip(hip, ref_nodes)):
    g = z.grad.permute(0, 2, 3, 1)
    print(f"{i:2d} {node.layer.kind:5s} C={node.C:4d} {node.H}x{node.W}  dz err {rel(dz, g):.2e}   z err {rel(node.z.cpu(), z.detach().permute(0,2,3,1)):.2e}")
print("--- norm_bwd in isolation (torch CPU backward of act->norm given the HIP gy and z)")
for i, (node, gy, dz) in enumerate(hip):
    z = node.z.cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    a = unet_ref._act(z, {0: None, 1: "relu", 2: "leakyrelu", 3: "elu", 4: "mish"}[node.act]) if node.act else z
    nm = node.layer.norm_mod
    if norm == "bn":
        y = F.batch_norm(a, None, None, nm.weight.detach().cpu(), nm.bias.detach().cpu(), True, 0.1, 1e-5)
    elif norm == "gn":
        y = F.group_norm(a, 8, nm.weight.detach().cpu(), nm.bias.detach().cpu(), 1e-5)
    else:
        y = F.instance_norm(a, eps=1e-5)
    y.backward(gy.permute(0, 3, 1, 2))
    print(f"{i:2d} {node.layer.kind:5s} C={node.C:4d}  norm_bwd err {rel(dz, z.grad.permute(0, 2, 3, 1)):.2e}")

print("--- gy (input of norm_bwd) vs autograd dL/dy")
ref_y = [ys[i] for i in order] if len(ys) == len(zs) - len(ref) else None
# ys has one entry per normalised conv (heads have no norm): same order as ref_nodes
ref_y = ys
for i, ((node, gy, dz), y) in enumerate(zip(hip, ref_y)):
    g = y.grad.permute(0, 2, 3, 1)
    d = (gy - g).abs()
    bad = d > 1e-3 * g.abs().max()
    print(f"{i:2d} {node.layer.kind:5s} gy err {rel(gy, g):.2e}  bad elems {int(bad.sum())}/{bad.numel()}")
    if bad.any() and i in (4, 6):
        idx = bad.nonzero()
        print("   n:", sorted(set(idx[:, 0].tolist())), " y range", idx[:, 1].min().item(), idx[:, 1].max().item(),
              " x range", idx[:, 2].min().item(), idx[:, 2].max().item(), " c range", idx[:, 3].min().item(), idx[:, 3].max().item())
        print("   first bad:", idx[:12].tolist())
        print("   per-channel bad counts:", bad.sum((0, 1, 2)).tolist())
