import sys, torch
sys.path.insert(0, "/root/repo")
from microbeseg_amd import engine
from microbeseg_amd.utils.unets import build_unet
from microbeseg_amd.training.losses import get_loss
from microbeseg_amd.training.optim import make_adam
engine.set_precision("bf16")
dev = torch.device("cuda:0")
net = build_unet("DU", "relu", "conv", "bn", dev, 1, filters=(64, 1024))
opt = make_adam(net.parameters())
crit = get_loss("smooth_l1", "distance")
x = torch.rand(8, 1, 320, 320, device=dev) * 2 - 1
l1, l2 = torch.rand(8, 1, 320, 320, device=dev), torch.rand(8, 1, 320, 320, device=dev)
net.train()
def step():
    opt.zero_grad()
    b, c = net(x)
    loss = crit["border"](b, l1) + crit["cell"](c, l2)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::clone", "aten::_foreach_add_", "aten::mul_", "aten::to")]
for e in sorted(rows, key=lambda e: -e.count)[:25]:
    print(e.key, e.count, e.input_shapes[:3])
