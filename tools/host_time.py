"""GPU box: host enqueue time vs device time of one training step (is the step host-bound?)."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd import engine
from microbeseg_amd.utils.unets import build_unet
from microbeseg_amd.training.losses import get_loss
from microbeseg_amd.training.optim import make_adam
prec, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
norm = sys.argv[4] if len(sys.argv) > 4 else "bn"
engine.set_precision(prec)
dev = torch.device("cuda:0")
net = build_unet("DU", "relu", "conv", norm, dev, 1, filters=(64, 1024))
opt = make_adam(net.parameters())
crit = get_loss("smooth_l1", "distance")
x = torch.rand(batch, 1, size, size, device=dev) * 2 - 1
l1, l2 = torch.rand(batch, 1, size, size, device=dev), torch.rand(batch, 1, size, size, device=dev)
net.train()
def step():
    opt.zero_grad()
    b, c = net(x)
    loss = crit["border"](b, l1) + crit["cell"](c, l2)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{prec} {size} b{batch}: host enqueue {1e3*(t1-t0)/K:.2f} ms/step, wall {1e3*(t2-t0)/K:.2f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
