"""GPU box: device augmentation of a 32 x 256 x 256 training batch vs the numpy / scipy restatement (one host core,
like one DataLoader worker of the reference).  python tools/bench_augment.py [batch] [size]"""
import pathlib
import random
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from microbeseg_amd.training import device_augment as da
from oracle import augment_ref

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(1)
img = rng.integers(0, 65536, (B, S, S)).astype(np.uint16)
lab = rng.random((B, S, S)).astype(np.float32)
t_img = torch.from_numpy(img.astype(np.int32)).cuda()
t_lab = torch.from_numpy(lab).cuda()
aug = da.DeviceAugment("distance", 0, 65535, seed=3)
for _ in range(3):
    aug(t_img, [(t_lab, "linear"), (t_lab, "linear")])
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    aug(t_img, [(t_lab, "linear"), (t_lab, "linear")])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"device: {dt*1e3:.2f} ms per batch of {B} x {S}^2 = {B/dt:.0f} crops/s (host parameter drawing + H2D of parameters included)")
p = da.draw_parameters(B, random.Random(3), np.random.default_rng(3))
nrng = np.random.default_rng(0)
t0 = time.perf_counter()
for i in range(B):
    augment_ref.augment_sample(img[i], [(lab[i], "linear"), (lab[i], "linear")], p, i, noise_rng=nrng)
dt_cpu = time.perf_counter() - t0
print(f"numpy/scipy restatement, 1 core: {dt_cpu*1e3:.0f} ms per batch = {B/dt_cpu:.0f} crops/s")
