import sys, time, pathlib
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from microbeseg_amd.utils import synth
from microbeseg_amd.inference import postprocessing as pp
for S in (512, 1024, 2048):
    rng = np.random.Generator(np.random.PCG64(7))
    cell, border = synth.synth_prediction_maps(rng, S, S, int(2500 * (S / 2048.0) ** 2), rmin=5.0, rmax=13.0)
    # boundary-method style probabilities: interior where cell high & border low, boundary where border high, else bg
    p1 = np.clip(cell * 2.0, 0, 1) * (1 - np.clip(border * 1.2, 0, 1))
    p2 = np.clip(border * 1.2, 0, 1) * (cell > 0.02)
    p0 = np.clip(1 - p1 - p2, 0.0, 1)
    probs = np.stack([p0, p1, p2], -1).astype(np.float32)
    probs /= probs.sum(-1, keepdims=True)
    t = torch.from_numpy(probs).cuda()
    lab, n, st = pp.boundary_postprocessing_device(t); torch.cuda.synchronize()
    t0 = time.time(); lab, n, st = pp.boundary_postprocessing_device(t); torch.cuda.synchronize(); dt = time.time() - t0
    print(S, "boundary postproc %.1f ms" % (dt * 1e3), "instances", int(n), "status", int(st), "fg", float((lab != 0).float().mean()))
