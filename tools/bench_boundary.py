"""GPU box: the boundary method's post-processing (reference postprocessing.py:62-90) on a synthetic 2048^2 frame with
~2500 cells — the closed-form marker phase (mseg_postproc_set_const_stream(1), default) against the replay of the heap (0):
time per frame, equal labels, and both against the C oracle."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from microbeseg_amd import _lib
from microbeseg_amd.inference import postprocessing as pp
from microbeseg_amd.utils import synth
from oracle import postproc_ref

lib = _lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(5)
cell, border = synth.synth_prediction_maps(rng, S, S, int(2500 * (S / 2048.0) ** 2), rmin=5.0, rmax=13.0)
probs = torch.from_numpy(bench._boundary_probs(cell, border)).cuda()
res = {}
for mode in (1, 0):
    lib.mseg_postproc_set_const_stream(mode)
    pp.boundary_postprocessing_device(probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        bl, bn, bs = pp.boundary_postprocessing_device(probs)
    e1.record()
    torch.cuda.synchronize()
    res[mode] = bl.cpu().numpy().view(np.uint16)
    print(f"const_stream={mode}: {e0.elapsed_time(e1) / 3:8.2f} ms per frame, {int(bn)} instances, status {int(bs)}", flush=True)
lib.mseg_postproc_set_const_stream(1)
print("stream == heap replay:", bool(np.array_equal(res[0], res[1])))
t0 = time.perf_counter()
want = postproc_ref.boundary_postprocessing(probs.cpu().numpy())
print(f"C oracle {time.perf_counter() - t0:.2f} s; == oracle:", bool(np.array_equal(res[1], want)))
