"""GPU box: randomized stress of both post-processing methods against the C oracle (bit-exact labels expected).
python tools/stress_postproc.py [n_frames]"""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from microbeseg_amd.inference import postprocessing as pp
from microbeseg_amd.utils import synth
from oracle import postproc_ref

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for k in range(n_frames):
    rng = np.random.Generator(np.random.PCG64(31337 + k))
    H, W = int(rng.integers(40, 420)), int(rng.integers(40, 420))
    n = int(rng.integers(1, max(2, H * W // 600)))
    cell, border = synth.synth_prediction_maps(rng, H, W, n, rmin=float(rng.uniform(3, 6)), rmax=float(rng.uniform(7, 16)))
    if k % 5 == 4:                                   # quantised maps: many exact ties -> taint / serial paths
        q = float(rng.choice([16, 64, 256]))
        cell, border = np.round(cell * q) / q, np.round(border * q) / q
    ths, thc = float(rng.choice([0.35, 0.45])), float(rng.choice([0.05, 0.075, 0.1, 0.125]))
    got = pp.distance_postprocessing(border[..., None].astype(np.float32), cell[..., None].astype(np.float32), ths, thc)
    want = postproc_ref.distance_postprocessing(border[..., None].astype(np.float32), cell[..., None].astype(np.float32), ths, thc)
    ok_d = np.array_equal(got, want)
    gaps = rng.uniform(0, 1, (H, W)) < float(rng.uniform(0, 0.5))
    p1 = np.clip(cell * 2.5, 0, 1) * (1 - np.clip(border * 1.2, 0, 1) * ~gaps)
    p2 = np.clip(border * 1.2, 0, 1) * (cell > 0.02) * ~gaps
    p0 = np.clip(1 - p1 - p2, 0.0, 1)
    probs = np.stack([p0, p1, p2], -1).astype(np.float32)
    probs /= np.maximum(probs.sum(-1, keepdims=True), 1e-6)
    gb = pp.boundary_postprocessing(probs)
    wb = postproc_ref.boundary_postprocessing(probs)
    ok_b = np.array_equal(gb, wb)
    if not (ok_d and ok_b):
        bad += 1
        print(f"frame {k} {H}x{W} n={n}: distance {'ok' if ok_d else 'MISMATCH ' + str(int((got != want).sum()))} "
              f"boundary {'ok' if ok_b else 'MISMATCH ' + str(int((gb != wb).sum()))}")
print(f"{n_frames} frames, {bad} with mismatches")
sys.exit(1 if bad else 0)
