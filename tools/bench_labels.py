#!/usr/bin/env python3
"""Distance-label creation on the MI355X vs the numpy / scipy oracle: parity statistics on random cell
layouts and timing of a batch of 320 x 320 crops (the reference's training-crop size).  GPU box only.
  python tools/bench_labels.py [--crops 64] [--check 24]"""
import argparse
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from microbeseg_amd import _lib  # noqa: E402
from microbeseg_amd.training import train_data_representations as T  # noqa: E402
from oracle import labels_ref  # noqa: E402
from test_labels import _cells  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=64)
    ap.add_argument("--check", type=int, default=24)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--cells", type=int, default=90)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    S = a.size
    masks = np.stack([_cells(rng, S, S, a.cells, 8, 22, int(rng.integers(0, 3))) for _ in range(a.crops)])
    sr = 40
    # parity statistics
    worst_c = worst_n = 0.0
    nbits = 0
    t0 = time.perf_counter()
    ref = [labels_ref.distance_label(masks[i], sr) for i in range(min(a.check, a.crops))]
    t_cpu = (time.perf_counter() - t0) / max(len(ref), 1)
    cell, nb = T.distance_label_batch(masks, sr)
    for i, (c, d) in enumerate(ref):
        worst_c = max(worst_c, float(np.abs(cell[i] - c).max()))
        worst_n = max(worst_n, float(np.abs(nb[i] - d).max()))
        nbits += int((cell[i] != c).sum()) + int((nb[i] != d).sum())
    print(f"parity on {len(ref)} crops {S}x{S}, ~{len(np.unique(masks[0])) - 1} cells each: max |cell diff| {worst_c:.2e}, "
          f"max |neighbor diff| {worst_n:.2e}, pixels not bit-identical {nbits} of {2 * len(ref) * S * S}")
    # device timing, data resident
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    N = a.crops
    m = torch.from_numpy(masks.astype(np.uint16).view(np.int16)).to(dev)
    need = lib.mseg_label_distance_workspace_bytes(N, S, S)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    c = torch.empty((N, S, S), dtype=torch.float32, device=dev)
    d = torch.empty_like(c)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(lib.mseg_label_distance(m.data_ptr(), N, S, S, sr, c.data_ptr(), d.data_ptr(), ws.data_ptr(), need, st),
                   "label_distance")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"device: {ms:.3f} ms per batch of {N} crops = {N / ms * 1e3:.0f} crops/s ({N * S * S / ms / 1e3:.1f} Mpx/s); "
          f"oracle (numpy/scipy, 1 core): {t_cpu * 1e3:.0f} ms per crop = {1 / t_cpu:.2f} crops/s")


if __name__ == "__main__":
    main()
