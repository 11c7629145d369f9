"""GPU box: network forward on large frames, tiled vs whole, and mask -> polygon tracing (§6e).
usage: python tools/bench_frames.py [sizes...]      (default 2048 3200 4096)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microbeseg_amd.utils.unets import build_unet                         # noqa: E402
from microbeseg_amd.inference import postprocessing as pp, tiling          # noqa: E402
from microbeseg_amd.utils import synth, hull_polygon                       # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 3200, 4096]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = build_unet("DU", "relu", "conv", "bn", dev, 1, ch_out=1, filters=(64, 1024)).eval()
    with torch.no_grad():
        for S in sizes:
            x = torch.rand(1, 1, S, S, device=dev) * 2 - 1
            t = timed(lambda: net(x), 3)
            line = f"forward {S}x{S}: {t:8.1f} ms  {S * S / t / 1e3:6.1f} Mpx/s"
            if S > 2048:
                tt = timed(lambda: tiling.tiled_forward(net, x, tile=2048), 2)
                line += f"   tiled (2048 + 128 halo): {tt:8.1f} ms"
            print(line, flush=True)
            del x
            torch.cuda.empty_cache()
    rng = np.random.Generator(np.random.PCG64(2024))
    S = 2048
    cell, border = synth.synth_prediction_maps(rng, S, S, 2500, rmin=5.0, rmax=13.0)
    c, b = torch.from_numpy(cell).to(dev), torch.from_numpy(border).to(dev)
    labels, n_inst, _ = pp.distance_postprocessing_device(b, c, 0.45, 0.10)
    lab16 = labels if labels.dtype == torch.int16 else labels.to(torch.int16)
    t = timed(lambda: hull_polygon.label_polygons_device(lab16), 5)
    ids, _, offsets, pts = hull_polygon.label_polygons_device(lab16)
    print(f"polygons 2048x2048: {int(n_inst)} instances, {len(ids)} polygons, {int(offsets[-1])} points: {t:.2f} ms "
          f"(device passes + the host sort of the records)", flush=True)
    t0 = time.perf_counter()
    hull_polygon.label_polygons(lab16)
    print(f"  incl. copy-out and the per-instance python dict: {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
