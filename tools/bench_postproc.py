"""GPU box: HIP post-processing vs the C oracle (single thread) on synthetic 2048x2048 prediction maps."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
from microbeseg_amd.utils import synth
from microbeseg_amd.inference import postprocessing as pp
from oracle import postproc_ref as R

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else int(2500 * (H * W) / 2048 ** 2)
rng = np.random.Generator(np.random.PCG64(2024))
t0 = time.time(); cell, border = synth.synth_prediction_maps(rng, H, W, n_cells, rmin=5.0, rmax=13.0); print("synth %.1fs" % (time.time() - t0))
c, b = torch.from_numpy(cell).cuda(), torch.from_numpy(border).cuda()
labels, n_inst, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    labels, n_inst, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"HIP  {ms:.2f} ms/frame  {H*W/ms/1e3:.1f} Mpx/s  instances {int(n_inst)} status {int(status)} fg {float((labels!=0).float().mean()):.3f}")
t0 = time.time(); want = R.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10); dt = time.time() - t0
print(f"CPU oracle (1 thread) {dt*1e3:.0f} ms/frame  {H*W/dt/1e6:.2f} Mpx/s")
got = labels.cpu().numpy().view(np.uint16)
print("bit-exact:", bool(np.array_equal(got, want)), "mismatch px", int((got != want).sum()))
from scipy import ndimage as ndi
comp, ncomp = ndi.label(got != 0)                       # 4-connected mask components = independent floods
areas = np.bincount(comp.ravel())[1:]
sl = ndi.find_objects(comp)
big = np.argsort(areas)[::-1][:5]
print(f"mask components {ncomp}: area mean {areas.mean():.0f} max {areas.max()}  largest bboxes "
      + ", ".join(f"{areas[i]}px in {sl[i][0].stop - sl[i][0].start}x{sl[i][1].stop - sl[i][1].start}" for i in big))
if len(sys.argv) > 3 and sys.argv[3] == "pp-only":
    sys.exit(0)

# ---- evaluation: 4 x 2 threshold sweep (one call, shared smoothing) vs 8 separate calls; AJI+ scoring of one pair ----
from microbeseg_amd.evaluation import stats_utils as su
from oracle import eval_ref
ths = [(tc, ts) for tc in (0.05, 0.075, 0.10, 0.125) for ts in (0.35, 0.45)]
pp.distance_postprocessing_sweep_device(b, c, ths); torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    lab8, _, _ = pp.distance_postprocessing_sweep_device(b, c, ths)
e1.record(); torch.cuda.synchronize()
ms_sweep = e0.elapsed_time(e1) / reps
e0.record()
for _ in range(reps):
    for tc, ts in ths:
        pp.distance_postprocessing_device(b, c, th_seed=ts, th_cell=tc)
e1.record(); torch.cuda.synchronize()
ms_sep = e0.elapsed_time(e1) / reps
print(f"sweep of 8 threshold pairs: {ms_sweep:.1f} ms in one call, {ms_sep:.1f} ms as 8 calls  ({8*H*W/ms_sweep/1e3:.0f} Mpx/s)")
m0 = lab8[2].contiguous(); m1 = lab8[5].contiguous()
su.aji_plus_masks(m0, m1); torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    aji = su.aji_plus_masks(m0, m1)
dt = (time.time() - t0) / reps
a, bb = m0.cpu().numpy().view(np.uint16), m1.cpu().numpy().view(np.uint16)
t0 = time.time(); want_aji = eval_ref.score_pair(a, bb); dt_cpu = time.time() - t0
print(f"AJI+ of two {H}x{W} masks ({int(a.max())} / {int(bb.max())} instances): HIP+host pairing {dt*1e3:.1f} ms, numpy/scipy oracle {dt_cpu*1e3:.0f} ms, "
      f"equal: {abs(aji - want_aji) < 1e-12}")
