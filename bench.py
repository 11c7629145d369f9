#!/usr/bin/env python3
"""bench.py — headline benchmark of the microbeSEG hot path on MI355X: training crops/sec at 256x256.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1 via torch.distributed.run, RCCL)
One "step" = zero_grad + forward + loss + backward (+ gradient all-reduce) + optimizer step on one per-GPU batch of
synthetic crops that already live in HBM.  Workload at N=1 = BASELINE.json configs[1]: distance-method DU-Net
`[64,1024]`, 256x256 crops, batch 32, fp32 (reference: train.py:184-194 + train_script.py defaults; ReLU/BatchNorm/
Adam-amsgrad = the reference's Adam configuration, train.py:174,380-385).
Prints ONE JSON line on rank 0 with `roofline` (dominant MFMA kernel, HIP-event timed inside the timed region) and
`cpu_baseline` (the CPU oracle = port of the reference's torch path, bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, no xf32 on gfx950
PEAK_BF16_MFMA_TFLOPS = 2516.6  # same guide: dense bf16 MFMA = 16 x the fp32 matrix rate (~2.5 PF)
# forward GFLOP per 256x256 crop, measured by hooking the reference modules (SURVEY.md §8d / BASELINE.md §2)
FWD_GFLOP_256 = {("DU", (64, 1024)): 163.30, ("U", (64, 1024)): 101.03}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--arch", default="DU", choices=["DU", "U"])
    ap.add_argument("--act", default="relu")
    ap.add_argument("--norm", default="bn")
    ap.add_argument("--filters", type=int, nargs=2, default=[64, 1024])
    ap.add_argument("--optimizer", default="adam", choices=["adam", "ranger"])
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE configs[2]: bf16 matrix-core kernels for every convolution that has one, "
                         "activations and activation gradients STORED as bf16 (DESIGN.md 4b), fp32 accumulate / weights / "
                         "normalisation statistics (the default line is the fp32 configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event bracketing of MFMA kernels")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (training/graph_step.py; 1 GPU, Adam; implies --no-kernel-timing): "
                         "for small batches / bf16, where the host cannot enqueue ~600 launches per step fast enough")
    ap.add_argument("--no-inference", action="store_true", help="skip the secondary inference (Mpixels/s) measurement")
    ap.add_argument("--train-only", action="store_true",
                    help="(internal) the training blocks (+ cpu_baseline) only: the default single-GPU run measures them in a "
                         "child process of their own")
    ap.add_argument("--inference-only", action="store_true",
                    help="(internal) run only the inference block and print {\"inference\": ...}: the default run measures it in "
                         "a child process of its own, started before the parent touches the GPU")
    ap.add_argument("--infer-size", type=int, default=2048)
    ap.add_argument("--infer-frames", type=int, default=16, help="T of the 2D+t stack (SURVEY.md §8d: T = 16)")
    ap.add_argument("--no-bf16-block", action="store_true",
                    help="skip the secondary BASELINE configs[2] measurement (320x320, bf16) of the default run")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0 = pick)")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of the multi-rank contract (gloo, no kernels): launcher, rendezvous, barrier + "
                         "max-over-ranks timing and the JSON line, with a sleep as the step; `value` is meaningless")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a rank environment: start the N ranks ourselves, one process per GPU, as
    fresh children of a process that has not touched the GPU (never re-exec after HIP is initialised), exactly the
    way the driver does it: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>.
    Rank 0's JSON line goes to our stdout; our exit code is the launcher's."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def inference_child(args):
    """The inference block in a process of its own, run to completion BEFORE this process touches the GPU (a child of a
    process without a HIP context, like launch_ranks).  Blocks that share one process disturb each other through the
    allocator: the 2048^2 frames of the inference block and the batch-32 tensors of the training blocks leave each other an
    address space cut into pieces, and whichever block ran second measured 4 % (bf16 training step, 40.63 vs 39.02 ms) to
    9 % (bf16 product-path inference, 258 vs 283 Mpx/s) slower than on its own — same box, same kernels.  Returns the
    block's dict, or None when the child failed (the caller then measures in-process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--inference-only", "--arch", args.arch, "--act", args.act,
           "--norm", args.norm, "--filters", str(args.filters[0]), str(args.filters[1]),
           "--infer-size", str(args.infer_size), "--infer-frames", str(args.infer_frames), "--precision", args.precision]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line).get("inference")
    except Exception:
        pass
    return None


def run_in_children(args):
    """Default single-GPU run: the training blocks and the inference block each in a fresh process, the training blocks
    FIRST — on a chip that has not just run 30 s of inference (the fp32 matrix kernels hold 2.33-2.37 GHz on a cool chip
    and the headline step measured 131.1-131.7 ms at the start of a process tree, 133.8-135.0 ms right behind the inference
    block) — and neither disturbed by the other's allocations.  This process never touches the GPU.  Returns the merged
    dict, or None when the training child failed (the caller then runs everything in-process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--train-only"] + [a for a in sys.argv[1:] if a != "--train-only"]
    out = None
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, timeout=3000)
        for line in reversed(r.stdout.splitlines()):
            if line.startswith("{"):
                out = json.loads(line)
                break
    except Exception:
        out = None
    if out is None:
        return None
    if not args.no_inference and args.arch == "DU":
        inf = inference_child(args)
        if inf is not None:
            out["inference"] = inf
    return out


def synthetic_batch(batch, size, arch, seed, device):
    """Synthetic crops of the training tensor contract (training_dataset.py:30-63 + mytransforms.py:380-406):
    img fp32 in [-1,1] (1,H,W); distance labels fp32 in [0,1]; boundary labels int64 in {0,1,2}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 1, size, size, generator=g) * 2 - 1
    if arch == "DU":
        l1 = torch.rand(batch, 1, size, size, generator=g)
        l2 = torch.rand(batch, 1, size, size, generator=g)
    else:
        l1 = torch.randint(0, 3, (batch, size, size), generator=g)
        l2 = None
    return tuple(t.to(device) if t is not None else None for t in (img, l1, l2))


def host_cores():
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota (containers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_train_steps(arch, act, norm, filters, size, batch, budget_s, max_steps, cores):
    """a few training steps of one configuration through oracle/unet_ref.py on the host cores -> (crops/s, steps)"""
    from oracle import unet_ref
    from microbeseg_amd.utils.unets import build_unet
    torch.manual_seed(0)
    holder = build_unet(arch, act, "conv", norm, "cpu", 1, ch_out=3 if arch == "U" else 1, filters=filters)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in holder.state_dict().items()}
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=8e-4, betas=(0.9, 0.999), eps=1e-8,
                           amsgrad=True)
    img, l1, l2 = synthetic_batch(batch, size, arch, 99, "cpu")

    def step(train=True):
        opt.zero_grad()
        with torch.set_grad_enabled(train):
            out = unet_ref.unet_forward(sd, img, arch, act, norm, filters, training=train, update_running_stats=train)
            if arch == "DU":
                loss = unet_ref.regression_loss(out[0], l1) + unet_ref.regression_loss(out[1], l2)
            else:
                loss = unet_ref.ce_dice(out, l1)
            if train:
                loss.backward()
                opt.step()
        return loss.item()

    step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < max_steps):
        step()
        n += 1
    return batch * n / (time.perf_counter() - t0), n, step


def cpu_baseline(args, filters):
    """The reference's CPU path (torch CPU fp32, all host cores) restated by oracle/unet_ref.py + oracle/postproc_ref.c,
    timed on the GPU box's host next to the GPU numbers.  Baseline only — never what is shipped or
    measured as `value`.  Headline entry = this run's workload at batch 2; `configs` holds the other BASELINE shapes:
    configs[0] in full (2-level [8,16] net, 32 train + 8 val crops, batch 2, one epoch), the configs[2] shape (320x320),
    and configs[4] (one 2048x2048 frame through the CPU network and the single-threaded C post-processing)."""
    import numpy as np
    from oracle import postproc_ref, unet_ref
    from microbeseg_amd.utils import synth
    from microbeseg_amd.utils.unets import build_unet
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    rate, n, _ = _cpu_train_steps(args.arch, args.act, args.norm, filters, args.size, 2, 15.0, 8, cores)
    out = {"value": round(rate, 4), "unit": "crops/s", "cores": cores, "kind": "port",
           "sample": f"{n} training steps of batch 2 ({args.size}x{args.size}) through oracle/unet_ref.py "
                     f"(torch CPU fp32, {cores} threads), same net/loss/Adam-amsgrad", "configs": {}}
    if args.arch != "DU" or tuple(filters) != (64, 1024):
        return out
    # configs[0] in full: one epoch = 16 training + 4 validation steps of batch 2 on 256x256 crops, [8,16] DU-Net
    _, _, step0 = _cpu_train_steps("DU", args.act, args.norm, (8, 16), 256, 2, 0.0, 1, cores)
    t0 = time.perf_counter()
    for _ in range(16):
        step0(True)
    for _ in range(4):
        step0(False)
    dt = time.perf_counter() - t0
    out["configs"]["configs[0]"] = {"value": round(40 / dt, 2), "unit": "crops/s", "epoch_s": round(dt, 3),
                                    "sample": "full: DU-Net [8,16], 32 train + 8 val 256x256 crops, batch 2, 1 epoch"}
    rate3, n3, _ = _cpu_train_steps("DU", args.act, args.norm, (64, 1024), 320, 2, 8.0, 3, cores)
    out["configs"]["configs[2] shape"] = {"value": round(rate3, 4), "unit": "crops/s",
                                          "sample": f"{n3} fp32 training steps of batch 2 at 320x320 (the CPU path has "
                                                    f"no bf16 mode)"}
    # configs[4]: 2048x2048 frame, network forward (eval) on the host cores + the C oracle of the post-processing
    torch.manual_seed(0)
    holder = build_unet("DU", args.act, "conv", args.norm, "cpu", 1, ch_out=1, filters=(64, 1024))
    sd = {k: v.detach() for k, v in holder.state_dict().items()}
    S = args.infer_size
    frame = torch.rand(1, 1, S, S) * 2 - 1
    t0 = time.perf_counter()
    with torch.no_grad():
        unet_ref.unet_forward(sd, frame, "DU", args.act, args.norm, (64, 1024), training=False)
    t_net = time.perf_counter() - t0
    rng = np.random.Generator(np.random.PCG64(2024))
    cell, border = synth.synth_prediction_maps(rng, S, S, int(2500 * (S / 2048.0) ** 2), rmin=5.0, rmax=13.0)
    t0 = time.perf_counter()
    postproc_ref.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10)
    t_pp = time.perf_counter() - t0
    out["configs"]["configs[4]"] = {"value": round(S * S / (t_net + t_pp) / 1e6, 3), "unit": "Mpx/s",
                                    "net_s": round(t_net, 2), "postproc_s": round(t_pp, 3),
                                    "sample": f"1 frame {S}x{S}: DU-Net [64,1024] forward on {cores} threads + C oracle of "
                                              f"the post-processing on 1 thread (scipy / scikit-image are single-threaded)"}
    return out


def _boundary_probs(cell, border):
    """softmax-like (H, W, 3) probabilities of the boundary method from synthetic cell / border maps"""
    import numpy as np
    p1 = np.clip(cell * 2.0, 0, 1) * (1 - np.clip(border * 1.2, 0, 1))
    p2 = np.clip(border * 1.2, 0, 1) * (cell > 0.02)
    p0 = np.clip(1 - p1 - p2, 0.0, 1)
    probs = np.stack([p0, p1, p2], -1).astype(np.float32)
    return probs / probs.sum(-1, keepdims=True)


def inference_metric(args, net, dev):
    """Secondary metric of BASELINE.json: inference Mpixels/s including the watershed, 1 GPU, configs[4] shape
    (2048x2048 2D+t stack, DU-Net [64,1024]).

    `value` = wall clock of the PRODUCT path: `InferWorker.infer_stack` (infer_script_local.py:118-161) on a
    [T, S, S] uint16 stack in host memory with a checkpoint written to a temporary directory — staging into pinned memory,
    upload, min / max + normalisation + padding (on the device, K14), network, watershed on the side stream, download of
    the masks; the same again with the opt-in bf16 network (`bf16`).  An untrained network predicts no distance maps, so the
    two thresholds are taken from the distribution of its own output on frame 0 (~15 % cell pixels, ~4 % seed pixels:
    thousands of instances per frame, `instances`).
    Next to it, on tensors resident in HBM: `net_ms` (one frame through the network), `postproc_ms` on synthetic prediction
    maps with ~2500 cells (labels checked bit for bit against the C oracle), `boundary_postproc` (the boundary method's
    post-processing, reference postprocessing.py:62-90) and the C oracle on one host core."""
    import json as _json
    import tempfile
    import numpy as np
    from microbeseg_amd import engine
    from microbeseg_amd.inference import postprocessing as pp
    from microbeseg_amd.inference.infer import InferWorker
    from microbeseg_amd.utils import synth
    from oracle import postproc_ref
    S, T = args.infer_size, args.infer_frames
    rng = np.random.Generator(np.random.PCG64(2024))
    # ~2500 cells / 2048^2 frame, semi-axes U[5,13] px -> ~15 % foreground, touching cells merge
    cell, border = synth.synth_prediction_maps(rng, S, S, int(2500 * (S / 2048.0) ** 2), rmin=5.0, rmax=13.0)
    c, b = torch.from_numpy(cell).to(dev), torch.from_numpy(border).to(dev)
    frame = (torch.rand(1, 1, S, S, device=dev) * 2 - 1)
    net.eval()
    n_res = min(T, 8)
    net_ms = {}
    with torch.no_grad():
        for prec in ("fp32", "bf16"):
            with engine.precision_scope(prec):
                net(frame)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n_res):
                    net(frame)
                e1.record()
                torch.cuda.synchronize()
                net_ms[prec] = e0.elapsed_time(e1) / n_res
        pp.distance_postprocessing_device(b, c, 0.45, 0.10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_res):
            labels, n_inst, status = pp.distance_postprocessing_device(b, c, 0.45, 0.10)
        e1.record()
        torch.cuda.synchronize()
        t_pp = e0.elapsed_time(e1) / n_res
        probs = torch.from_numpy(_boundary_probs(cell, border)).to(dev)
        pp.boundary_postprocessing_device(probs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            bl, bn, bs = pp.boundary_postprocessing_device(probs)
        e1.record()
        torch.cuda.synchronize()
        t_bpp = e0.elapsed_time(e1) / 3
        # the floods of 8 frames in one launch (one workgroup per frame): what a frame of a STACK costs
        pp.boundary_postprocessing_batch_device([probs] * 8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            outs8 = pp.boundary_postprocessing_batch_device([probs] * 8)
        e1.record()
        torch.cuda.synchronize()
        t_bpp8 = e0.elapsed_time(e1) / 3 / 8
        bpp8_equal = all(bool(torch.equal(o[0], bl)) for o in outs8)
    t0 = time.perf_counter()
    want = postproc_ref.distance_postprocessing(border[..., None], cell[..., None], 0.45, 0.10)
    t_cpu = time.perf_counter() - t0
    exact = bool(np.array_equal(labels.cpu().numpy().view(np.uint16), want))
    with torch.no_grad():                                # what the worker must return for these maps (its id order)
        labels_cm = pp.distance_postprocessing_device(b, c, th_seed=0.45, th_cell=0.10, col_major_ids=True)[0]
    labels_cm = labels_cm.cpu().numpy().view(np.uint16)
    want_b = postproc_ref.boundary_postprocessing(probs.cpu().numpy())
    exact_b = bool(np.array_equal(bl.cpu().numpy().view(np.uint16), want_b))

    # ---- the product path ------------------------------------------------------------------------------------------------
    frames = []
    for _ in range(2):                                   # two distinct synthetic frames, repeated over the stack
        cm, _ = synth.synth_prediction_maps(rng, S, S, int(2500 * (S / 2048.0) ** 2), rmin=5.0, rmax=13.0)
        # (a camera offset keeps the background noise off the clip at 0: flat input regions would give an untrained network
        # flat, i.e. exactly tied, predictions — the watershed's tie-handling paths, not its normal load)
        frames.append(np.clip(cm * 50000 + 3000 + rng.normal(0, 800, cm.shape), 0, 65535).astype(np.uint16))
    stack = np.stack([frames[t & 1] for t in range(T)])
    product = {}
    with tempfile.TemporaryDirectory() as tmp:
        base = os.path.join(tmp, "distance_model_00")
        torch.save(net.state_dict(), base + ".pth")
        with open(base + ".json", "w") as f:
            _json.dump({"architecture": [args.arch, "conv", args.act, args.norm, list(args.filters)], "label_type": "distance"}, f)
        worker = InferWorker(model=base, device=str(dev), ths=(0.10, 0.45))
        # The network of the offline bench is untrained: its "distance maps" threshold to one confluent blob, i.e. a flood that
        # is a single sequential component (seconds per frame) — not the load of a trained model.  The worker's measurement
        # hook swaps the network's OUTPUT for the synthetic prediction maps (~2500 cells) right before the post-processing;
        # every other stage (staging, upload, device normalisation, the full network forward, watershed, download) runs as
        # in production.  `untrained_predictions` reports the same path without the hook on two frames.
        c4, b4 = c[None, None].contiguous(), b[None, None].contiguous()
        pad_cache = {}

        def hook(pred):
            shp = tuple(pred[0].shape)
            if shp not in pad_cache:                     # the prediction of a padded frame: pad the maps at the top / left alike
                ph, pw = shp[2] - S, shp[3] - S
                pad_cache[shp] = (torch.nn.functional.pad(b4, (pw, 0, ph, 0)), torch.nn.functional.pad(c4, (pw, 0, ph, 0)))
            return pad_cache[shp]
        for prec in ("fp32", "bf16"):
            worker.precision = prec
            worker.prediction_hook = hook
            worker.infer_stack(stack[:2])                # warm-up (buffers, weight packs of this precision)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            masks = worker.infer_stack(stack)
            dt = time.perf_counter() - t0
            product[prec] = {"value": round(T * S * S / dt / 1e6, 2), "unit": "Mpx/s", "ms_per_frame": round(1e3 * dt / T, 2),
                             "instances_frame0": int(masks[0].max()), "net_ms": round(net_ms[prec], 2),
                             "masks_equal_resident_run": bool(np.array_equal(masks[0], labels_cm))}
        worker.prediction_hook = None
        worker.precision = args.precision
        f0 = stack[0]
        with torch.no_grad():
            x0 = torch.from_numpy((2 * (f0.astype(np.float32) - f0.min()) / (f0.max() - f0.min()) - 1)[None, None]).to(dev)
            bo, ce = worker.net(x0)
            bb = torch.tan(bo[0, 0].clamp(0, 1) ** 2)
            bb = torch.where(bb < 0.05, torch.zeros_like(bb), bb).clamp(0, 1)
            sub = slice(None, None, 7)
            worker.ths = [float(torch.quantile(ce[0, 0][sub, sub].flatten(), 0.85)),
                          float(torch.quantile((ce[0, 0] - bb)[sub, sub].flatten(), 0.96))]
            del x0, bo, ce, bb
        t0 = time.perf_counter()
        m2 = worker.infer_stack(stack[:2])
        dt2 = time.perf_counter() - t0
        untrained = {"ms_per_frame": round(1e3 * dt2 / 2, 1), "instances_frame0": int(m2[0].max()), "frames": 2,
                     "note": "no hook: the untrained network's own output, thresholds at its 85 % / 96 % quantiles"}
        del worker
    # ---- the same product path for the boundary method (U-Net, 3 classes): its flood is ONE wavefront busy for ~45 ms per
    # frame, so infer_stack collects InferWorker.BOUNDARY_BATCH frames and launches their floods together (one workgroup per
    # frame); `*_batch_1` is the same stack with one frame per launch ------------------------------------------------------
    boundary_stack = None
    try:
        from microbeseg_amd.utils.unets import build_unet
        torch.manual_seed(5)
        net_u = build_unet("U", args.act, "conv", args.norm, dev, 1, ch_out=3, filters=tuple(args.filters))
        logits_s = torch.log(probs.permute(2, 0, 1)[None].contiguous() + 1e-6)      # (1, 3, S, S): the synthetic classes
        lcache = {}

        def hook_u(pred):
            shp = tuple(pred.shape)
            if shp not in lcache:
                ph, pw = shp[2] - S, shp[3] - S
                lcache[shp] = torch.nn.functional.pad(logits_s, (pw, 0, ph, 0), value=0.0)
            return lcache[shp]
        with tempfile.TemporaryDirectory() as tmp:
            base = os.path.join(tmp, "boundary_model_00")
            torch.save(net_u.state_dict(), base + ".pth")
            with open(base + ".json", "w") as f:
                _json.dump({"architecture": ["U", "conv", args.act, args.norm, list(args.filters)], "label_type": "boundary"}, f)
            wu = InferWorker(model=base, device=str(dev))
            wu.prediction_hook = hook_u
            boundary_stack = {"unit": "Mpx/s", "frames": T, "frames_per_flood_launch": int(wu.BOUNDARY_BATCH)}
            for prec in ("fp32", "bf16"):
                wu.precision = prec
                for ns in (wu.BOUNDARY_BATCH, 1):
                    wu.BOUNDARY_BATCH = ns
                    wu.infer_stack(stack[:2 * ns])           # warm-up: both groups' workspaces exist
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    mu = wu.infer_stack(stack)
                    dtu = time.perf_counter() - t0
                    key = prec if ns > 1 else prec + "_batch_1"
                    boundary_stack[key] = {"value": round(T * S * S / dtu / 1e6, 2), "ms_per_frame": round(1e3 * dtu / T, 2),
                                           "instances_frame0": int(mu[0].max())}
                wu.BOUNDARY_BATCH = boundary_stack["frames_per_flood_launch"]
                with torch.no_grad():                    # one frame alone, on the main stream
                    single = wu._postprocess(logits_s, [0, 0]).cpu().numpy().view(np.uint16)
                boundary_stack[prec]["masks_equal_single_call"] = bool(np.array_equal(mu[0], single) and
                                                                       np.array_equal(mu[T - 1], single))
            del wu
        del net_u
    except Exception as err:                               # (secondary measurement: never costs the headline line)
        boundary_stack = {"error": repr(err)[:200]}
    net.train()
    main = product[args.precision]
    return {"metric": "inference Mpixels/sec incl. watershed", "value": main["value"],
            "unit": "Mpx/s", "frame": f"{S}x{S}", "frames": T, "stack_ms_per_frame": main["ms_per_frame"],
            "path": "InferWorker.infer_stack on a host uint16 stack (staging, H2D, device min/max + normalise + pad, network, "
                    "watershed, D2H of the masks); the untrained network's output is replaced by synthetic prediction maps "
                    "(~2500 cells) right before the post-processing (InferWorker.prediction_hook)",
            "instances_frame0": main["instances_frame0"],
            "serial_Mpx_s": round(S * S / (net_ms[args.precision] + t_pp) / 1e3, 2), "net_ms": round(net_ms[args.precision], 2),
            "postproc_ms": round(t_pp, 2),
            "net_precision": args.precision,
            "bf16": product["bf16"], "fp32": product["fp32"], "untrained_predictions": untrained,
            "postproc_Mpx_s": round(S * S / t_pp / 1e3, 1), "instances": int(n_inst), "postproc_status": int(status),
            "labels_bit_exact_vs_oracle": exact,
            "boundary_stack": boundary_stack,
            "boundary_postproc_batch8": {"value": round(S * S / t_bpp8 / 1e3, 1), "unit": "Mpx/s", "ms_per_frame": round(t_bpp8, 2),
                                         "labels_equal_single_frame": bpp8_equal},
            "boundary_postproc": {"value": round(S * S / t_bpp / 1e3, 1), "unit": "Mpx/s", "ms": round(t_bpp, 2),
                                  "instances": int(bn), "status": int(bs), "labels_bit_exact_vs_oracle": exact_b},
            "cpu_postproc": {"value": round(S * S / t_cpu / 1e6, 2), "unit": "Mpx/s", "cores": 1, "kind": "port",
                             "sample": "1 frame through oracle/postproc_ref.c (single thread, like scipy/skimage)"}}


def workload_key(args):
    return {"batch": args.batch, "size": args.size, "arch": args.arch, "act": args.act, "norm": args.norm,
            "filters": list(args.filters), "optimizer": args.optimizer, "precision": args.precision}


# profiles written before the traffic files carried their own `__config__` (tools/pmc_traffic.sh now stores it)
_LEGACY_PROFILE_CONFIG = {
    "r01f": {"batch": 32, "size": 256, "arch": "DU", "act": "relu", "norm": "bn", "filters": [64, 1024],
             "optimizer": "adam", "precision": "fp32"},
    "r01g": {"batch": 32, "size": 256, "arch": "DU", "act": "relu", "norm": "bn", "filters": [64, 1024],
             "optimizer": "adam", "precision": "fp32"},
    "r01h_bf16_320": {"batch": 32, "size": 320, "arch": "DU", "act": "relu", "norm": "bn", "filters": [64, 1024],
                      "optimizer": "adam", "precision": "bf16"},
}


def csrc_digest():
    """sha1 over the kernel sources: a committed counter profile describes the kernels of exactly one such tree"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "microbeseg_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "microbeseg_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the newest committed TCC-counter profile OF THIS WORKLOAD
    (tools/pmc_traffic.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied; the
    profile records the bench configuration it was taken on and only a profile of the same batch / size / net /
    precision is used).  bench.py cannot collect PMC counters itself; the profile is part of the repo."""
    import glob
    want = workload_key(args)
    sha = csrc_digest()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc.json")), reverse=True)
    stale = None
    for f in files:
        tab = json.load(open(f))
        cfg = tab.get("__config__")
        if cfg is None:
            cfg = _LEGACY_PROFILE_CONFIG.get(os.path.basename(f)[:-len("_hbm_traffic_pmc.json")])
        cfg = dict(cfg or {})
        prof_sha = cfg.pop("csrc_sha", None)
        if cfg != want:
            continue
        ent = tab.get(kernel)
        if ent is None:
            continue
        if prof_sha != sha:                              # measured on other kernel sources: not presented as current
            stale = stale or os.path.basename(f)
            continue
        return {"traffic": int(ent["hbm_bytes_per_launch"]), "traffic_unit": "bytes/launch (avg)",
                "traffic_source": "profiles/" + os.path.basename(f), "traffic_kernel": kernel}
    if stale:
        return {"traffic": None, "traffic_source": "stale", "traffic_stale_profile": "profiles/" + stale}
    return {"traffic": None, "traffic_source": "none"}


def rehearse(args, world, rank):
    """CPU rehearsal (gloo): everything of the N-rank contract except the kernels."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "training crops/sec (256px)", "value": 0.0, "unit": "crops/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "rehearsal": True,
                          "data": "none", "config": {"workload": "launcher rehearsal on CPU (gloo), no kernels",
                                                     "global_batch": args.batch * world, "parallelism": f"dp{world}"}}),
              flush=True)


def train_metric(args, dev, world, rank, timing=True):
    """W untimed + K timed training steps of the workload `args` names; returns (JSON dict, net)."""
    import torch.distributed as dist
    from microbeseg_amd import engine
    from microbeseg_amd.utils.unets import build_unet
    from microbeseg_amd.training.losses import get_loss

    filters = tuple(args.filters)
    engine.set_precision(args.precision)
    torch.manual_seed(0)
    net = build_unet(args.arch, args.act, "conv", args.norm, dev, world if world > 1 else 1,
                     ch_out=3 if args.arch == "U" else 1, filters=filters)
    graph = args.graph
    if graph and (world > 1 or args.optimizer != "adam"):
        raise SystemExit("--graph: single GPU and Adam only")
    if args.optimizer == "adam":
        from microbeseg_amd.training.optim import make_adam
        opt = make_adam(net.parameters(), capturable=graph)
    else:
        from microbeseg_amd.training.ranger2020 import Ranger
        opt = Ranger(net.parameters(), lr=6e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-6)
    crit = get_loss("smooth_l1", "distance") if args.arch == "DU" else get_loss("ce_dice", "boundary")
    batches = [synthetic_batch(args.batch, args.size, args.arch, 1234 + 17 * rank + i, dev) for i in range(2)]
    net.train()

    def one_step(img, l1, l2):
        opt.zero_grad()
        if args.arch == "DU":
            border, cell = net(img)
            loss = crit["border"](border, l1) + crit["cell"](cell, l2)
        else:
            loss = crit(net(img), l1)
        loss.backward()
        opt.step()
        return loss

    warmup = args.warmup
    eager_step = one_step
    if graph:
        from microbeseg_amd.training.graph_step import GraphedTrainStep
        one_step = GraphedTrainStep(one_step, opt, warmup=2)
        warmup = max(warmup, 3)             # two eager calls + the capture happen before the timed region

    def step(i):
        return one_step(*batches[i % len(batches)])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    # the ~1300 event records + dispatch queries of the kernel bracketing cost a few ms per step (bf16 steps of 30-50 ms
    # become host-bound): `value` comes from K un-bracketed steps and the kernel table / roofline from K more, bracketed
    # steps
    two_pass = timing
    prof = engine.KernelTimer() if timing else None
    engine.set_kernel_timer(None if two_pass else prof)
    fence()
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_host = time.perf_counter() - t0        # the host's share: all K steps enqueued (it may have waited for queue space)
    fence()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # hipMalloc / hipFree calls of the caching allocator inside the timed region (each one synchronises the device: a
    # steady-state step makes none)
    alloc_calls = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    engine.set_kernel_timer(None)
    dt_bracketed = None
    if two_pass:
        engine.set_kernel_timer(prof)
        overlap, dec_overlap = engine.get_wgrad_overlap(), engine.get_decoder_overlap()
        engine.set_decoder_overlap(False)
        engine.set_wgrad_overlap(False)      # one kernel at a time under the brackets: an event pair around a launch that
        fence()                              # shares the chip with another stream's kernel would time both
        t0 = time.perf_counter()
        for i in range(args.steps):
            eager_step(*batches[i % len(batches)])       # (a replayed graph has no launches to bracket)
        fence()
        dt_bracketed = time.perf_counter() - t0
        engine.set_kernel_timer(None)
        engine.set_wgrad_overlap(overlap)
        engine.set_decoder_overlap(dec_overlap)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    crops = args.batch * world * args.steps
    value = crops / dt
    fwd_gflop = FWD_GFLOP_256.get((args.arch, filters))
    cfg_no = 1 if args.precision == "fp32" else 2
    if world > 1:
        cfg_no = 3
    prec_txt = "fp32" if args.precision == "fp32" else \
        "bf16 forward/backward (bf16 matrix-core operands, fp32 accumulate and norm statistics)"
    if args.arch == "DU":
        wl = (f"BASELINE configs[{cfg_no}]: {args.arch}-Net distance-map training step, filters {list(filters)}, "
              f"{args.size}x{args.size} crops, per-GPU batch {args.batch}, {prec_txt}, {args.act}/{args.norm}, "
              f"SmoothL1x2 + Adam(amsgrad)")
    else:
        wl = (f"{args.arch}-Net boundary training step, filters {list(filters)}, {args.size}px, batch {args.batch}, "
              f"{prec_txt}, ce_dice + Adam")
    out = {
        "metric": "training crops/sec (256px)" if args.size == 256 else f"training crops/sec ({args.size}px)",
        "value": round(value, 3), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": wl, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                   "launch": "hipGraph replay" if graph else "eager"},
        "per_gpu_value": round(value / world, 3),
        "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 3),
        "allocator_calls_in_timed_region": alloc_calls,
    }
    if world > 1:
        out["config"]["collective"] = "RCCL all-reduce of fp32 gradient buckets, overlapped with the backward"
    if fwd_gflop is not None:
        train_tflop_per_crop = 3.0 * fwd_gflop * (args.size / 256.0) ** 2 / 1e3
        out["model_tflops_per_gpu"] = round(value / world * train_tflop_per_crop, 2)
        if args.precision == "fp32":
            out["frac_of_fp32_mfma_peak"] = round(value / world * train_tflop_per_crop / PEAK_FP32_MFMA_TFLOPS, 4)
        else:
            out["frac_of_bf16_mfma_peak"] = round(value / world * train_tflop_per_crop / PEAK_BF16_MFMA_TFLOPS, 4)
    if dt_bracketed is not None:
        out["kernel_timing"] = ("separate bracketed pass of %d steps (%.3f ms/step with HIP-event bracketing, weight "
                                "gradients on the main stream: one kernel at a time)" % (
                                    args.steps, 1e3 * dt_bracketed / args.steps))
    if prof is not None:
        kernels = prof.summary()
        if kernels:
            # the dominant kernel FAMILY (all template instantiations of one kernel: they are one piece of code with
            # compile-time branches, and a layer stack spreads over several of them)
            fams = {}
            for k, v in kernels.items():
                fam = fams.setdefault(k.split("<")[0], {"family": k.split("<")[0], "total_ms": 0.0, "flops": 0.0, "launches": 0,
                                                         "members": []})
                fam["total_ms"] += v["total_ms"]
                fam["flops"] += v["flops"]
                fam["launches"] += v["launches"]
                fam["members"].append(k)
            dom = max(fams.values(), key=lambda f: f["total_ms"])
            dom_tf = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
            big = max(dom["members"], key=lambda k: kernels[k]["total_ms"])
            peak = PEAK_BF16_MFMA_TFLOPS if ("bf16" in dom["family"] or "p8" in dom["family"]) else PEAK_FP32_MFMA_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": dom["family"], "instantiations": sorted(dom["members"]),
                               "achieved": round(dom_tf, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(dom_tf / peak, 4), "traffic": None,
                               "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4), "launches": dom["launches"],
                               "ms_per_step": round(dom["total_ms"] / args.steps, 3)}
            out["roofline"].update(pmc_traffic(big, args))
            out["kernel_families"] = {f["family"]: {"tflops": round(f["flops"] / (f["total_ms"] * 1e-3) / 1e12, 2),
                                                     "total_ms_per_step": round(f["total_ms"] / args.steps, 3),
                                                     "launches_per_step": f["launches"] // args.steps}
                                      for f in sorted(fams.values(), key=lambda f: -f["total_ms"])}
            out["kernels"] = {k: {"tflops": round(v["tflops"], 2), "total_ms_per_step": round(v["total_ms"] / args.steps, 3),
                                  "avg_launch_ms": round(v["avg_ms"], 4), "launches_per_step": v["launches"] // args.steps}
                              for k, v in kernels.items()}
    del opt, batches
    for p_ in net.parameters():
        p_.grad = None
    return out, net


def release(*objs):
    """Drop a finished block's network for good before the next block is timed: the packed-weight caches and the optimizer's
    arenas sit in reference cycles (param -> pack -> param), so `del` alone leaves gigabytes to the cyclic collector — which
    then runs, and returns them to the allocator, somewhere inside a later block's timed loop."""
    import gc
    del objs
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def main():
    args = parse()
    in_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_rank_env:
        sys.exit(launch_ranks(args))          # nothing has touched the GPU in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.rehearse:
        return rehearse(args, world, rank)
    if world == 1 and not in_rank_env and not args.train_only and not args.inference_only:
        merged = run_in_children(args)
        if merged is not None:
            print(json.dumps(merged), flush=True)
            return
    want_inference = rank == 0 and world == 1 and not args.no_inference and args.arch == "DU" and not args.train_only
    inference = inference_child(args) if (want_inference and not args.inference_only) else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # one rank per GPU.  MSEG_BENCH_BACKEND=gloo is a single-GPU rehearsal of the N-rank path (ranks share the visible
    # GPUs round-robin, collectives go through gloo) — RCCL itself needs one GPU per rank
    backend = os.environ.get("MSEG_BENCH_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ngpu:
        raise SystemExit(f"bench.py: rank {rank} has no GPU ({ngpu} visible); RCCL needs one GPU per rank")
    dev_index = local_rank % max(ngpu, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from microbeseg_amd import engine
    if args.inference_only:
        from microbeseg_amd.utils.unets import build_unet
        engine.set_precision(args.precision)
        torch.manual_seed(3)
        net_i = build_unet(args.arch, args.act, "conv", args.norm, dev, 1, ch_out=1, filters=tuple(args.filters))
        print(json.dumps({"inference": inference_metric(args, net_i, dev)}), flush=True)
        return
    out, net = train_metric(args, dev, world, rank, timing=not args.no_kernel_timing)
    headline = workload_key(args) == {"batch": 32, "size": 256, "arch": "DU", "act": "relu", "norm": "bn",
                                      "filters": [64, 1024], "optimizer": "adam", "precision": "fp32"}
    del net
    release()
    if rank == 0 and world == 1 and headline and not args.no_bf16_block:
        # BASELINE configs[2] next to the fp32 headline, so that the driver's default run records it too
        import copy
        a2 = copy.copy(args)
        a2.precision, a2.size, a2.graph = "bf16", 320, False
        a2.steps, a2.warmup = max(args.steps, 10), max(args.warmup, 3)
        blk, net2 = train_metric(a2, dev, 1, 0, timing=not args.no_kernel_timing)   # two-pass timing, see train_metric
        del net2
        release()
        engine.set_precision(args.precision)
        out["bf16_320"] = blk
        # further configurations of the path, each a short run with its own roofline: the boundary method's
        # U-Net with ce_dice, GroupNorm (the north-star's norm), the GUI / CLI default batch of 4
        out["secondary"] = {}
        for tag, kw in (("unet_ce_dice_bf16_256", dict(arch="U", precision="bf16")),
                        ("groupnorm_bf16_320", dict(norm="gn", precision="bf16", size=320)),
                        ("batch4_fp32_256", dict(batch=4, steps=20)),
                        ("batch4_bf16_256", dict(batch=4, precision="bf16", steps=20))):
            a3 = copy.copy(args)
            a3.graph = False
            a3.steps, a3.warmup = 6, 3
            for k_, v_ in kw.items():
                setattr(a3, k_, v_)
            blk, net3 = train_metric(a3, dev, 1, 0, timing=not args.no_kernel_timing)
            del net3
            release()
            blk.pop("kernels", None)                     # the per-instantiation table stays with the two main blocks
            if a3.batch == 4:
                # batch 4 (the GUI / CLI default), ~600 launches per step: the same step replayed from a hipGraph
                # (training/graph_step.py, TrainWorker.graph_steps) — what the recording buys is host time
                a4 = copy.copy(a3)
                a4.graph = True
                rep, net4 = train_metric(a4, dev, 1, 0, timing=False)
                del net4
                release()
                blk["graph_replay"] = {k: rep[k] for k in ("value", "ms_per_step", "host_enqueue_ms_per_step")}
            out["secondary"][tag] = blk
        engine.set_precision(args.precision)
    # (the inference block was measured first, in a child process of its own: inference_child)
    if inference is not None:
        out["inference"] = inference
    elif want_inference:
        # (in-process fallback)
        # a freshly initialised network (fixed seed): the few optimisation steps above on random labels leave a network whose
        # outputs are nearly constant, i.e. frames without seeds and a watershed with nothing to do
        from microbeseg_amd.utils.unets import build_unet
        torch.manual_seed(3)
        net_i = build_unet(args.arch, args.act, "conv", args.norm, dev, 1, ch_out=1, filters=tuple(args.filters))
        out["inference"] = inference_metric(args, net_i, dev)
        del net_i
    release()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, tuple(args.filters))
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
