"""GPU (one MI355X, two processes sharing it, gloo transport): the data-parallel wrapper around the HIP net gives every
rank the average of the per-rank gradients, keeps replicas identical and lets rank 0's BatchNorm buffers win.
(RCCL itself needs one GPU per rank; the driver exercises it at N=2..8.  The collective calls are backend-agnostic.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
FILTERS = (8, 16)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank):
    g = torch.Generator().manual_seed(50 + rank)
    return (torch.rand(2, 1, 32, 32, generator=g) * 2 - 1, torch.rand(2, 1, 32, 32, generator=g),
            torch.rand(2, 1, 32, 32, generator=g))


def _step(net, batch, dev):
    from microbeseg_amd.training.losses import get_loss
    crit = get_loss("smooth_l1", "distance")
    x, lb, lc = (t.to(dev) for t in batch)
    net.train()
    for p in net.parameters():
        p.grad = None
    border, cell = net(x)
    loss = crit["border"](border, lb) + crit["cell"](cell, lc)
    loss.backward()
    return {k: p.grad.detach().cpu() for k, p in net.named_parameters()}


def _worker(rank, world, port, ret, precision="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from microbeseg_amd import engine
        from microbeseg_amd.utils.unets import build_unet
        engine.set_precision(precision)
        dev = torch.device("cuda:0")
        torch.manual_seed(123 + rank)       # replicas start different; the wrapper must broadcast rank 0's weights
        net = build_unet("DU", "relu", "conv", "bn", dev, world, filters=FILTERS)
        grads = _step(net, _batch(rank), dev)
        sd = {k: v.detach().cpu() for k, v in net.module.state_dict().items()}
        ret[rank] = ({k.replace("module.", "", 1): v for k, v in grads.items()}, sd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_rank_gradient_average(precision, request):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd import engine
    request.addfinalizer(lambda: engine.set_precision("fp32"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, precision), nprocs=2, join=True)
    engine.set_precision(precision)          # the single-process reference below runs in the same mode
    (g0, sd0), (g1, sd1) = ret[0], ret[1]
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k                       # every rank holds the same averaged gradient
    # single-process reference: rank-0 initial weights, per-rank gradients averaged by hand
    from microbeseg_amd.utils.unets import build_unet
    dev = torch.device("cuda:0")
    torch.manual_seed(123)
    net = build_unet("DU", "relu", "conv", "bn", dev, 1, filters=FILTERS)
    init = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ga = _step(net, _batch(0), dev)
    net.load_state_dict(init)
    gb = _step(net, _batch(1), dev)
    for k in ga:
        want = (ga[k] + gb[k]) / 2
        assert torch.allclose(g0[k], want, rtol=1e-5, atol=1e-7), k
    # parameters identical on both ranks (broadcast at first forward); BN buffers: rank 0's batch statistics
    for k in sd0:
        if "running" not in k and "num_batches" not in k:
            assert torch.equal(sd0[k], sd1[k]), k


def _nccl_worker(port, ret):
    """ONE rank, backend "nccl" (= RCCL on ROCm): the collective path of parallel.py executed by the real backend"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from microbeseg_amd import parallel
        from microbeseg_amd.training.optim import FusedAdam
        from microbeseg_amd.utils.unets import build_unet
        # 1. GradBucketer on device tensors: arena slices in place + a loose tensor, async work handles, stream ordering
        g = torch.Generator().manual_seed(5)
        arena = torch.zeros(3 * 4096, device=dev)
        views = [arena[i * 4096:i * 4096 + 4000].view(40, 100) for i in range(3)]
        vals = [torch.randn(40, 100, generator=g).to(dev) for _ in range(3)]
        loose, loose_val = torch.empty(777, device=dev), torch.randn(777, generator=g).to(dev)
        b = parallel.GradBucketer(bucket_bytes=8192)
        for v, x in zip(views[::-1], vals[::-1]):
            v.copy_(x * 3.0).div_(3.0)                    # produced by kernels on the current stream right before add()
            b.add([v])
        loose.copy_(loose_val)
        b.add([loose])
        b.finish()
        torch.cuda.synchronize()
        ok_bucket = all(torch.allclose(v, x, rtol=1e-6, atol=1e-6) for v, x in zip(views, vals)) and \
            torch.equal(loose, loose_val)
        # 2. buffers: packed broadcast and the in-place arena
        bn = torch.nn.BatchNorm2d(8).to(dev)
        bn.running_mean.fill_(3.0)
        parallel.broadcast_buffers(bn, 0)
        a = parallel.BufferArena(bn)
        a.broadcast(0)
        torch.cuda.synchronize()
        ok_buf = bool((bn.running_mean == 3.0).all()) and bn.running_mean.data_ptr() == a.flat.data_ptr()
        # 3. a whole training step through RcclDataParallel with the collectives forced on: with one rank the averaged
        #    gradients and the updated weights must equal the plain single-GPU step bit for bit
        from microbeseg_amd.training.losses import get_loss
        crit = get_loss("smooth_l1", "distance")
        x, lb, lc = (t.to(dev) for t in _batch(0))

        def run(wrapped):
            torch.manual_seed(9)
            net = build_unet("DU", "relu", "conv", "bn", dev, 2 if wrapped else 1, filters=FILTERS)
            opt = FusedAdam(net.parameters(), lr=1e-3)
            net.train()
            for _ in range(2):
                opt.zero_grad()
                border, cell = net(x)
                loss = crit["border"](border, lb) + crit["cell"](cell, lc)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            mod = net.module if wrapped else net
            return {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
        parallel._FORCE_COLLECTIVES = True
        try:
            sd_w = run(True)
        finally:
            parallel._FORCE_COLLECTIVES = False
        sd_p = run(False)
        ok_step = all(torch.equal(sd_w[k], sd_p[k]) for k in sd_p)
        ret["result"] = (ok_bucket, ok_buf, ok_step, dist.get_backend())
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one_executes_the_collective_path():
    """No second GPU exists on the test box, so RCCL cannot run N = 2 here — but a process group of ONE rank on the real
    backend ("nccl" = RCCL) executes every collective call of parallel.py on the device: async all-reduce of in-place arena
    slices and of a packed bucket (work handles, wait(), ordering against kernels on the current stream), buffer
    broadcasts, and a full two-step training run through RcclDataParallel that must equal the plain run bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.Manager()
    ret = mgr.dict()
    p = mp.get_context("spawn").Process(target=_nccl_worker, args=(_free_port(), ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0, f"nccl worker exit code {p.exitcode}"
    ok_bucket, ok_buf, ok_step, backend = ret["result"]
    assert backend == "nccl"
    assert ok_bucket and ok_buf and ok_step, (ok_bucket, ok_buf, ok_step)
