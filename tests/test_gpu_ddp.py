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
