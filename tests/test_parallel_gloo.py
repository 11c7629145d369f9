"""CPU, world_size 2 over gloo: the data-parallel plumbing of microbeseg_amd.parallel (gradient bucketing/averaging,
buffer broadcast, global Dice sums) — the N>1 path of bench.py minus the GPU kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return dict(ret)


def _bucket_job(rank, world):
    from microbeseg_amd.parallel import GradBucketer
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(5, 7, generator=g), torch.randn(3, generator=g), torch.randn(1000, generator=g),
             torch.randn(2, 2, 3, 3, generator=g)]
    mine = [t.clone() for t in grads]
    b = GradBucketer(bucket_bytes=2048)          # forces several buckets + a tail bucket
    b.add(mine[:2]); b.add([None, mine[2]]); b.add(mine[3:])
    b.finish()
    return [t.tolist() for t in mine], [t.tolist() for t in grads]


def test_grad_bucketer_averages_across_ranks():
    out = _run(_bucket_job)
    avg0, raw0 = out[0]
    avg1, raw1 = out[1]
    for a0, a1, r0, r1 in zip(avg0, avg1, raw0, raw1):
        want = (torch.tensor(r0) + torch.tensor(r1)) / 2
        assert torch.allclose(torch.tensor(a0), want, atol=1e-6)
        assert torch.allclose(torch.tensor(a1), want, atol=1e-6)


def _buffer_job(rank, world):
    from microbeseg_amd.parallel import broadcast_buffers, allreduce_dice_sums, allreduce_scalar_sum
    bn = torch.nn.BatchNorm2d(4)
    bn.running_mean.fill_(float(rank + 1))
    bn.num_batches_tracked.fill_(10 * (rank + 1))
    broadcast_buffers(bn, 0)
    sums = torch.arange(6, dtype=torch.float64) * (rank + 1)
    sums, w = allreduce_dice_sums(sums, 100.0)
    tot = allreduce_scalar_sum(1.5 * (rank + 1), "cpu")
    return bn.running_mean.tolist(), int(bn.num_batches_tracked), sums.tolist(), w, tot


def test_buffers_follow_rank0_and_dice_sums_are_global():
    out = _run(_buffer_job)
    for r in (0, 1):
        rm, nbt, sums, w, tot = out[r]
        assert rm == [1.0] * 4 and nbt == 10
        assert sums == [0.0, 3.0, 6.0, 9.0, 12.0, 15.0] and w == 2.0
        assert tot == pytest.approx(4.5)


def _wrapper_job(rank, world):
    """RcclDataParallel around a tiny torch module: identical initial replicas + .module contract."""
    from microbeseg_amd.parallel import RcclDataParallel
    torch.manual_seed(rank)                      # different init per rank on purpose
    m = RcclDataParallel(torch.nn.Linear(3, 2))
    x = torch.ones(1, 3)
    y = m(x)
    return m.module.weight.detach().tolist(), y.detach().tolist(), hasattr(m, "module")


def test_wrapper_broadcasts_parameters_from_rank0():
    out = _run(_wrapper_job)
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][2]


def _eval_switch_job(rank, world):
    """training forwards let each replica's BatchNorm statistics drift (per-replica batches); the first eval forward after
    them must run on rank 0's buffers — the ones the checkpoint holds (parallel.py)."""
    from microbeseg_amd.parallel import RcclDataParallel
    torch.manual_seed(0)
    m = RcclDataParallel(torch.nn.Sequential(torch.nn.Conv2d(1, 4, 1), torch.nn.BatchNorm2d(4)))
    g = torch.Generator().manual_seed(50 + rank)
    m.train()
    for _ in range(3):
        m(torch.randn(8, 1, 5, 5, generator=g) * (1 + 3 * rank))      # rank 1 sees much larger inputs
    drift = m.module[1].running_var.clone()
    m.eval()
    y = m(torch.ones(2, 1, 5, 5))
    return drift.tolist(), m.module[1].running_var.tolist(), y.detach().flatten().tolist()


def test_eval_after_training_uses_rank0_buffers():
    out = _run(_eval_switch_job)
    assert out[0][0] != out[1][0]                       # the replicas did drift apart while training
    assert out[0][1] == out[1][1] == out[0][0]          # ... and validate on rank 0's statistics
    assert out[0][2] == out[1][2]


def _arena_job(rank, world):
    """gradients that are views of one flat arena (FusedAdam's layout: parameter order, 16-byte aligned views) handed over in
    BACKWARD order: reduced in place, slice by slice, no packed copies"""
    from microbeseg_amd.parallel import GradBucketer
    g = torch.Generator().manual_seed(200 + rank)
    shapes = [(5, 7), (3,), (1000,), (2, 2, 3, 3), (130,), (64, 9)]
    offs, n = [], 0
    for sh in shapes:
        offs.append(n)
        n += (torch.Size(sh).numel() + 3) // 4 * 4
    arena = torch.zeros(n)
    views = []
    for sh, o in zip(shapes, offs):
        k = torch.Size(sh).numel()
        v = arena[o:o + k].view(sh)
        v.copy_(torch.randn(sh, generator=g))
        views.append(v)
    loose = torch.randn(17, generator=g)                    # a gradient outside the arena rides along (packed path)
    raw = [v.clone() for v in views] + [loose.clone()]
    b = GradBucketer(bucket_bytes=2048)
    b.add(views[4:][::-1]); b.add([loose, None]); b.add(views[2:4][::-1]); b.add(views[:2][::-1])
    n_slices = sum(1 for _, t, _ in b.inflight if t is None)
    b.finish()
    return [v.tolist() for v in views] + [loose.tolist()], [t.tolist() for t in raw], n_slices


def test_grad_bucketer_reduces_arena_views_in_place():
    out = _run(_arena_job)
    avg0, raw0, slices0 = out[0]
    avg1, raw1, _ = out[1]
    assert slices0 >= 1                                      # at least one bucket went out as an arena slice before finish()
    for a0, a1, r0, r1 in zip(avg0, avg1, raw0, raw1):
        want = (torch.tensor(r0) + torch.tensor(r1)) / 2
        assert torch.allclose(torch.tensor(a0), want, atol=1e-6)
        assert torch.allclose(torch.tensor(a1), want, atol=1e-6)


def _buffer_arena_job(rank, world):
    from microbeseg_amd.parallel import BufferArena
    m = torch.nn.Sequential(torch.nn.BatchNorm2d(4), torch.nn.BatchNorm2d(6))
    for i, bn in enumerate(m):
        bn.running_mean.fill_(float(10 * rank + i + 1))
        bn.running_var.fill_(float(rank + 2))
    a = BufferArena(m)
    same = all(b.data_ptr() >= a.flat.data_ptr() and b.data_ptr() < a.flat.data_ptr() + a.flat.numel() * 4
               for b in m.buffers() if b.is_floating_point())
    a.broadcast(0)
    m[0].running_mean += 1.0                                  # the module's buffers ARE the arena
    return same, m[0].running_mean.tolist(), m[1].running_var.tolist(), float(a.flat[0])


def test_buffer_arena_is_one_in_place_broadcast():
    out = _run(_buffer_arena_job)
    for r in (0, 1):
        same, rm, rv, first = out[r]
        assert same and rm == [2.0] * 4 and rv == [2.0] * 6 and first == 2.0
