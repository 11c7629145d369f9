"""Data-parallel training across the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl" on
ROCm) gradient all-reduce over xGMI, overlapped with the explicit backward pass.

This replaces the reference's single-process ``nn.DataParallel`` (src/utils/unets.py:51-52; per step it broadcasts
parameters/buffers, scatters the batch on dim 0 and reduce-adds the gradients onto device 0).  Semantics kept:
  * every rank holds a full replica; a step sees the *global* batch = concatenation of the per-rank batches;
  * gradients of the mean-reduced losses equal those of the global batch (sum over ranks / world size);
  * BatchNorm statistics are per replica (NOT synchronised) and replica 0's running stats are the ones that
    persist: buffers are broadcast from rank 0 before each training forward;
  * ``ce_dice`` uses Dice sums over the global batch (losses.py:65-66 run on the gathered outputs): the six partial
    sums are all-reduced inside the loss;
  * ``.module`` is the bare model (``get_weights`` / checkpoint writers use it, unets.py:74-75, train.py:512-513).

xGMI is point-to-point (7 links x ~153 GB/s per GPU); the 185.5 MB of fp32 gradients of the default DU-Net cost
~2 ms per step on a ring, against ~200 ms of compute, so buckets of ~32 MB launched as soon as the explicit backward
has produced them are fully hidden behind the remaining dgrad/wgrad kernels.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

BUCKET_BYTES = 32 * 1024 * 1024


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier():
    if world_size() > 1:
        dist.barrier()


_FORCE_COLLECTIVES = False     # tests: run the collective path with a group of ONE rank (the real backend executes it)


def collectives_on():
    return world_size() > 1 or (_FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def _arena_span(t):
    """(arena, first element, one past the last element) if `t` is a contiguous view of a flat 1-D arena tensor (the
    gradient views training/optim.FusedAdam hands out), else None"""
    base = t._base
    if base is None or base.dim() != 1 or not t.is_contiguous() or base.dtype != t.dtype:
        return None
    lo = t.storage_offset() - base.storage_offset()
    return base, lo, lo + t.numel()


class GradBucketer:
    """Collects gradient tensors as the backward pass produces them, all-reduces them asynchronously (RCCL stream) in
    buckets of ~``bucket_bytes`` and averages them at ``finish()``.  Backend agnostic (gloo in tests).

    Gradients that are views of a flat arena (``FusedAdam``: all gradients of the network live in ONE fp32 buffer, in
    parameter order) are reduced IN PLACE: the backward pass produces them from the last layer to the first, i.e. the
    finished part of the arena is a growing tail, and every bucket is one contiguous slice of it — no ``torch.cat``, no
    copy back, one ``mul_`` over the arena at the end.  Other gradients (stock optimizers) go through a packed copy."""

    GAP = 4            # arena views are 16-byte aligned: up to 3 padding floats between two neighbours

    def __init__(self, bucket_bytes=BUCKET_BYTES, group=None):
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.pending = []      # loose tensors of the bucket being filled
        self.pending_bytes = 0
        self.inflight = []     # (flat buffer, [tensors] or None for an arena slice, work handle)
        self.arenas = {}       # id(arena) -> [arena, sorted list of ready [lo, hi) spans not launched yet]

    def add(self, tensors):
        for t in tensors:
            if t is None:
                continue
            span = _arena_span(t)
            if span is None:
                self.pending.append(t)
                self.pending_bytes += t.numel() * t.element_size()
                continue
            arena, lo, hi = span
            ent = self.arenas.setdefault(id(arena), [arena, []])
            self._insert(ent[1], lo, hi)
            self._launch_arena(ent, force=False)
        if self.pending_bytes >= self.bucket_bytes:
            self._launch()

    def _insert(self, spans, lo, hi):
        """merge [lo, hi) into the sorted span list (neighbours up to GAP - 1 elements apart are joined: padding)"""
        spans.append([lo, hi])
        spans.sort()
        merged = [spans[0]]
        for s in spans[1:]:
            if s[0] - merged[-1][1] < self.GAP:
                merged[-1][1] = max(merged[-1][1], s[1])
            else:
                merged.append(s)
        spans[:] = merged

    def _launch_arena(self, ent, force):
        arena, spans = ent
        esz = arena.element_size()
        keep = []
        for lo, hi in spans:
            if force or (hi - lo) * esz >= self.bucket_bytes:
                view = arena[lo:hi]
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.inflight.append((view, None, work))
            else:
                keep.append([lo, hi])
        spans[:] = keep

    def _launch(self):
        if not self.pending:
            return
        tensors, self.pending, self.pending_bytes = self.pending, [], 0
        flat = torch.cat([t.reshape(-1) for t in tensors])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.inflight.append((flat, tensors, work))

    def finish(self):
        """Wait for all buckets; the averaged gradients are in the original tensors afterwards."""
        self._launch()
        for ent in self.arenas.values():
            self._launch_arena(ent, force=True)
        inv = 1.0 / dist.get_world_size(self.group)
        for flat, tensors, work in self.inflight:
            work.wait()
            flat.mul_(inv)                             # an arena slice: in place, done
            if tensors is not None:
                off, views = 0, []
                for t in tensors:
                    n = t.numel()
                    views.append(flat[off:off + n].view_as(t))
                    off += n
                torch._foreach_copy_(tensors, views)      # one multi-tensor launch per bucket
        self.inflight = []
        self.arenas = {}


class BufferArena:
    """The floating-point buffers of a module (BatchNorm running statistics) re-pointed into ONE flat tensor, so that
    "replica 0's buffers win" (nn.DataParallel semantics) is one in-place broadcast per step — no cat, no copy back.
    ``num_batches_tracked`` advances by one per training forward on every rank alike: it needs no collective."""

    def __init__(self, module):
        bufs = [b for b in module.buffers() if b.numel() > 0 and b.is_floating_point()]
        self.flat = None
        if not bufs:
            return
        n = sum((b.numel() + 3) // 4 * 4 for b in bufs)
        self.flat = torch.zeros(n, dtype=bufs[0].dtype, device=bufs[0].device)
        off = 0
        with torch.no_grad():
            for b in bufs:
                k = b.numel()
                self.flat[off:off + k].copy_(b.reshape(-1))
                b.data = self.flat[off:off + k].view_as(b)
                off += (k + 3) // 4 * 4

    def broadcast(self, src=0, group=None):
        if self.flat is not None:
            dist.broadcast(self.flat, src=src, group=group)


def broadcast_buffers(module, src=0, group=None):
    """Replica-0 buffers (BatchNorm running stats, num_batches_tracked) win, as under nn.DataParallel.  (The wrapper below
    uses a BufferArena instead; this packed form serves modules whose buffers were not re-pointed.)"""
    bufs = [b for b in module.buffers() if b.numel() > 0]
    if not bufs:
        return
    by_dtype = {}
    for b in bufs:
        by_dtype.setdefault(b.dtype, []).append(b)
    for dtype, group_bufs in by_dtype.items():
        flat = torch.cat([b.reshape(-1) for b in group_bufs])
        dist.broadcast(flat, src=src, group=group)
        off, views = 0, []
        for b in group_bufs:
            n = b.numel()
            views.append(flat[off:off + n].view_as(b))
            off += n
        torch._foreach_copy_(group_bufs, views)       # one multi-tensor launch instead of one copy per buffer


def broadcast_parameters(module, src=0, group=None):
    for p in module.parameters():
        dist.broadcast(p.data, src=src, group=group)


class RcclDataParallel(nn.Module):
    """Wrapper returned by ``build_unet(num_gpus > 1)``.  If torch.distributed is not initialised (single
    process) it degrades to a transparent wrapper that only provides the ``.module`` attribute."""

    def __init__(self, module, group=None):
        super().__init__()
        self.module = module
        self.group = group
        self._synced_init = False
        self._buffers_diverged = False   # a training forward has updated this replica's BatchNorm statistics
        self._buffer_arena = None

    def _broadcast_buffers(self):
        if self._buffer_arena is None:
            self._buffer_arena = BufferArena(self.module)
        self._buffer_arena.broadcast(0, self.group)

    def forward(self, *args, **kwargs):
        if collectives_on():
            if not self._synced_init:
                broadcast_parameters(self.module, 0, self.group)   # identical replicas (DataParallel re-broadcasts
                broadcast_buffers(self.module, 0, self.group)      # every step; replicas never diverge here)
                self._synced_init = True
            if self.module.training:
                self._broadcast_buffers()
                self._buffers_diverged = True
            elif self._buffers_diverged:
                # first eval forward after a training phase: every replica has since updated its running statistics from
                # its OWN last batch.  nn.DataParallel evaluates all replicas with device 0's buffers, and the state dict
                # that gets saved is rank 0's — validate exactly that model on every rank.
                self._broadcast_buffers()
                self._buffers_diverged = False
            self.module._grad_sync_factory = self._make_bucketer
        else:
            self.module._grad_sync_factory = None
        return self.module(*args, **kwargs)

    def _make_bucketer(self):
        return GradBucketer(group=self.group)


def sync_eval_buffers(net):
    """Called by EVERY rank at the start of a validation phase (also by ranks that will not run a single validation
    batch): replicas whose BatchNorm statistics have drifted during the training phase take rank 0's."""
    if isinstance(net, RcclDataParallel) and collectives_on() and net._buffers_diverged:
        net._broadcast_buffers()
        net._buffers_diverged = False


def allreduce_dice_sums(sums, total):
    """ce_dice: Dice numerators/denominators over the global batch (C3).  Returns (sums, dice_weight)."""
    w = world_size()
    if w > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums, float(w)


def allreduce_scalar_sum(value, device):
    """Sum a python float over ranks (validation loss bookkeeping so that every rank takes the same decisions)."""
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
