"""hipGraph capture of one training step (forward, loss, backward, optimizer step).

A step of the hot path is ~600 kernel launches; at the GUI's default batch sizes (4, 8) or in bf16 mode the host cannot
enqueue them as fast as the MI355X executes them.  ``GraphedTrainStep`` records the step once (``torch.cuda.CUDAGraph`` is a
hipGraph on ROCm) and replays it: the batch is copied into static input tensors, everything else — activations, gradients,
optimizer state — lives in the graph's private memory pool at fixed addresses.

Rules of the capture (all met by this build's step): no host synchronisation inside the step (losses stay device
scalars), every kernel on the capturing stream (the C ABI takes the stream from torch), a capturable optimizer:
``training/optim.FusedAdam`` (step counter AND learning rate on the device: a scheduler's change reaches the recorded step
through ``sync_device_scalars``, nothing is re-recorded) or ``torch.optim.Adam(..., capturable=True)`` (the rate is baked
into the graph, ``set_lr`` re-records it).  Single-GPU only: the data-parallel wrapper overlaps RCCL all-reduces with the
backward pass, which is left to the eager path.
The first ``warmup`` calls run eagerly (workspaces, lazily built tables and the optimizer state must exist before the
capture) — with real batches, so the trajectory is the eager one.
"""
import torch

from .. import engine


class GraphedTrainStep:
    def __init__(self, step_fn, optimizer, warmup=2):
        """step_fn(*batch) -> loss tensor (device scalar); must run forward, backward and optimizer.step() on the given
        tensors and call optimizer.zero_grad(set_to_none=True) first."""
        self.step_fn = step_fn
        self.optimizer = optimizer
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.static_loss = None
        self.shapes = None
        self.stream = torch.cuda.Stream()            # warm-up and capture run on the same side stream (torch's recipe)

    def _capture(self, batch):
        self.static_in = [b.clone() if b is not None else None for b in batch]
        self.shapes = self._signature(batch)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=self.stream):
            self.static_loss = self.step_fn(*self.static_in)
        self.graph = g

    @staticmethod
    def _signature(batch):
        return [(tuple(b.shape), b.dtype) if b is not None else None for b in batch]

    def invalidate(self):
        """forget the recorded step (learning rate or any other baked-in scalar changed)"""
        self.graph = None

    def set_lr(self, lr):
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        if not hasattr(self.optimizer, "sync_device_scalars"):
            self.invalidate()

    def __call__(self, *batch):
        self.calls += 1
        if self.calls <= self.warmup:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                loss = self.step_fn(*batch)
            torch.cuda.current_stream().wait_stream(self.stream)
            return loss
        # the FIRST captured signature stays the one that is ever recorded: after invalidate() (a learning-rate step) the next
        # call may well be the short last batch of an epoch, and recording that shape would leave every full batch eager
        if self.shapes is not None and self._signature(batch) != self.shapes:
            return self.step_fn(*batch)              # e.g. the last, smaller batch of an epoch: eager
        fused = hasattr(self.optimizer, "sync_device_scalars")
        if fused:
            self.optimizer.sync_device_scalars()     # learning rate / step count as the host sees them now
        if self.graph is None:
            self._capture(batch)                     # (records only; the replay below is the step)
            if fused:
                self._rewind_capture_bookkeeping()
        else:
            for s, b in zip(self.static_in, batch):
                if s is not None:
                    s.copy_(b, non_blocking=True)
        self.graph.replay()
        engine.note_training_step()                  # the replay updated running statistics without running any Python
        if fused:
            self.optimizer.note_replayed_step()
        else:
            # the parameters changed without their version counters moving: anything cached against them (the packed GEMM
            # operands of engine.py) must look stale to the next eager forward
            torch.autograd.graph.increment_version([p for g in self.optimizer.param_groups for p in g["params"]])
        return self.static_loss

    def _rewind_capture_bookkeeping(self):
        """FusedAdam.step() ran (in Python) while the step was being recorded and counted a step that no kernel has
        executed yet"""
        opt = self.optimizer
        for (_, ps, _, _), ds in zip(opt._groups, opt._dev_state):
            step = opt.state[ps[0]]["step"] - 1
            for p in ps:
                opt.state[p]["step"] = step
            ds[2] = step
