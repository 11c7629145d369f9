"""Adam (amsgrad) of the training hot path as ONE HIP launch per step.

The reference trains with ``torch.optim.Adam(net.parameters(), lr=8e-4, betas=(0.9, 0.999), eps=1e-08, weight_decay=0,
amsgrad=True)`` (src/training/train.py:380-385).  On PyTorch-ROCm that is ~8 multi-tensor launches per step plus, in
this build, one weight repack per convolution operand (74 small launches).  ``FusedAdam`` keeps the same semantics and
the ``torch.optim.Optimizer`` interface (``param_groups[...]['lr']`` is what ``ReduceLROnPlateau`` drives,
``state_dict()`` holds ``step / exp_avg / exp_avg_sq / max_exp_avg_sq`` per parameter like the stock optimizer) but lays
the network out for the update:

* parameters, gradients and the three moment tensors are views of five flat fp32 arenas (the parameters are re-pointed
  once, at construction), so the update of all 46 M parameters of the default DU-Net is ``mseg_adam_amsgrad_step`` on one
  range: 36 B per parameter at HBM speed;
* the explicit backward of ``engine.py`` writes each gradient straight into its arena view (``p.grad`` is allocated
  once and never freed: ``zero_grad`` only marks it as consumed), so no gradient is copied or accumulated by autograd;
* right after the update all packed GEMM operands of the convolutions are refreshed in one launch
  (``engine.repack_all``), the next forward / backward finds them current;
* the step counter and the learning rate the kernel uses live on the device (``mseg_adam_amsgrad_step_dev``): a step has
  no per-step host scalar, so ``training/graph_step.py`` can record it in a hipGraph and replay it.  ``state[p]['step']``
  stays a host integer (the stock optimizer's state layout); a replayed step is counted through ``note_replayed_step``.
"""
import torch
from torch.optim.optimizer import Optimizer

from .. import _lib


class FusedAdam(Optimizer):
    """``torch.optim.Adam(..., amsgrad=True, weight_decay=0)`` for fp32 CUDA (ROCm) parameters of one device."""

    def __init__(self, params, lr=8e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True):
        if not lr > 0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not eps >= 0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Invalid betas: {betas}")
        if weight_decay != 0 or not amsgrad:
            raise ValueError("FusedAdam implements the reference's configuration only: amsgrad=True, weight_decay=0")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        # per param group: (index into self.param_groups, params, arenas, length).  The INDEX, not the dict:
        # load_state_dict replaces the group dicts, and schedulers edit the ones in self.param_groups
        self._groups = []
        self._dev_state = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                if not p.is_cuda or p.dtype != torch.float32 or p.device != dev:
                    raise RuntimeError("FusedAdam: fp32 parameters on one CUDA (ROCm) device expected — the HIP path "
                                       "has no CPU fallback")
            offs, n = [], 0
            for p in ps:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4          # every view 16-byte aligned
            flat = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(5)]
            fp, fg, fm, fv, fx = flat
            with torch.no_grad():
                for p, o in zip(ps, offs):
                    k = p.numel()
                    fp[o:o + k].copy_(p.detach().reshape(-1))
                    p.data = fp[o:o + k].view_as(p)
                    p.grad = fg[o:o + k].view_as(p)
                    p.__dict__["_mseg_grad_direct"] = True     # engine.backward writes the gradient in place
                    p.__dict__["_mseg_grad_fresh"] = True      # ... once per zero_grad()
                    p.__dict__["_mseg_grad_ver"] = p.grad._version
                    st = self.state[p]
                    st["step"] = 0
                    st["exp_avg"] = fm[o:o + k].view_as(p)
                    st["exp_avg_sq"] = fv[o:o + k].view_as(p)
                    st["max_exp_avg_sq"] = fx[o:o + k].view_as(p)
            # [lr, steps done, scratch, scratch] for mseg_adam_amsgrad_step_dev
            dev_state = torch.tensor([float(group["lr"]), 0.0, 0.0, 0.0], dtype=torch.float64, device=dev)
            self._groups.append((gi, ps, flat, n))
            self._dev_state.append([dev_state, float(group["lr"]), 0])     # tensor, lr and step count it holds

    def sync_device_scalars(self):
        """Bring the device copies of (learning rate, step count) in line with ``param_groups`` / ``state``.  step() does it
        itself; a replayed hipGraph does not run step(), so GraphedTrainStep calls this before each replay (a scheduler may
        have changed the rate, load_state_dict the count).  Must not be called while a stream is being captured with
        changed values — the fill would be recorded — hence the assertion."""
        for (gi, ps, _, _), ds in zip(self._groups, self._dev_state):
            lr, step = float(self.param_groups[gi]["lr"]), int(self.state[ps[0]]["step"])
            if lr != ds[1] or step != ds[2]:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdam: learning rate / step count changed inside a hipGraph capture")
                ds[0][:2] = torch.tensor([lr, float(step)], dtype=torch.float64)      # one small H2D copy
                ds[1], ds[2] = lr, step

    def note_replayed_step(self):
        """a recorded step() ran on the device without this object's step() being called"""
        from .. import engine
        for (_, ps, _, _), ds in zip(self._groups, self._dev_state):
            step = self.state[ps[0]]["step"] + 1
            for p in ps:
                self.state[p]["step"] = step
            ds[2] = step
            torch.autograd.graph.increment_version(ps)
            engine.mark_packs_fresh(ps)              # the recorded repack_all ran after the recorded update

    def zero_grad(self, set_to_none=True):
        """The gradient arena is persistent: the next backward overwrites it (set_to_none semantics without the frees)."""
        for _, ps, flat, _ in self._groups:
            for p in ps:
                p.__dict__["_mseg_grad_fresh"] = True
                if p.grad is None:                 # somebody dropped the view: restore it
                    raise RuntimeError("FusedAdam: p.grad was replaced; gradients must stay views of the arena")
                p.__dict__["_mseg_grad_ver"] = p.grad._version

    def load_state_dict(self, state_dict):
        """values are copied INTO the arenas (the state tensors must stay views of them); a state dict without moments
        (a fresh torch.optim.Adam, or one saved before its first step) leaves the arenas at their current values"""
        views = {id(p): dict(self.state[p]) for _, ps, _, _ in self._groups for p in ps}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for _, ps, _, _ in self._groups:
                for p in ps:
                    new, old = self.state[p], views[id(p)]
                    for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                        if key in new:
                            old[key].copy_(new[key])
                        new[key] = old[key]
                    new["step"] = int(new.get("step", old["step"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        from .. import engine
        self.sync_device_scalars()
        for (gi, ps, (fp, fg, fm, fv, fx), n), ds in zip(self._groups, self._dev_state):
            group = self.param_groups[gi]                # looked up per step: lr schedulers / load_state_dict edit THESE
            # written since zero_grad(): by the engine's backward (raw pointers: it clears the flag) or in place by the caller
            # (p.grad.copy_(...): the tensor's version counter moves)
            stale = [p for p in ps if p.__dict__.get("_mseg_grad_fresh") and
                     p.grad._version == p.__dict__.get("_mseg_grad_ver")]
            if stale:
                # nobody has written these gradients since zero_grad(): the one launch below updates the whole arena,
                # so (unlike torch.optim.Adam, which skips parameters without a gradient) it would re-apply the previous
                # step's values.  The networks of this path always produce every gradient; anything else is a bug upstream.
                raise RuntimeError(f"FusedAdam.step(): {len(stale)} of {len(ps)} parameters received no gradient since "
                                   "zero_grad() (step() without backward, or an unused parameter)")
            step = self.state[ps[0]]["step"] + 1
            b1, b2 = group["betas"]
            _lib.check(lib.mseg_adam_amsgrad_step_dev(fp.data_ptr(), fg.data_ptr(), fm.data_ptr(), fv.data_ptr(),
                                                      fx.data_ptr(), n, ds[0].data_ptr(), b1, b2, group["eps"], stream),
                       "adam_amsgrad_step_dev")
            for p in ps:
                self.state[p]["step"] = step
            ds[2] = step
            torch.autograd.graph.increment_version(ps)   # the kernel wrote through raw pointers
            engine.repack_all(ps)
        return loss


def make_adam(params, lr=8e-4, capturable=False):
    """The reference's Adam configuration (train.py:380-385) as FusedAdam (eager or recorded in a hipGraph: its step
    counter is on the device).  `capturable` is accepted for callers of earlier rounds and changes nothing."""
    return FusedAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True)
