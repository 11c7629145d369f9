"""Ranger = RAdam + Lookahead + Gradient Centralisation, same constructor / semantics as the optimizer vendored by the
reference (``src/training/ranger2020.py``: ``centralized_gradient`` :31-40, ``Ranger.__init__`` :45-95,
``Ranger.step`` :101-208; upstream lessw2020/Ranger-Deep-Learning-Optimizer, Apache-2.0).

Re-implemented for the MI355X path: the per-parameter update is expressed with a handful of in-place device ops
(optimizer plumbing on PyTorch-ROCm; HBM-bound, ~28 B/param).  Behaviour that callers can observe is kept:
  * gradient centralisation of every gradient with dim > 1 (``gc_conv_only=False``) before the moments (``gc_loc``),
  * the rectification term is cached in a 10-slot ring shared by all parameters and keyed by ``step % 10``,
  * while N_sma <= threshold the update is plain bias-corrected momentum (steps 1-5 for beta2 = 0.999),
  * lookahead per parameter every ``k`` steps with the slow weights initialised from the initial parameters,
  * the constructor prints the same banner lines.
"""
import math

import torch
from torch.optim.optimizer import Optimizer


def centralized_gradient(x, use_gc=True, gc_conv_only=False):
    """In-place gradient centralisation: subtract the mean over all dims but the first."""
    if use_gc and x.dim() > (3 if gc_conv_only else 1):
        x.sub_(x.mean(dim=tuple(range(1, x.dim())), keepdim=True))
    return x


class Ranger(Optimizer):

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False, gc_loc=True):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f'Invalid slow update rate: {alpha}')
        if not 1 <= k:
            raise ValueError(f'Invalid lookahead steps: {k}')
        if not lr > 0:
            raise ValueError(f'Invalid Learning Rate: {lr}')
        if not eps > 0:
            raise ValueError(f'Invalid eps: {eps}')
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold,
                        eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha = alpha
        self.k = k
        self.radam_buffer = [[None, None, None] for _ in range(10)]
        self.gc_loc = gc_loc
        self.use_gc = use_gc
        self.gc_conv_only = gc_conv_only
        print(f"Ranger optimizer loaded. \nGradient Centralization usage = {self.use_gc}")
        if self.use_gc and not self.gc_conv_only:
            print("GC applied to both conv and fc layers")
        elif self.use_gc and self.gc_conv_only:
            print("GC applied to conv layers only")

    def __setstate__(self, state):
        print("set state called")
        super().__setstate__(state)

    def _rectification(self, step, beta1, beta2):
        slot = self.radam_buffer[step % 10]
        if slot[0] != step:
            beta2_t = beta2 ** step
            n_max = 2 / (1 - beta2) - 1
            n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
            if n_sma > self.N_sma_threshhold:
                size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max /
                                 (n_max - 2)) / (1 - beta1 ** step)
            else:
                size = 1.0 / (1 - beta1 ** step)
            slot[0], slot[1], slot[2] = step, n_sma, size
        return slot[1], slot[2]

    def _fused_job(self, job, p, state, group, beta1, beta2):
        """MI355X path: fill the record of `p` for the multi-tensor launch (csrc/loss.hip: ranger_step_multi_kernel does
        the whole update of every tensor of the group: centralisation, moments, rectified step, lookahead)."""
        if len(state) == 0:
            state['step'] = 0
            state['exp_avg'] = torch.zeros_like(p)
            state['exp_avg_sq'] = torch.zeros_like(p)
            state['slow_buffer'] = p.detach().clone()
        state['step'] += 1
        n_sma, step_size = self._rectification(state['step'], beta1, beta2)
        g = p.grad.detach().contiguous()
        do_gc = self.use_gc and g.dim() > (3 if self.gc_conv_only else 1)
        job.p, job.g = p.data_ptr(), g.data_ptr()
        job.m, job.v, job.slow = (state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr(),
                                  state['slow_buffer'].data_ptr())
        job.n = p.numel()
        job.rows = g.shape[0] if do_gc else 0
        job.step_lr = step_size * group['lr']
        job.flags = int(n_sma > self.N_sma_threshhold) | (int(state['step'] % group['k'] == 0) << 1)
        return g                                     # kept alive until the launch is enqueued

    def _fused_group(self, ps, group, beta1, beta2):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        jobs = (_lib.MsegRangerJob * len(ps))()
        keep = [self._fused_job(j, p, self.state[p], group, beta1, beta2) for j, p in zip(jobs, ps)]
        _lib.check(lib.mseg_ranger_step_multi(ctypes.addressof(jobs), len(ps), beta1, beta2, group['eps'], self.alpha,
                                              torch.cuda.current_stream(ps[0].device).cuda_stream), "ranger_step_multi")
        del keep
        torch.autograd.graph.increment_version(ps)   # written through raw pointers: packed operands are now stale

    @torch.no_grad()
    def step(self, closure=None):
        fused = []
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            batch = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('Ranger optimizer does not support sparse gradients')
                state = self.state[p]
                if p.is_cuda and p.dtype == torch.float32 and group['weight_decay'] == 0 and self.gc_loc:
                    batch.setdefault(p.device, []).append(p)
                    continue
                grad = p.grad.detach().float().clone()
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, dtype=torch.float32)
                    state['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32)
                    state['slow_buffer'] = p.detach().clone()
                m, v = state['exp_avg'], state['exp_avg_sq']
                if self.gc_loc:
                    centralized_gradient(grad, self.use_gc, self.gc_conv_only)
                state['step'] += 1
                v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
                m.mul_(beta1).add_(grad, alpha=1 - beta1)
                n_sma, step_size = self._rectification(state['step'], beta1, beta2)
                if n_sma > self.N_sma_threshhold:
                    update = m / v.sqrt().add_(group['eps'])
                else:
                    update = m.clone()
                if group['weight_decay'] != 0:
                    update.add_(p.detach().float(), alpha=group['weight_decay'])
                if not self.gc_loc:
                    centralized_gradient(update, self.use_gc, self.gc_conv_only)
                p.add_(update.to(p.dtype), alpha=-step_size * group['lr'])
                if state['step'] % group['k'] == 0:
                    slow = state['slow_buffer']
                    slow.add_(p - slow, alpha=self.alpha)
                    p.copy_(slow)
            for ps in batch.values():                # the CUDA parameters of the group: one launch per 48 tensors
                self._fused_group(ps, group, beta1, beta2)
                fused += ps
        if fused:            # all packed convolution operands of the updated weights in ONE launch (engine.repack_all)
            from .. import engine
            engine.repack_all(fused)
        return None
