"""Loss functions of the training hot path as fused HIP kernels (forward value + gradient).

Mirror of ``src/training/losses.py`` (reference): ``get_loss`` (:6-37) returns a callable for ``label_type ==
'boundary'`` (``ce_dice`` :71-97 or plain cross entropy) and a ``{'border', 'cell'}`` dict of callables for
``'distance'`` (SmoothL1 / L1 / MSE, :24-32).  The callables take the same (prediction, target) tensors as the
``torch.nn`` losses they replace and return a scalar tensor that supports ``.backward()`` and ``.item()``.
"""
import torch
import torch.nn as nn

from .. import _lib

_KIND = {"smooth_l1": 0, "l1": 1, "l2": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"microbeseg_amd {what}: expected a CUDA (ROCm) tensor — the HIP path has no CPU fallback")


class _RegressionLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind):
        lib = _lib.load()
        pred = pred.contiguous()
        target = target.contiguous().to(torch.float32)
        n = pred.numel()
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        ws = torch.empty(lib.mseg_loss_workspace_bytes(n), dtype=torch.uint8, device=pred.device)
        _lib.check(lib.mseg_regression_loss(pred.data_ptr(), target.data_ptr(), n, kind, out.data_ptr(),
                                            ws.data_ptr(), _stream()), "regression_loss")
        ctx.save_for_backward(pred, target)
        ctx.kind = kind
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        g = gout.contiguous().to(torch.float32).reshape(1)
        grad = torch.empty_like(pred)
        _lib.check(lib.mseg_regression_loss_bwd(pred.data_ptr(), target.data_ptr(), pred.numel(), ctx.kind,
                                                g.data_ptr(), grad.data_ptr(), _stream()), "regression_loss_bwd")
        return grad, None, None


class RegressionLoss(nn.Module):
    """nn.SmoothL1Loss() / nn.L1Loss() / nn.MSELoss() (mean reduction) of one distance head."""

    def __init__(self, kind):
        super().__init__()
        self.kind = _KIND[kind]

    def forward(self, pred, target):
        _require_cuda(pred, "regression loss")
        if pred.shape != target.shape:
            raise RuntimeError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        return _RegressionLossFn.apply(pred, target, self.kind)


class _CeDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, with_dice):
        lib = _lib.load()
        y_pred = y_pred.contiguous()
        y_true = y_true.contiguous().to(torch.int64)
        N, Cc, H, W = y_pred.shape
        if Cc != 3:
            raise RuntimeError("ce_dice expects 3 classes")
        dev = y_pred.device
        sums = torch.empty(6, dtype=torch.float64, device=dev)
        ce = torch.empty(1, dtype=torch.float64, device=dev)
        ws = torch.empty(lib.mseg_loss_workspace_bytes(N * H * W), dtype=torch.uint8, device=dev)
        _lib.check(lib.mseg_ce_dice_fwd(y_pred.data_ptr(), y_true.data_ptr(), N, H * W, int(with_dice),
                                        sums.data_ptr(), ce.data_ptr(), ws.data_ptr(), _stream()), "ce_dice_fwd")
        sums, dice_weight = _allreduce_dice(sums, float(N * H * W))
        loss = ce[0] / (N * H * W)
        if with_dice:
            for c in (1, 2):
                i, p, g = sums[(c - 1) * 3 + 0], sums[(c - 1) * 3 + 1], sums[(c - 1) * 3 + 2]
                loss = loss + 0.5 * c * (1.0 - (2.0 * i + 1.0) / (g + p + 1.0))
        ctx.save_for_backward(y_pred, y_true, sums)
        ctx.with_dice = int(with_dice)
        ctx.dice_weight = dice_weight
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        y_pred, y_true, sums = ctx.saved_tensors
        N, _, H, W = y_pred.shape
        g = gout.contiguous().to(torch.float32).reshape(1)
        grad = torch.empty_like(y_pred)
        _lib.check(lib.mseg_ce_dice_bwd(y_pred.data_ptr(), y_true.data_ptr(), N, H * W, ctx.with_dice,
                                        sums.data_ptr(), float(N * H * W), float(ctx.dice_weight), g.data_ptr(),
                                        grad.data_ptr(), _stream()),
                   "ce_dice_bwd")
        return grad, None, None


def _allreduce_dice(sums, total):
    """Data-parallel training: the reference computes the Dice sums on the gathered global batch (nn.DataParallel,
    losses.py:65-66) -> all-reduce the six partial sums.  Returns (global sums, dice weight)."""
    from ..parallel import allreduce_dice_sums
    return allreduce_dice_sums(sums, total)


def dice_loss(y_pred, y_true, use_sigmoid=True):
    """Dice loss of ONE output channel (reference losses.py:40-68): 1 - (2 sum(g p) + 1) / (sum(g^2) + sum(p^2) + 1), p = the
    prediction (through a sigmoid when ``use_sigmoid``), g = the ground truth, sums over the whole batch.  Not called by
    the training path (the reference's ``ce_dice`` carries its own per-class form, :71-97 -> the fused kernel below);
    kept for callers of the module's public surface.  Device tensors only, like every loss here."""
    _require_cuda(y_pred, "dice_loss")
    p = (torch.sigmoid(y_pred) if use_sigmoid else y_pred).reshape(-1)
    g = y_true.reshape(-1).to(p.dtype)
    return 1 - (2. * torch.dot(g, p) + 1.) / (torch.dot(g, g) + torch.dot(p, p) + 1.)


def ce_dice(y_pred, y_true, num_classes=3):
    """Sum of cross-entropy and channel-wise Dice loss (reference losses.py:71-97), logits [N,3,H,W], labels [N,H,W]."""
    _require_cuda(y_pred, "ce_dice")
    if num_classes != 3:
        raise RuntimeError("the HIP ce_dice kernel is specialised for the reference's 3 classes")
    return _CeDiceFn.apply(y_pred, y_true, True)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() for the 3-class boundary logits ('ce' option, reference losses.py:19-20)."""

    def forward(self, y_pred, y_true):
        _require_cuda(y_pred, "cross entropy")
        return _CeDiceFn.apply(y_pred, y_true, False)


def get_loss(loss_function, label_type):
    """ Get loss function(s) for the training process (same contract as the reference, losses.py:6-37). """
    if label_type == 'boundary':
        if loss_function == 'ce_dice':
            criterion = ce_dice
        elif loss_function == 'ce':
            criterion = CrossEntropyLoss()
        else:
            raise Exception('Loss unknown')
    elif label_type == 'distance':
        if loss_function not in _KIND:
            raise Exception('Loss unknown')
        criterion = {'border': RegressionLoss(loss_function), 'cell': RegressionLoss(loss_function)}
    else:
        raise Exception('Label type unknown')
    return criterion
