"""Executor of the U-Net / dual-decoder U-Net hot path on libmseg_hip (gfx950).

Host-side mirror of ``UNet.forward`` / ``DUNet.forward`` (reference: src/utils/unets.py:349-377, :463-506) and of
their autograd backward (train.py:488).  Python only sequences kernels; every FLOP runs in the HIP library:

* activations are NHWC fp32 and are kept in *pre-activation* form ``z = conv(x) + b``; the consumer applies
  activation + Batch/Group/InstanceNorm while staging its operand ("norm-on-load"), so neither the activated nor
  the normalised tensor, nor ``torch.cat([up, skip], 1)``, is ever written to HBM;
* the backward pass is explicit (dgrad / wgrad / norm-bwd kernels) and is exposed to PyTorch as ONE
  ``torch.autograd.Function`` so that ``loss.backward()`` and the stock optimizers keep working (plumbing only).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (ACT, NORM, MODE_CONV, MODE_TCONV, EPI_PLAIN, EPI_SCATTER2X2, MORDER_LINEAR, MORDER_PARITY,
                   ST_F32, ST_BF16, MsegSrc, MsegIgemm, MsegWgrad, MsegKernelInfo, check)

BN_EPS = 1e-5       # torch defaults used by the reference (unets.py:127-132)
BN_MOMENTUM = 0.1


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pad4(c):
    return (c + 3) // 4 * 4


def _st(t):
    """storage code (MSEG_ST_*) of an activation tensor"""
    return ST_BF16 if t.dtype == torch.bfloat16 else ST_F32


class Node:
    """An activation tensor in pre-norm form plus the tables its consumers need to normalise it on load."""
    __slots__ = ("z", "a", "N", "H", "W", "C", "act", "norm", "scale", "shift", "ss", "mean", "rstd", "layer", "inputs",
                 "grad")

    def __init__(self, z, N, H, W, C):
        self.z, self.N, self.H, self.W, self.C = z, N, H, W, C
        self.a = None          # act(z), materialised once for the expensive activations (see norm_stats)
        self.act, self.norm = 0, None
        self.scale = self.shift = self.mean = self.rstd = None
        self.ss = 0
        self.layer, self.inputs, self.grad = None, (), None

    def src(self, transform=True):
        s = MsegSrc()
        s.ptr = self.z.data_ptr()
        s.C = self.C
        s.dtype = _st(self.z)
        if transform:
            if self.a is not None:      # consumers read the stored activation: no transcendentals in their K-loops
                s.ptr = self.a.data_ptr()
                s.act = ACT["none"]
            else:
                s.act = self.act
            s.scale = _ptr(self.scale)
            s.shift = _ptr(self.shift)
            s.ss = self.ss
        return s


def plain_src(t, C_):
    s = MsegSrc()
    s.ptr = t.data_ptr()
    s.C = C_
    s.dtype = _st(t)
    return s


class Workspace:
    """Grow-only scratch buffers (the library never allocates; SURVEY.md §8b ownership)."""

    def __init__(self, device):
        self.device = device
        self.buf = {}

    def get(self, name, nbytes, zero=False):
        """`zero`: a buffer whose head holds device-side counters the kernels keep at zero (mseg_norm_*): zero-filled
        whenever it is (re)allocated"""
        b = self.buf.get(name)
        if b is None or b.numel() < nbytes:
            make = torch.zeros if zero else torch.empty
            b = make(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self.buf[name] = b
        return b


# ---- optional HIP-event timing of the MFMA kernels (bench.py roofline leg) --------------------------------------
class KernelTimer:
    """Brackets every launch of the matrix-core kernels with HIP events on the launch stream and keeps the
    algorithmic FLOPs of each launch; summary() resolves the events after a synchronize."""

    def __init__(self):
        self.records = []   # (kernel name, flops, start event, end event)

    def bracket(self, name, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        self.records.append((name, flops, e0, e1))
        return e0, e1

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1 in self.records:
            d = out.setdefault(name, {"kernel": name, "launches": 0, "total_ms": 0.0, "flops": 0.0})
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["launches"]
            d["tflops"] = d["flops"] / (d["total_ms"] * 1e-3) / 1e12 if d["total_ms"] > 0 else 0.0
        return out


_timer = None


def set_kernel_timer(t):
    global _timer
    _timer = t


_precision = "fp32"


def set_precision(mode):
    """'fp32' (default; the reference's arithmetic) or 'bf16' (BASELINE configs[2]): every convolution launch that has a
    bf16 matrix-core kernel (mseg_igemm_query / mseg_wgrad_query answer that) rounds its (normalised) operands to bf16 and
    accumulates in fp32; when ALL launches of a network have one, activations and activation gradients are also STORED as
    bf16 (set_bf16_storage / bf16_storage_ok).  Weights, weight gradients, normalisation statistics, loss and
    optimizer state stay fp32."""
    global _precision
    if mode not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _precision = mode


def get_precision():
    return _precision


class precision_scope:
    """``with engine.precision_scope('bf16'): ...`` — set_precision for a block, restored on exit"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.previous = get_precision()
        set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        set_precision(self.previous)
        return False


_bf16_storage = True


def set_bf16_storage(flag):
    """bf16 mode: also STORE activations / activation gradients as bf16 (default) or keep fp32 tensors in HBM and round
    only the matrix-core operands (round 1 behaviour; kept for ablation and for networks without a full set of bf16 kernels)"""
    global _bf16_storage
    _bf16_storage = bool(flag)


_DUMMY = 256          # a non-null, 16-byte aligned "pointer" for dispatch queries (a query dereferences nothing)


def _dummy_src(c, dtype=ST_BF16):
    s = MsegSrc()
    s.ptr, s.C, s.dtype = _DUMMY, c, dtype
    return s


def _dummy_pack(n, k):
    return (_round_up(n, 128), _round_up(k, 32))


def bf16_storage_ok(spec, N, cin, H, W, training):
    """Can this network run with bf16 tensor storage?  The bf16-storage forms exist for the bf16 matrix-core kernels, the
    first-layer / head / normalisation kernels: every convolution launch of the forward pass (and, when a tape is kept, of
    the backward pass) must map to one of them, channel counts must be multiples of 8 (16-byte staging loads of 8 bf16),
    and the first layer must take the VALU kernels.  Otherwise bf16 mode keeps fp32 tensors (operand rounding only).
    Whether a launch has a bf16 kernel is the LIBRARY's answer (mseg_igemm_query / mseg_wgrad_query on the descriptor the
    launch would carry, with bf16 operands and destinations), not a rule restated here."""
    if spec.pool_method == "max":
        return False

    def ig(cins, ngemm, nb, hi, wi, ho, wo, kh, kw, stride, pad, mode, epi=EPI_PLAIN, morder=MORDER_LINEAR, cq=0, taps=None):
        npad, kpad = _dummy_pack(ngemm, sum(cins))
        return igemm_query([_dummy_src(v) for v in cins], kpad, npad, nb, hi, wi, ho, wo, kh, kw, stride, pad, mode, ngemm,
                           ld0=(cq if epi == EPI_SCATTER2X2 else ngemm), epi=epi, Cq=cq, morder=morder, precision="bf16",
                           dst_dtype=ST_BF16) is not None

    def wg(pc, qcs, nb, hp, wp, hq, wq, kh, kw, stride, pad):
        return wgrad_query(_dummy_src(pc), [_dummy_src(v) for v in qcs], nb, hp, wp, hq, wq, kh, kw, stride, pad,
                           precision="bf16") is not None

    def conv_ok(cins, cout, hi, wi, stride):
        ho, wo = (hi + 2 - 3) // stride + 1, (wi + 2 - 3) // stride + 1
        c = sum(cins)
        if any(v % 8 for v in cins) or cout % 8:
            return False
        if not ig(cins, cout, N, hi, wi, ho, wo, 3, 3, stride, 1, MODE_CONV):
            return False
        if training:
            morder = MORDER_PARITY if stride == 2 else MORDER_LINEAR
            if not ig([cout], c, N, ho, wo, hi, wi, 3, 3, stride, 1, MODE_TCONV, morder=morder):
                return False
            if not wg(cout, cins, N, ho, wo, hi, wi, 3, 3, stride, 1):
                return False
        return True

    def up_ok(ci, co, hi, wi):
        if ci % 8 or co % 8:
            return False
        if not ig([ci], 4 * co, N, hi, wi, hi, wi, 1, 1, 1, 0, MODE_CONV, epi=EPI_SCATTER2X2, cq=co):
            return False
        if training:
            if not ig([co], ci, N, 2 * hi, 2 * wi, hi, wi, 2, 2, 2, 0, MODE_CONV):
                return False
            if not wg(ci, [co], N, hi, wi, 2 * hi, 2 * wi, 2, 2, 2, 0):
                return False
        return True

    h, w = H, W
    skips = []
    c_prev = None
    for i, e in enumerate(spec.enc):
        c1o, c1i = e["c1"].conv.weight.shape[:2]
        if i == 0:
            if not _first_layer_ok(c1i, c1o, 1) or (training and c1i != 1) or c1o % 8:
                return False
        elif not conv_ok([c1i], c1o, h, w, 1):
            return False
        c2o = e["c2"].conv.weight.shape[0]
        if not conv_ok([c1o], c2o, h, w, 1):
            return False
        c_prev = c2o
        if e["pool"] is not None:
            skips.append((c2o, h, w))
            if not conv_ok([c2o], c2o, h, w, 2):
                return False
            h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    for d in spec.decoders:
        ch, hh, ww = c_prev, h, w
        for lv, (cs, hs, wsk) in zip(d["levels"], reversed(skips)):
            ci, co = lv["up"].conv.weight.shape[:2]
            if ci != ch or not up_ok(ci, co, hh, ww):
                return False
            hh, ww = 2 * hh, 2 * ww
            c1o = lv["c1"].conv.weight.shape[0]
            if not conv_ok([co, cs], c1o, hh, ww, 1):
                return False
            c2o = lv["c2"].conv.weight.shape[0]
            if not conv_ok([c1o], c2o, hh, ww, 1):
                return False
            ch = c2o
        if ch % 8:
            return False
    return True


# ---- thin kernel wrappers ------------------------------------------------------------------------------------
def _round_up(v, m):
    return (v + m - 1) // m * m


_splitk_ws = {}      # (device, stream) -> grow-only scratch of the split-K igemm launches (stream-ordered reuse)


class PackedW:
    """GEMM operand [T][Npad][Kpad] (Npad % 128 == 0, Kpad % 32 == 0, zero filled) of one conv weight, kept in a
    PERSISTENT buffer next to the parameter it was packed from (`param._mseg_packs[kind]`) and refreshed only when the
    parameter has changed: torch's version counter of the parameter (every in-place update of an optimizer,
    load_state_dict, ... bumps it) and its data pointer are recorded at pack time.  Inference never repacks; a training
    step repacks each operand once — per layer on first use, or, with training/optim.FusedAdam, all operands of the
    network in ONE launch right after the update (`repack_all`)."""
    __slots__ = ("param", "job", "merge", "t", "Npad", "Kpad", "_t16", "version", "src_ptr")

    def __init__(self, param, T, R, Cc, st, sr, sc, merge_taps):
        self.param = param
        self.merge = merge_taps
        self.Kpad = _round_up(Cc, 32)
        if merge_taps:      # the T blocks of R rows back to back (one GEMM with N = T*R), padded as a whole
            self.Npad = _round_up(T * R, 128)
            rpad = R
            self.t = torch.zeros(self.Npad * self.Kpad, dtype=torch.float32, device=param.device)
        else:
            self.Npad = rpad = _round_up(R, 128)
            self.t = torch.empty(T * self.Npad * self.Kpad, dtype=torch.float32, device=param.device)
        self.job = (T, R, rpad, Cc, self.Kpad, st, sr, sc)
        self._t16 = None
        self.version, self.src_ptr = -1, 0

    def stale(self):
        return self.version != self.param._version or self.src_ptr != self.param.data_ptr()

    def mark_fresh(self):
        self.version, self.src_ptr = self.param._version, self.param.data_ptr()

    def refresh(self):
        lib = _lib.load()
        T, R, rpad, Cc, kpad, st, sr, sc = self.job
        src = self.param.detach()
        check(lib.mseg_pack_weight(src.data_ptr(), self.t.data_ptr(), T, R, rpad, Cc, kpad, st, sr, sc, _stream()),
              "pack_weight")
        if self._t16 is not None:
            check(lib.mseg_f32_to_bf16(self.t.data_ptr(), self._t16.data_ptr(), self.t.numel(), _stream()), "f32_to_bf16")
        self.mark_fresh()

    def bf16(self):
        """the same operand rounded to bf16 (operand of a MSEG_PREC_BF16 launch); created on first use, then kept
        in step with the fp32 operand by refresh() / repack_all()"""
        if self._t16 is None:
            if self.merge:
                self._t16 = torch.zeros(self.t.numel(), dtype=torch.bfloat16, device=self.t.device)
            else:
                self._t16 = torch.empty(self.t.numel(), dtype=torch.bfloat16, device=self.t.device)
            check(_lib.load().mseg_f32_to_bf16(self.t.data_ptr(), self._t16.data_ptr(), self.t.numel(), _stream()),
                  "f32_to_bf16")
        return self._t16


def pack_weight(param, T, R, Cc, st, sr, sc, merge_taps=False, kind=None):
    """dst[(t*Rpad + r)*Kpad + c] = param[t*st + r*sr + c*sc], zero padded.  With `kind` ('fwd' / 'dgrad') the operand is
    cached on the parameter and only refreshed when the parameter has changed; without it a one-off operand is built."""
    if kind is None:
        e = PackedW(param, T, R, Cc, st, sr, sc, merge_taps)
        e.refresh()
        return e
    packs = param.__dict__.get("_mseg_packs")
    if packs is None:
        packs = param.__dict__["_mseg_packs"] = {}
    e = packs.get(kind)
    if e is None:
        e = packs[kind] = PackedW(param, T, R, Cc, st, sr, sc, merge_taps)
    if e.stale():
        e.refresh()
    return e


_pack_tables = {}    # tuple of job keys -> (device table, njobs, total blocks, keep-alive)


def repack_all(params):
    """Refresh every packed operand of `params` in one launch (mseg_pack_weights_multi).  Called by FusedAdam.step()
    after the update kernel; operands that do not exist yet (first step) are packed lazily by their first consumer."""
    entries = []
    for p in params:
        packs = p.__dict__.get("_mseg_packs")
        if packs:
            entries += [e for _, e in sorted(packs.items())]
    if not entries:
        return 0
    key = tuple((id(e), e.param.data_ptr(), e.t.data_ptr(), 0 if e._t16 is None else e._t16.data_ptr(), e.job)
                for e in entries)         # (the geometry too: ids and allocator addresses are reused across networks)
    tab = _pack_tables.get(key)
    if tab is None:
        _pack_tables.clear()                        # one live network per process is the common case
        jobs = (_lib.MsegPackJob * len(entries))()
        nblk = 0
        for j, e in zip(jobs, entries):
            T, R, rpad, Cc, kpad, st, sr, sc = e.job
            j.src, j.dst = e.param.data_ptr(), e.t.data_ptr()
            j.dst16 = None if e._t16 is None else e._t16.data_ptr()
            j.T, j.R, j.Rpad, j.C, j.Cpad, j.st, j.sr, j.sc = T, R, rpad, Cc, kpad, st, sr, sc
            j.first_block = nblk
            nblk += _lib.load().mseg_pack_job_blocks(T, rpad, kpad)
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        tab = (raw.to(entries[0].t.device), len(entries), nblk)
        _pack_tables[key] = tab
    check(_lib.load().mseg_pack_weights_multi(tab[0].data_ptr(), tab[1], tab[2], _stream()), "pack_weights_multi")
    for e in entries:
        e.mark_fresh()
    return len(entries)


def mark_packs_fresh(params):
    """the packed operands of `params` were refreshed on the device by a replayed hipGraph (its recorded repack_all)"""
    for p in params:
        packs = p.__dict__.get("_mseg_packs")
        if packs:
            for e in packs.values():
                e.mark_fresh()


class KernelChoice:
    """the library's answer to "which kernel does this launch take" (MsegKernelInfo)"""
    __slots__ = ("name", "bf16", "launches", "grid", "block", "workspace", "stats_rows")

    def __init__(self, info):
        self.name = info.name.decode()
        self.bf16 = info.precision == 1
        self.launches, self.grid, self.block, self.workspace = info.launches, info.grid, info.block, info.workspace
        self.stats_rows = info.stats_rows

    def __repr__(self):
        return f"KernelChoice({self.name!r}, bf16={self.bf16}, grid={self.grid})"


def _fill_igemm(p, srcs, kpad, npad, NB, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode, Ngemm, ld0, acc0, ld1, acc1, split, epi,
                Cq, morder):
    for i, s in enumerate(srcs):
        p.src[i] = s
    p.nsrc = len(srcs)
    p.Cin = sum(s.C for s in srcs)
    p.Kpad, p.Npad = kpad, npad
    p.NB, p.Hi, p.Wi, p.Ho, p.Wo = NB, Hi, Wi, Ho, Wo
    p.KH, p.KW, p.stride, p.pad, p.mode, p.morder = KH, KW, stride, pad, mode, morder
    p.Ngemm, p.epi = Ngemm, epi
    p.split = Ngemm if split is None else split
    p.ld0, p.ld1, p.acc0, p.acc1, p.Cq = ld0, ld1, acc0, acc1, Cq


def _query(fn, p):
    info = MsegKernelInfo()
    rc = fn(C.byref(p), C.byref(info))
    return KernelChoice(info) if rc == 0 else None


def igemm_query(srcs, kpad, npad, NB, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode, Ngemm, ld0, acc0=0, ld1=0, acc1=0,
                split=None, epi=EPI_PLAIN, Cq=0, morder=MORDER_LINEAR, precision=None, dst_dtype=ST_F32, bias=True):
    """mseg_igemm_query for a launch described by its shapes (placeholder pointers): the kernel the library would take, or
    None when it has none for this descriptor (e.g. precision 'bf16' on a shape without a bf16 kernel)."""
    p = MsegIgemm()
    _fill_igemm(p, srcs, kpad, npad, NB, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode, Ngemm, ld0, acc0, ld1, acc1, split, epi, Cq,
                morder)
    p.w = p.dst0 = _DUMMY
    p.dst1 = _DUMMY if (split is not None and split < Ngemm) else None
    p.bias = _DUMMY if bias else None
    p.precision = 1 if (precision or _precision) == "bf16" else 0
    p.dst_dtype = dst_dtype
    return _query(_lib.load().mseg_igemm_query, p)


_bf16_ok_cache = {}     # launch signature -> does the library have a bf16 kernel for it (toggles of the ablation hooks
                        # switch between kernels of one precision, never the answer to this)


def _igemm_bf16_ok(p):
    key = (p.nsrc, tuple((p.src[i].C, p.src[i].act, bool(p.src[i].scale), p.src[i].ss != 0, p.src[i].dtype)
                         for i in range(p.nsrc)), p.Kpad, p.Npad, p.NB, p.Hi, p.Wi, p.Ho, p.Wo, p.KH, p.KW, p.stride, p.pad,
           p.mode, p.morder, p.Ngemm, p.epi, p.split, p.ld0, p.ld1, p.acc0, p.acc1, p.Cq, p.dst_dtype, bool(p.bias))
    ok = _bf16_ok_cache.get(key)
    if ok is None:
        p.precision = 1
        ok = _bf16_ok_cache[key] = _query(_lib.load().mseg_igemm_query, p) is not None
    return ok


# BatchNorm statistics taken in the producing convolution's epilogue where its kernel can (MsegIgemm.stats, igemm_p8.hip)
# instead of by a pass over the stored tensor.  Measured round 3 (bf16, 320x320, batch 32, one box, rocprofv3): the 20 layers
# concerned lose 3.85 ms of statistics passes per 7 steps and their convolutions gain 4.6 ms (the epilogue of the
# one-workgroup-per-CU kernel is on its critical path: + 6 % on 256 x 256 tiles, + 21 % on 512 x 128) — a wash, so it is
# off by default; MSEG_CONV_STATS=1 / set_conv_stats(True) turn it on.
_conv_stats = os.environ.get("MSEG_CONV_STATS", "0") == "1"


def set_conv_stats(flag):
    global _conv_stats
    old, _conv_stats = _conv_stats, bool(flag)
    return old


def igemm(srcs, w, bias, NB, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode, Ngemm, dst0, ld0, acc0=0,
          dst1=None, ld1=0, acc1=0, split=None, epi=EPI_PLAIN, Cq=0, morder=MORDER_LINEAR, real_cin=None,
          precision=None, stats=None):
    """`stats`: None, or (activation of the output, Workspace) — the caller would like the statistics of act(output) from
    this launch.  Returns None, or (partial-sum tensor, rows) when the kernel the library chose has taken them."""
    lib = _lib.load()
    p = MsegIgemm()
    _fill_igemm(p, srcs, w.Kpad, w.Npad, NB, Hi, Wi, Ho, Wo, KH, KW, stride, pad, mode, Ngemm, ld0, acc0, ld1, acc1, split, epi,
                Cq, morder)
    p.w = w.t.data_ptr()
    p.dst_dtype = _st(dst0)                 # bf16 tensor storage: the C ABI accepts it for the bf16 kernels only
    p.bias = _ptr(bias)
    p.dst0 = dst0.data_ptr()
    p.dst1 = _ptr(dst1)
    # bf16 mode: the launch runs on the bf16 matrix cores iff the library has a bf16 kernel for it (its dispatch, queried)
    bf16 = (precision or _precision) == "bf16" and _igemm_bf16_ok(p)
    if bf16:
        p.w = w.bf16().data_ptr()
    p.precision = 1 if bf16 else 0
    need = lib.mseg_igemm_workspace_bytes(C.byref(p))      # split-K scratch (small batches / deep levels only)
    if need:
        key = (dst0.device, _stream())                  # one scratch per stream: launches of two streams may overlap
        buf = _splitk_ws.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(need, dtype=torch.uint8, device=dst0.device)
            _splitk_ws[key] = buf
        p.ws, p.ws_bytes = buf.data_ptr(), buf.numel()
    taken = None
    if stats is not None and bf16 and _conv_stats:
        # ask the dispatch (the same code the launch runs) whether its kernel for this descriptor takes statistics
        p.stats, p.stats_act = _DUMMY, stats[0]
        choice = _query(lib.mseg_igemm_query, p)
        rows = choice.stats_rows if choice else 0
        if rows > 0:
            part = stats[1].get("convstats", rows * 2 * Ngemm * 4)
            p.stats = part.data_ptr()
            taken = (part, rows)
        else:
            p.stats = None
    if _timer is None:
        check(lib.mseg_igemm(C.byref(p), _stream()), "igemm")
        return taken
    flops = 2.0 * NB * Ho * Wo * Ngemm * (p.Cin if real_cin is None else real_cin) * KH * KW
    if mode == MODE_TCONV:
        flops /= stride * stride
    choice = _query(lib.mseg_igemm_query, p)               # the kernel's name as rocprofv3 prints it
    e0, e1 = _timer.bracket(choice.name if choice else "igemm?", flops)
    e0.record()
    check(lib.mseg_igemm(C.byref(p), _stream()), "igemm")
    e1.record()
    return taken


def _fill_wgrad(p, P, Qs, NB, Hp, Wp, Hq, Wq, KH, KW, stride, pad, nch_store):
    p.P = P
    for i, s in enumerate(Qs):
        p.Q[i] = s
    p.nq = len(Qs)
    p.Nch = sum(s.C for s in Qs)
    p.Nch_store = p.Nch if nch_store is None else nch_store
    p.NB, p.Hp, p.Wp, p.Hq, p.Wq = NB, Hp, Wp, Hq, Wq
    p.KH, p.KW, p.stride, p.pad = KH, KW, stride, pad
    p.splits = 0


def wgrad_query(P, Qs, NB, Hp, Wp, Hq, Wq, KH, KW, stride, pad, nch_store=None, precision=None):
    """mseg_wgrad_query: the weight-gradient kernel the library would take for these operands, or None."""
    p = MsegWgrad()
    _fill_wgrad(p, P, Qs, NB, Hp, Wp, Hq, Wq, KH, KW, stride, pad, nch_store)
    p.precision = 1 if (precision or _precision) == "bf16" else 0
    p.dst = p.ws = _DUMMY
    p.phase = 1
    return _query(_lib.load().mseg_wgrad_query, p)


def _wgrad_bf16_ok(p):
    key = ((p.P.C, p.P.act, bool(p.P.scale), p.P.ss != 0, p.P.dtype),
           tuple((p.Q[i].C, p.Q[i].act, bool(p.Q[i].scale), p.Q[i].ss != 0, p.Q[i].dtype) for i in range(p.nq)),
           p.Nch_store, p.NB, p.Hp, p.Wp, p.Hq, p.Wq, p.KH, p.KW, p.stride, p.pad)
    ok = _bf16_ok_cache.get(key)
    if ok is None:
        q = MsegWgrad()
        C.memmove(C.byref(q), C.byref(p), C.sizeof(MsegWgrad))
        q.precision, q.phase = 1, 1
        q.dst = q.ws = _DUMMY
        ok = _bf16_ok_cache[key] = _query(_lib.load().mseg_wgrad_query, q) is not None
    return ok


def wgrad(P, Qs, dst, NB, Hp, Wp, Hq, Wq, KH, KW, stride, pad, ws, nch_store=None, precision=None):
    lib = _lib.load()
    p = MsegWgrad()
    _fill_wgrad(p, P, Qs, NB, Hp, Wp, Hq, Wq, KH, KW, stride, pad, nch_store)
    bf16 = (precision or _precision) == "bf16" and _wgrad_bf16_ok(p)
    p.precision = 1 if bf16 else 0
    p.dst = dst.data_ptr()
    need = lib.mseg_wgrad_workspace_bytes(C.byref(p))
    if need == 0:
        raise RuntimeError("libmseg_hip wgrad: unsupported shape")
    p.ws = ws.get("wgrad", need).data_ptr()
    if _timer is None:
        check(lib.mseg_wgrad(C.byref(p), _stream()), "wgrad")
        return
    flops = 2.0 * NB * Hp * Wp * P.C * p.Nch_store * KH * KW
    p.phase = 1     # split-K partial kernel only (timed) ...
    choice = _query(lib.mseg_wgrad_query, p)
    e0, e1 = _timer.bracket(choice.name if choice else "wgrad?", flops)
    e0.record()
    check(lib.mseg_wgrad(C.byref(p), _stream()), "wgrad")
    e1.record()
    p.phase = 2     # ... then the fixed-order reduction
    check(lib.mseg_wgrad(C.byref(p), _stream()), "wgrad_reduce")


_stats_epoch = 0


def note_training_step():
    """running BatchNorm statistics may have changed (a training forward ran, eagerly or inside a replayed graph)"""
    global _stats_epoch
    _stats_epoch += 1


def norm_stats(node, norm, gamma, beta, running_mean, running_var, training, ws, conv_part=None):
    """Fill node.scale/shift (+mean/rstd) from a = act(z); BatchNorm eval mode uses the running statistics.
    `conv_part`: (partial sums, rows) left by the producing convolution's epilogue (training BatchNorm): the tensor is not
    read again."""
    lib = _lib.load()
    dev = node.z.device
    node.norm = norm
    N, HW, Cc = node.N, node.H * node.W, node.C
    expensive = node.act not in (ACT["none"], ACT["relu"])
    if expensive:
        node.a = torch.empty_like(node.z)
    if norm == NORM["bn"]:
        node.ss = 0
        if not training:
            # eval-mode tables depend on the layer's parameters and running statistics only: computed once and kept on the
            # running_mean buffer until one of the four tensors changes (a frame of a stack re-used 34 tiny launches)
            key = (_stats_epoch,) + tuple((t.data_ptr(), t._version) if t is not None else None
                                          for t in (gamma, beta, running_mean, running_var))
            cached = running_mean.__dict__.get("_mseg_eval")
            if cached is None or cached[0] != key:
                scale = torch.empty(Cc, dtype=torch.float32, device=dev)
                shift = torch.empty(Cc, dtype=torch.float32, device=dev)
                check(lib.mseg_bn_eval_coeffs(_ptr(gamma), _ptr(beta), running_mean.data_ptr(), running_var.data_ptr(),
                                              BN_EPS, Cc, scale.data_ptr(), shift.data_ptr(), _stream()), "bn_eval_coeffs")
                cached = (key, scale, shift)
                running_mean.__dict__["_mseg_eval"] = cached
            node.scale, node.shift = cached[1], cached[2]
            if expensive:
                check(lib.mseg_activation(node.z.data_ptr(), N, HW, Cc, _st(node.z), node.act, node.a.data_ptr(),
                                          _stream()), "activation")
            return
        node.scale = torch.empty(Cc, dtype=torch.float32, device=dev)
        node.shift = torch.empty(Cc, dtype=torch.float32, device=dev)
        node.mean = torch.empty(Cc, dtype=torch.float32, device=dev)
        node.rstd = torch.empty(Cc, dtype=torch.float32, device=dev)
    else:
        node.ss = Cc
        node.scale = torch.empty(N * Cc, dtype=torch.float32, device=dev)
        node.shift = torch.empty(N * Cc, dtype=torch.float32, device=dev)
        ng = 8 if norm == NORM["gn"] else Cc
        node.mean = torch.empty(N * ng, dtype=torch.float32, device=dev)
        node.rstd = torch.empty(N * ng, dtype=torch.float32, device=dev)
        running_mean = running_var = None
    w = ws.get("norm", lib.mseg_norm_workspace_bytes(N, HW, Cc), zero=True)
    if conv_part is not None and norm == NORM["bn"] and not expensive:
        check(lib.mseg_norm_stats_from_conv(conv_part[0].data_ptr(), conv_part[1], Cc, N * HW, _ptr(gamma), _ptr(beta),
                                            BN_EPS, node.scale.data_ptr(), node.shift.data_ptr(), node.mean.data_ptr(),
                                            node.rstd.data_ptr(), _ptr(running_mean), _ptr(running_var), BN_MOMENTUM,
                                            w.data_ptr(), _stream()), "norm_stats_from_conv")
    else:
        check(lib.mseg_norm_stats(node.z.data_ptr(), N, HW, Cc, _st(node.z), node.act, norm, _ptr(gamma), _ptr(beta),
                                  BN_EPS, node.scale.data_ptr(), node.shift.data_ptr(), node.mean.data_ptr(),
                                  node.rstd.data_ptr(), _ptr(running_mean), _ptr(running_var), BN_MOMENTUM,
                                  _ptr(node.a), w.data_ptr(), _stream()), "norm_stats")
    if running_mean is not None:
        # the kernel updated the running statistics through raw pointers: tell torch, and start a new epoch of the eval-mode
        # table cache above (a recorded step replayed by a hipGraph runs no Python: graph_step.py calls note_training_step)
        torch.autograd.graph.increment_version((running_mean, running_var))
        note_training_step()


def norm_bwd(node, gy, gamma, dgamma, dbeta, dbias, ws):
    """gy (dL/d normalised output) -> dz in place; returns the same tensor holding dz."""
    lib = _lib.load()
    N, HW, Cc = node.N, node.H * node.W, node.C
    w = ws.get("norm", lib.mseg_norm_workspace_bytes(N, HW, Cc), zero=True)
    check(lib.mseg_norm_bwd(gy.data_ptr(), node.z.data_ptr(), N, HW, Cc, _st(node.z), node.act, node.norm, _ptr(gamma),
                            node.mean.data_ptr(), node.rstd.data_ptr(), gy.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                            _ptr(dbias), _ptr(node.a), w.data_ptr(), _stream()), "norm_bwd")
    return gy


# ---- layer descriptions (built by utils/unets.py from the nn.Module tree) ------------------------------------
class ConvSpec:
    """conv (3x3 s1 | 3x3 s2 | convT 2x2 s2) -> activation -> norm, parameters held by stock torch modules."""

    def __init__(self, kind, conv, norm_mod, act, norm):
        self.kind = kind            # 'conv' | 'pool' | 'up'
        self.conv = conv            # nn.Conv2d / nn.ConvTranspose2d (parameter holder)
        self.norm_mod = norm_mod    # nn.BatchNorm2d / nn.GroupNorm / nn.InstanceNorm2d
        self.act = ACT[act]
        self.norm = NORM[norm]

    def params(self):
        ps = [self.conv.weight, self.conv.bias]
        if self.norm in (NORM["bn"], NORM["gn"]):
            ps += [self.norm_mod.weight, self.norm_mod.bias]
        return ps


class HeadSpec:
    def __init__(self, conv):
        self.conv = conv

    def params(self):
        return [self.conv.weight, self.conv.bias]


class NetSpec:
    def __init__(self, ch_in, enc, decoders, pool_method):
        self.ch_in = ch_in
        self.enc = enc              # list of dict(c1=ConvSpec, c2=ConvSpec, pool=ConvSpec|None)
        self.decoders = decoders    # list (1 for U, 2 for DU) of dict(levels=[dict(up,c1,c2)], head=HeadSpec)
        self.pool_method = pool_method

    def layers(self):
        out = []
        for e in self.enc:
            out += [e["c1"], e["c2"]]
            if e["pool"] is not None and e["pool"] != "max":
                out.append(e["pool"])
        for d in self.decoders:
            for lv in d["levels"]:
                out += [lv["up"], lv["c1"], lv["c2"]]
            out.append(d["head"])
        return out

    def params(self):
        ps = []
        for l in self.layers():
            ps += l.params()
        return ps


# ---- forward ---------------------------------------------------------------------------------------------------
class Tape:
    """What the backward needs: nodes in execution order + head records."""

    def __init__(self):
        self.nodes = []
        self.heads = []
        self.x4 = None


class RawFrame:
    """One raw uint8 / uint16 frame on the device, to be normalised and padded by the network's first kernel (K14 on the
    device: reference infer.py:346-348 + utils.py:124-163).  `minmax`: int32[2] device tensor filled by `reduce_minmax`."""
    PIX = {torch.uint8: 0, torch.uint16: 1, torch.int16: 1}      # (int16 storage of uint16 data: same bits)

    def __init__(self, raw, pad_top=0, pad_left=0):
        if raw.dim() != 2 or raw.dtype not in self.PIX or not raw.is_cuda or not raw.is_contiguous():
            raise RuntimeError("RawFrame: a contiguous 2-D uint8 / uint16 CUDA tensor expected")
        self.raw, self.pad_top, self.pad_left = raw, int(pad_top), int(pad_left)
        self.minmax = torch.empty(2, dtype=torch.int32, device=raw.device)
        self.H, self.W = raw.shape[0] + self.pad_top, raw.shape[1] + self.pad_left
        check(_lib.load().mseg_frame_minmax(raw.data_ptr(), self.PIX[raw.dtype], raw.numel(), self.minmax.data_ptr(),
                                            _stream()), "frame_minmax")

    @property
    def device(self):
        return self.raw.device

    def args(self):
        return (self.raw.data_ptr(), self.PIX[self.raw.dtype], self.raw.shape[0], self.raw.shape[1], self.pad_top,
                self.pad_left, self.minmax.data_ptr())

    def normalized(self):
        """the padded, normalised frame as an fp32 tensor (1, 1, H, W): first layers without the fused kernel, tests"""
        out = torch.empty((1, 1, self.H, self.W), dtype=torch.float32, device=self.raw.device)
        check(_lib.load().mseg_frame_normalize(*self.args(), out.data_ptr(), _stream()), "frame_normalize")
        return out


def _first_layer_ok(cin, cout, stride):
    return stride == 1 and 1 <= cin <= 4 and cout % 4 == 0 and cout <= 256 and 256 % (cout // 4) == 0


def _run_conv(spec, in_nodes, training, ws, tape, first_layer_cin=None, st=torch.float32):
    """`st`: storage type of the layer's output z (torch.float32, or torch.bfloat16 in bf16-storage mode)"""
    conv = spec.conv
    n0 = in_nodes[0]
    N, Hi, Wi = n0.N, n0.H, n0.W
    dev = n0.z.device
    conv_part = None          # (partial sums, rows) when the convolution's epilogue has taken the BatchNorm statistics
    if isinstance(n0.z, RawFrame):
        srcs = []
    cin_total = sum(n.C for n in in_nodes)
    if not isinstance(n0.z, RawFrame):
        srcs = [n.src() for n in in_nodes]
    bias = conv.bias.detach()
    wt = conv.weight.detach()
    if spec.kind == "up":
        cin, cout = wt.shape[0], wt.shape[1]
        assert cin == cin_total
        wp = pack_weight(conv.weight, 4, cout, cin, 1, 4, cout * 4, merge_taps=True, kind="fwd")
        Ho, Wo = 2 * Hi, 2 * Wi
        z = torch.empty((N, Ho, Wo, cout), dtype=st, device=dev)
        igemm(srcs, wp, bias, N, Hi, Wi, Hi, Wi, 1, 1, 1, 0, MODE_CONV, 4 * cout, z, cout,
              epi=EPI_SCATTER2X2, Cq=cout)
        act = ACT["none"]
    else:
        cout, cin = wt.shape[0], wt.shape[1]
        stride = 2 if spec.kind == "pool" else 1
        if first_layer_cin is None:
            assert cin == cin_total, (cin, cin_total)
        else:
            assert _pad4(cin) == cin_total, (cin, cin_total)   # network input is zero-padded to 4 channels
        Ho, Wo = (Hi + 2 - 3) // stride + 1, (Wi + 2 - 3) // stride + 1
        z = torch.empty((N, Ho, Wo, cout), dtype=st, device=dev)
        if first_layer_cin is not None and isinstance(n0.z, RawFrame):
            # inference on a raw frame: normalisation and padding happen in this kernel's loads
            check(_lib.load().mseg_first_conv_fwd_raw(*n0.z.args(), wt.contiguous().data_ptr(), bias.data_ptr(), cout,
                                                      z.data_ptr(), _st(z), _stream()), "first_conv_fwd_raw")
        elif first_layer_cin is not None and _first_layer_ok(cin, cout, stride):
            # raw network input, 9..36 MACs per output: HBM-bound VALU kernel instead of a 32-channel MFMA K-step
            check(_lib.load().mseg_first_conv_fwd(n0.z.data_ptr(), wt.contiguous().data_ptr(), bias.data_ptr(), N, Hi,
                                                  Wi, cin, cout, z.data_ptr(), _st(z), _stream()), "first_conv_fwd")
        else:
            wp = pack_weight(conv.weight, 9, cout, cin, 1, cin * 9, 9, kind="fwd")
            want = (spec.act, ws) if (training and spec.norm == NORM["bn"] and
                                      spec.act in (ACT["none"], ACT["relu"])) else None
            conv_part = igemm(srcs, wp, bias, N, Hi, Wi, Ho, Wo, 3, 3, stride, 1, MODE_CONV, cout, z, cout, real_cin=cin,
                              stats=want)
        act = spec.act
    node = Node(z, N, Ho, Wo, cout)
    node.act = act
    node.layer = spec
    node.inputs = tuple(in_nodes)
    nm = spec.norm_mod
    if spec.norm == NORM["bn"]:
        norm_stats(node, spec.norm, nm.weight.detach(), nm.bias.detach(), nm.running_mean, nm.running_var,
                   training, ws, conv_part=conv_part)
        if training:
            _nbt_pending.append(nm.num_batches_tracked)    # all layers' counters advance in ONE launch (end of forward)
    elif spec.norm == NORM["gn"]:
        norm_stats(node, spec.norm, nm.weight.detach(), nm.bias.detach(), None, None, training, ws)
    else:
        norm_stats(node, spec.norm, None, None, None, None, training, ws)
    if tape is not None:
        tape.nodes.append(node)
    return node


class _MaxPoolMarker:
    kind = "maxpool"


def _run_maxpool(src_node, tape):
    """nn.MaxPool2d(2, 2) on the normalised operand; the pooled tensor is stored plain (no pending transform)."""
    lib = _lib.load()
    N, H, W, Cc = src_node.N, src_node.H, src_node.W, src_node.C
    out = torch.empty((N, H // 2, W // 2, Cc), dtype=torch.float32, device=src_node.z.device)
    s = src_node.src()
    check(lib.mseg_maxpool2x2_fwd(C.byref(s), N, H, W, out.data_ptr(), _stream()), "maxpool_fwd")
    node = Node(out, N, H // 2, W // 2, Cc)
    node.layer = _MaxPoolMarker
    node.inputs = (src_node,)
    if tape is not None:
        tape.nodes.append(node)
    return node


_nbt_pending = []      # BatchNorm num_batches_tracked tensors of the training forward in progress


def forward(spec, x, training, keep_tape, ws):
    """x: (N, ch_in, H, W) fp32 CUDA tensor.  Returns (list of NCHW outputs, Tape|None)."""
    lib = _lib.load()
    del _nbt_pending[:]
    raw = x if isinstance(x, RawFrame) else None
    if raw is not None:
        # a raw frame (inference): fused into the first convolution when that layer takes the row-walking VALU kernel,
        # else normalised into an fp32 tensor by one streaming kernel
        c1 = spec.enc[0]["c1"].conv.weight
        fused = (not keep_tape and c1.shape[1] == 1 and c1.shape[0] % 8 == 0 and c1.shape[0] <= 256 and
                 256 % (c1.shape[0] // 8) == 0)
        if not fused:
            x, raw = raw.normalized(), None
    if raw is not None:
        N, cin, H, W = 1, 1, raw.H, raw.W
    else:
        N, cin, H, W = x.shape
    nlev = len(spec.enc)
    div = 2 ** (nlev - 1)
    if H % div or W % div:
        raise RuntimeError(f"input {H}x{W} is not divisible by {div}")
    dev = x.device
    c4 = _pad4(cin)
    if raw is not None:
        xin = Node(raw, N, H, W, c4)
    else:
        x4 = torch.zeros((N, H, W, c4), dtype=torch.float32, device=dev)
        x4[..., :cin] = x.detach().permute(0, 2, 3, 1)
        xin = Node(x4, N, H, W, c4)
    tape = Tape() if keep_tape else None
    if tape is not None:
        tape.x4 = xin
    # bf16 mode: activations and their gradients are STORED as bf16 when every launch of this network has a bf16 kernel
    st = torch.bfloat16 if (_precision == "bf16" and _bf16_storage and
                            bf16_storage_ok(spec, N, cin, H, W, keep_tape)) else torch.float32

    cur = xin
    skips = []
    for i, e in enumerate(spec.enc):
        cur = _run_conv(e["c1"], [cur], training, ws, tape, first_layer_cin=cin if i == 0 else None, st=st)
        cur = _run_conv(e["c2"], [cur], training, ws, tape, st=st)
        if e["pool"] is not None:
            skips.append(cur)
            if e["pool"] == "max":
                cur = _run_maxpool(cur, tape)
            else:
                cur = _run_conv(e["pool"], [cur], training, ws, tape, st=st)
    bottom = cur
    outs = []
    # The decoders of a DU-Net are independent of each other: in bf16 mode the second one runs on its own stream, so that the
    # HBM-bound statistics passes and the store bursts of one decoder lie under the matrix kernels of the other.  Its tensors
    # are allocated on that stream; nothing of them is freed before the streams have joined (tape / outputs), and every
    # region opens with the side stream waiting for the main one.
    dec_stream = None
    if _decoder_overlap and _precision == "bf16" and len(spec.decoders) > 1 and dev.type == "cuda" and \
            not torch.cuda.is_current_stream_capturing():
        dec_stream = _decoder_side_stream(dev)
        main = torch.cuda.current_stream(dev)

    def run_decoder(d, ws_):
        cur = bottom
        for lv, skip in zip(d["levels"], reversed(skips)):
            up = _run_conv(lv["up"], [cur], training, ws_, tape, st=st)
            cur = _run_conv(lv["c1"], [up, skip], training, ws_, tape, st=st)
            cur = _run_conv(lv["c2"], [cur], training, ws_, tape, st=st)
        hc = d["head"].conv
        co = hc.weight.shape[0]
        out = torch.empty((N, co, H, W), dtype=torch.float32, device=dev)
        s = cur.src()
        check(lib.mseg_head_fwd(C.byref(s), N, H * W, hc.weight.detach().data_ptr(), hc.bias.detach().data_ptr(), co,
                                out.data_ptr(), _stream()), "head_fwd")
        if tape is not None:
            tape.heads.append((d["head"], cur))
        return out

    for k, d in enumerate(spec.decoders):
        if dec_stream is not None and k == len(spec.decoders) - 1:
            # (the LAST decoder goes aside, so that the tape keeps the order the backward pass walks)
            dec_stream.wait_stream(main)
            with torch.cuda.stream(dec_stream):
                outs.append(run_decoder(d, _SideWs(ws, "@dec")))
        else:
            outs.append(run_decoder(d, ws))
    if dec_stream is not None:
        main.wait_stream(dec_stream)
    if _nbt_pending:
        torch._foreach_add_(_nbt_pending, 1)
        del _nbt_pending[:]
    return outs, tape


# ---- backward --------------------------------------------------------------------------------------------------
def _accumulate_target(node, shape_like):
    """Return (tensor, acc_flag) for writing a gradient contribution into node.grad."""
    if node.grad is None:
        node.grad = torch.empty_like(node.z)
        return node.grad, 0
    return node.grad, 1


def _grad_buf(param, direct):
    """Where the gradient of `param` is written.  A parameter owned by training/optim.FusedAdam carries a persistent
    `.grad` view of the optimizer's gradient arena: the kernels write into it directly, once per zero_grad() (`direct`
    collects those parameters; autograd then gets None for them, i.e. it neither copies nor accumulates).  Otherwise a
    fresh tensor that autograd accumulates into `.grad` as usual."""
    d = param.__dict__
    if d.get("_mseg_grad_direct") and d.get("_mseg_grad_fresh") and param.grad is not None:
        d["_mseg_grad_fresh"] = False
        direct.add(id(param))
        return param.grad
    return torch.empty_like(param)


_wgrad_overlap = os.environ.get("MSEG_WGRAD_OVERLAP", "1") != "0"
_side_streams = {}


def set_wgrad_overlap(flag):
    """Weight gradients on a second HIP stream (default on; MSEG_WGRAD_OVERLAP=0 in the environment turns it off)."""
    global _wgrad_overlap
    _wgrad_overlap = bool(flag)


def get_wgrad_overlap():
    return _wgrad_overlap


# inside a hipGraph capture the side stream joins the capture at its first wait and leaves it at the join of backward():
# the recorded step keeps the two branches
_overlap_in_capture = os.environ.get("MSEG_OVERLAP_IN_CAPTURE", "0") == "1"


# measured round 3 (bf16 320x320 batch 32): 39.5-39.7 ms/step with it, 39.3-39.4 without — every launch of this path already
# fills the chip, two streams only interleave them.  Off by default; kept as a switch for smaller crops.
_decoder_overlap = os.environ.get("MSEG_DECODER_OVERLAP", "0") == "1"


def set_decoder_overlap(flag):
    """The second decoder of a DU-Net on its own stream in the forward pass (bf16 mode; MSEG_DECODER_OVERLAP=1: on)."""
    global _decoder_overlap
    _decoder_overlap = bool(flag)


def get_decoder_overlap():
    return _decoder_overlap


def _decoder_side_stream(dev):
    s = _side_streams.get((dev, "dec"))
    if s is None:
        s = _side_streams[(dev, "dec")] = torch.cuda.Stream(device=dev)
    return s


_SIDE_HOLD_BYTES = 16 << 30    # activation gradients kept alive for the side stream before the streams join early


def _wgrad_side_stream(dev):
    s = _side_streams.get(dev)
    if s is None:
        # lowest priority: the data-gradient chain on the main stream gets the CUs first, the weight gradients fill in
        s = _side_streams[dev] = torch.cuda.Stream(device=dev, priority=0)
    return s


def backward(spec, tape, grad_outs, ws, on_grads=None, direct=None):
    """grad_outs: list of NCHW gradients (None allowed) matching the forward outputs.  Returns {id(param): grad}.

    The weight gradient of a layer depends on that layer's dz only and feeds nothing but the optimizer, while the chain
    dz -> data gradient -> next layer's normalisation backward is what the rest of the pass waits for.  The weight-gradient
    kernels (matrix-core bound) therefore run on a SECOND stream, behind an event on the dz they read: they fill the
    partial last rounds of the persistent data-gradient kernels and run under the HBM-bound normalisation passes of the
    layers that follow; the streams join at the end of the pass.  Same kernels, same arguments, same results.  Not used
    while a hipGraph is being captured or when gradients are handed to a data-parallel bucketer as they are produced."""
    lib = _lib.load()
    grads = {}
    if direct is None:
        direct = set()
    dev = tape.nodes[0].z.device
    side = None
    # (bf16 mode only: there the step has long HBM-bound stretches next to short matrix kernels; the fp32 step is matrix-core
    # bound almost everywhere and measured 0.8 % slower with the second stream)
    if _wgrad_overlap and _precision == "bf16" and on_grads is None and dev.type == "cuda" and \
            (_overlap_in_capture or not torch.cuda.is_current_stream_capturing()):
        side = _wgrad_side_stream(dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)                           # nothing of this pass runs ahead of what is already queued

    held, held_bytes = [], [0]

    def on_side(dz, fn):
        """run the launches of `fn` on the side stream once everything queued so far (dz included) is done"""
        if side is None:
            return fn(ws)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            fn(_SideWs(ws))
        # dz is read by the side stream after the main stream is done with it: it is kept alive until the streams have
        # joined and freed then — on the main stream, behind the join — instead of being marked with record_stream().  A
        # marked block returns to the allocator only once the side stream's work is seen finished; with the host a step
        # ahead of the device the allocator then kept growing its pool inside steady-state steps (hipMalloc calls, each a
        # device synchronisation: 112 of them in 10 timed bf16 steps, and one run in a few with the host stalled for 150 ms
        # per step).  Cost: the activation gradients of one pass stay allocated until its end (2.6 GB at 32 x 320 x 320
        # in bf16); beyond _SIDE_HOLD_BYTES the streams join early.
        held.append(dz)
        held_bytes[0] += dz.numel() * dz.element_size()
        if held_bytes[0] > _SIDE_HOLD_BYTES:
            main.wait_stream(side)
            held.clear()
            held_bytes[0] = 0

    # heads
    for (head, node), go in zip(tape.heads, grad_outs):
        hc = head.conv
        co = hc.weight.shape[0]
        if go is None:
            go = torch.zeros((node.N, co, node.H, node.W), dtype=torch.float32, device=node.z.device)
        go = go.contiguous().float()
        gy = torch.empty_like(node.z)
        dW = _grad_buf(hc.weight, direct)
        db = _grad_buf(hc.bias, direct)
        w = ws.get("head", lib.mseg_head_bwd_workspace_bytes(node.N, node.H * node.W, node.C, co))
        s = node.src()
        check(lib.mseg_head_bwd(C.byref(s), node.N, node.H * node.W, hc.weight.detach().data_ptr(), co,
                                go.data_ptr(), gy.data_ptr(), _st(gy), dW.data_ptr(), db.data_ptr(), w.data_ptr(),
                                _stream()), "head_bwd")
        assert node.grad is None
        node.grad = gy
        grads[id(hc.weight)] = dW
        grads[id(hc.bias)] = db
        if on_grads is not None:
            on_grads([dW, db])

    for node in reversed(tape.nodes):
        sp = node.layer
        gy = node.grad
        node.grad = None
        if gy is None:
            raise RuntimeError("internal: node without gradient")
        if sp is _MaxPoolMarker:
            i0 = node.inputs[0]
            tgt, acc = _accumulate_target(i0, None)
            s = i0.src()
            check(lib.mseg_maxpool2x2_bwd(C.byref(s), i0.N, i0.H, i0.W, gy.data_ptr(), tgt.data_ptr(), acc, _stream()),
                  "maxpool_bwd")
            continue
        conv = sp.conv
        nm = sp.norm_mod
        has_affine = sp.norm in (NORM["bn"], NORM["gn"])
        dgamma = _grad_buf(nm.weight, direct) if has_affine else None
        dbeta = _grad_buf(nm.bias, direct) if has_affine else None
        dbias = _grad_buf(conv.bias, direct)
        dz = norm_bwd(node, gy, nm.weight.detach() if has_affine else None, dgamma, dbeta, dbias, ws)
        if has_affine:
            grads[id(nm.weight)] = dgamma
            grads[id(nm.bias)] = dbeta
        grads[id(conv.bias)] = dbias
        wt = conv.weight.detach()
        dW = _grad_buf(conv.weight, direct)
        ins = node.inputs
        i0 = ins[0]
        N = node.N
        if sp.kind == "up":
            cin, cout = wt.shape[0], wt.shape[1]
            # dW[ci][co][a][b] = sum x[p][ci] * dz[2p+(a,b)][co]
            on_side(dz, lambda w_, i0=i0, dz=dz, dW=dW, cout=cout, node=node: wgrad(
                i0.src(), [plain_src(dz, cout)], dW, N, i0.H, i0.W, node.H, node.W, 2, 2, 2, 0, w_))
            # dx[p][ci] = sum_{ab,co} dz[2p+(a,b)][co] * W[ci][co][a][b]
            wp = pack_weight(conv.weight, 4, cin, cout, 1, cout * 4, 4, kind="dgrad")
            tgt, acc = _accumulate_target(i0, None)
            igemm([plain_src(dz, cout)], wp, None, N, node.H, node.W, i0.H, i0.W, 2, 2, 2, 0, MODE_CONV, cin,
                  tgt, cin, acc0=acc)
        else:
            cout, cin = wt.shape[0], wt.shape[1]
            stride = 2 if sp.kind == "pool" else 1
            is_first = i0 is tape.x4
            if is_first and cin == 1 and _first_layer_ok(cin, cout, stride):
                w1 = ws.get("first_wgrad", lib.mseg_first_wgrad_workspace_bytes(N, node.H, node.W, cout))
                check(lib.mseg_first_wgrad(i0.z.data_ptr(), dz.data_ptr(), _st(dz), N, node.H, node.W, cout, dW.data_ptr(),
                                           w1.data_ptr(), _stream()), "first_wgrad")
            else:
                on_side(dz, lambda w_, dz=dz, ins=ins, dW=dW, cout=cout, node=node, i0=i0, stride=stride, cin=cin,
                        is_first=is_first: wgrad(plain_src(dz, cout), [n.src() for n in ins], dW, N, node.H, node.W, i0.H,
                                                 i0.W, 3, 3, stride, 1, w_, nch_store=cin if is_first else None))
            if not is_first:
                wp = pack_weight(conv.weight, 9, cin, cout, 1, 9, cin * 9, kind="dgrad")
                morder = MORDER_PARITY if stride == 2 else MORDER_LINEAR
                if len(ins) == 1:
                    tgt, acc = _accumulate_target(i0, None)
                    igemm([plain_src(dz, cout)], wp, None, N, node.H, node.W, i0.H, i0.W, 3, 3, stride, 1,
                          MODE_TCONV, cin, tgt, cin, acc0=acc, morder=morder)
                else:
                    i1 = ins[1]
                    t0, a0 = _accumulate_target(i0, None)
                    t1, a1 = _accumulate_target(i1, None)
                    igemm([plain_src(dz, cout)], wp, None, N, node.H, node.W, i0.H, i0.W, 3, 3, stride, 1,
                          MODE_TCONV, cin, t0, i0.C, acc0=a0, dst1=t1, ld1=i1.C, acc1=a1,
                          split=i0.C, morder=morder)
        grads[id(conv.weight)] = dW
        if on_grads is not None:        # data-parallel: gradients of this layer are final -> start their all-reduce
            on_grads([dW, dbias, dgamma, dbeta])
        del dz, gy
    if side is not None:
        main.wait_stream(side)                           # the optimizer (and whoever reads .grad) sees finished weight gradients
        held.clear()                                     # freed on the main stream, behind the join
    return grads


class _SideWs:
    """the side stream's own scratch names inside the module's Workspace: buffers allocated and used on that stream only"""

    def __init__(self, ws, tag="@side"):
        self.ws, self.tag = ws, tag

    def get(self, name, nbytes, zero=False):
        return self.ws.get(name + self.tag, nbytes, zero)


# ---- autograd glue -------------------------------------------------------------------------------------------
class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        training = module.training
        outs, tape = forward(module._spec, x, training, True, module._workspace(x.device))
        ctx.module = module
        ctx.tape = tape
        ctx.nparams = len(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        module = ctx.module
        tape = ctx.tape
        if tape is None:
            raise RuntimeError("backward through the HIP U-Net twice is not supported")
        ctx.tape = None
        factory = getattr(module, "_grad_sync_factory", None)
        sync = factory() if factory is not None else None
        direct = set()
        grads = backward(module._spec, tape, list(grad_outs), module._workspace(tape.nodes[0].z.device),
                         on_grads=sync.add if sync is not None else None, direct=direct)
        if sync is not None:
            sync.finish()               # averaged in place (RCCL all-reduce overlapped with the kernels above)
        plist = module._spec.params()
        return (None, None) + tuple(None if id(p) in direct else grads.get(id(p)) for p in plist)


def run_module(module, x):
    """Entry used by UNet/DUNet.forward: returns a tuple of NCHW outputs."""
    if isinstance(x, RawFrame):                      # inference on a raw frame (never differentiated)
        outs, _ = forward(module._spec, x, module.training, False, module._workspace(x.device))
        return tuple(outs)
    if not x.is_cuda:
        raise RuntimeError(
            "microbeseg_amd: the U-Net runs on the MI355X HIP path only (got a CPU tensor); there is deliberately "
            "no CPU fallback in the product path — use the oracle under oracle/ for CPU reference runs")
    _lib.load()
    x = x.contiguous().float()
    plist = module._spec.params()
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in plist)
    if need_grad:
        return _NetFunction.apply(module, x, *plist)
    outs, _ = forward(module._spec, x, module.training, False, module._workspace(x.device))
    return tuple(outs)
