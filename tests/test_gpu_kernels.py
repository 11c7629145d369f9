"""GPU (MI355X): each HIP kernel of libmseg_hip through the C ABI vs a plain PyTorch fp32 CPU reference of the same op.
Tolerance: 1e-4 relative (BASELINE.json north_star: "distance maps within 1e-4 relative fp32")."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from microbeseg_amd import engine, _lib
    _lib.load()
    return engine


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def rnd(g, *shape):
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def _act_cpu(x, act):
    return {"none": lambda v: v, "relu": F.relu, "leakyrelu": lambda v: F.leaky_relu(v, 0.01),
            "elu": F.elu, "mish": lambda v: v * torch.tanh(F.softplus(v))}[act](x)


def _mk_node(eng, z_nchw, act="none", scale=None, shift=None, per_sample=False):
    from microbeseg_amd._lib import ACT
    N, Cc, H, W = z_nchw.shape
    node = eng.Node(nhwc(z_nchw).cuda(), N, H, W, Cc)
    node.act = ACT[act]
    if scale is not None:
        node.scale = scale.reshape(-1).cuda()
        node.shift = shift.reshape(-1).cuda()
        node.ss = Cc if per_sample else 0
    return node


def _transform_cpu(z, act, scale, shift, per_sample):
    a = _act_cpu(z, act)
    if scale is None:
        return a
    if per_sample:
        return a * scale[:, :, None, None] + shift[:, :, None, None]
    return a * scale[None, :, None, None] + shift[None, :, None, None]


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride", [
    (2, 8, 16, 16, 16, 1), (1, 64, 64, 32, 48, 1), (3, 16, 8, 20, 12, 1), (2, 32, 160, 16, 16, 1),
    (2, 8, 8, 16, 32, 2), (1, 64, 64, 32, 32, 2), (2, 136, 72, 10, 14, 1),
])
@pytest.mark.parametrize("act", ["none", "mish"])
def test_conv3x3_forward(eng, N, Cin, Cout, H, W, stride, act):
    g = torch.Generator().manual_seed(N * 1000 + Cin + Cout + H)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, N, Cin) * 0.5 + 1.0, rnd(g, N, Cin) * 0.2
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_transform_cpu(z, act, scale, shift, True), w, b, stride=stride, padding=1)
    node = _mk_node(eng, z, act, scale, shift, per_sample=True)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
    eng.igemm([node.src()], wp, b.cuda(), N, H, W, Ho, Wo, 3, 3, stride, 1, eng.MODE_CONV, Cout, out, Cout)
    assert rel_err(nchw(out.cpu()), ref) < TOL


def test_conv3x3_concat_two_sources(eng):
    g = torch.Generator().manual_seed(7)
    N, C0, C1, Cout, H, W = 2, 8, 24, 40, 12, 20
    z0, z1 = rnd(g, N, C0, H, W), rnd(g, N, C1, H, W)
    s0, h0 = rnd(g, C0) * 0.3 + 1, rnd(g, C0) * 0.1
    s1, h1 = rnd(g, N, C1) * 0.3 + 1, rnd(g, N, C1) * 0.1
    w, b = rnd(g, Cout, C0 + C1, 3, 3) * 0.1, rnd(g, Cout)
    x = torch.cat([_transform_cpu(z0, "none", s0, h0, False), _transform_cpu(z1, "relu", s1, h1, True)], 1)
    ref = F.conv2d(x, w, b, padding=1)
    n0 = _mk_node(eng, z0, "none", s0, h0, False)
    n1 = _mk_node(eng, z1, "relu", s1, h1, True)
    wp = eng.pack_weight(w.cuda(), 9, Cout, C0 + C1, 1, (C0 + C1) * 9, 9)
    out = torch.empty((N, H, W, Cout), device="cuda")
    eng.igemm([n0.src(), n1.src()], wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout)
    assert rel_err(nchw(out.cpu()), ref) < TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride", [(2, 8, 16, 16, 16, 1), (1, 32, 72, 12, 20, 1), (2, 16, 16, 16, 24, 2),
                                                   (1, 64, 64, 32, 32, 2)])
def test_conv3x3_dgrad(eng, N, Cin, Cout, H, W, stride):
    """data gradient = TCONV mode; stride 2 uses the parity M-order (dead-tap skipping)."""
    g = torch.Generator().manual_seed(11 + Cin)
    x = rnd(g, N, Cin, H, W).requires_grad_(True)
    w = rnd(g, Cout, Cin, 3, 3) * 0.1
    y = F.conv2d(x, w, None, stride=stride, padding=1)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    ref = x.grad
    wp = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    Ho, Wo = y.shape[2], y.shape[3]
    dz = nhwc(gy).cuda()
    out = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    morder = eng.MORDER_PARITY if stride == 2 else eng.MORDER_LINEAR
    eng.igemm([eng.plain_src(dz, Cout)], wp, None, N, Ho, Wo, H, W, 3, 3, stride, 1, eng.MODE_TCONV, Cin, out, Cin,
              morder=morder)
    assert rel_err(nchw(out.cpu()), ref) < TOL
    # accumulate flag + split destinations
    base = rnd(g, N, H, W, Cin)
    c0 = Cin // 2 // 4 * 4 or 4
    d0, d1 = base[..., :c0].contiguous().cuda(), base[..., c0:].contiguous().cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wp, None, N, Ho, Wo, H, W, 3, 3, stride, 1, eng.MODE_TCONV, Cin, d0, c0,
              acc0=1, dst1=d1, ld1=Cin - c0, acc1=0, split=c0, morder=morder)
    refn = nhwc(ref)
    assert rel_err(d0.cpu(), refn[..., :c0] + base[..., :c0]) < TOL
    assert rel_err(d1.cpu(), refn[..., c0:]) < TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 8, 8, 8), (1, 128, 64, 8, 12), (2, 32, 40, 6, 10)])
def test_conv_transpose_forward_and_dgrad(eng, N, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(5 + Cin)
    x = rnd(g, N, Cin, H, W).requires_grad_(True)
    w, b = rnd(g, Cin, Cout, 2, 2) * 0.1, rnd(g, Cout)
    y = F.conv_transpose2d(x, w, b, stride=2)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    node = _mk_node(eng, x.detach())
    wp = eng.pack_weight(w.cuda(), 4, Cout, Cin, 1, 4, Cout * 4, merge_taps=True)
    out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    eng.igemm([node.src()], wp, b.cuda(), N, H, W, H, W, 1, 1, 1, 0, eng.MODE_CONV, 4 * Cout, out, Cout,
              epi=eng.EPI_SCATTER2X2, Cq=Cout)
    assert rel_err(nchw(out.cpu()), y.detach()) < TOL
    wd = eng.pack_weight(w.cuda(), 4, Cin, Cout, 1, Cout * 4, 4)
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    dz = nhwc(gy).cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, 2 * H, 2 * W, H, W, 2, 2, 2, 0, eng.MODE_CONV, Cin, dx, Cin)
    assert rel_err(nchw(dx.cpu()), x.grad) < TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride", [(2, 8, 16, 16, 16, 1), (3, 72, 40, 10, 14, 1), (2, 16, 16, 16, 24, 2),
                                                   (4, 64, 64, 32, 32, 1), (1, 136, 200, 9, 7, 1)])
def test_conv3x3_wgrad(eng, N, Cin, Cout, H, W, stride):
    g = torch.Generator().manual_seed(3 + Cout)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1, rnd(g, Cin) * 0.1
    xin = _transform_cpu(z, "relu", scale, shift, False)
    w = (rnd(g, Cout, Cin, 3, 3) * 0.1).requires_grad_(True)
    y = F.conv2d(xin, w, None, stride=stride, padding=1)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    node = _mk_node(eng, z, "relu", scale, shift, False)
    dz = nhwc(gy).cuda()
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    ws = eng.Workspace(torch.device("cuda"))
    eng.wgrad(eng.plain_src(dz, Cout), [node.src()], dW, N, y.shape[2], y.shape[3], H, W, 3, 3, stride, 1, ws)
    assert rel_err(dW.cpu(), w.grad) < TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W,act,per_sample,two_src", [
    (2, 64, 64, 16, 32, "relu", False, False),        # 4 x 8 pixel blocks (row length a multiple of 8)
    (3, 72, 40, 30, 24, "relu", False, False),        # channel tails, block rows hanging over the image bottom
    (2, 64, 128, 16, 20, "relu", False, False),       # row length a multiple of 4 only: 8 x 4 blocks
    (2, 128, 64, 24, 16, "mish", False, False),       # any-activation transform
    (3, 64, 64, 16, 16, "relu", True, False),         # per-sample (Group / InstanceNorm) tables
    (2, 128, 64, 16, 16, "relu", False, True),        # virtual concat: two Q sources
    (5, 64, 64, 64, 64, "relu", False, False),        # many steps: several pixel-range splits
])
def test_conv3x3_wgrad_all_taps_kernel(eng, N, Cin, Cout, H, W, act, per_sample, two_src):
    """wgrad_halo9_kernel: one workgroup holds the accumulators of all nine taps of a 64 x 64 channel tile and reads them
    from ONE staged halo image per 32-pixel block; every variant against torch's weight gradient (1e-4)."""
    g = torch.Generator().manual_seed(17 + Cin + W)
    z = rnd(g, N, Cin, H, W)
    shape = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *shape) * 0.3 + 1, rnd(g, *shape) * 0.1
    xin = _transform_cpu(z, act, scale, shift, per_sample)
    w = (rnd(g, Cout, Cin, 3, 3) * 0.1).requires_grad_(True)
    y = F.conv2d(xin, w, None, padding=1)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    if two_src:
        c0 = Cin // 2
        nodes = [_mk_node(eng, z[:, :c0], act, scale[..., :c0], shift[..., :c0], per_sample),
                 _mk_node(eng, z[:, c0:], act, scale[..., c0:], shift[..., c0:], per_sample)]
    else:
        nodes = [_mk_node(eng, z, act, scale, shift, per_sample)]
    qs = [n.src() for n in nodes]          # MsegSrc holds raw pointers: the nodes must outlive the launch
    dz = nhwc(gy).cuda()
    P = eng.plain_src(dz, Cout)
    assert eng.wgrad_query(P, qs, N, H, W, H, W, 3, 3, 1, 1).name.startswith("wgrad_halo9_kernel<%d" % (3 if W % 8 == 0 else 2))
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    eng.wgrad(P, qs, dW, N, H, W, H, W, 3, 3, 1, 1, eng.Workspace(torch.device("cuda")))
    assert rel_err(dW.cpu(), w.grad) < TOL


def test_convT_wgrad(eng):
    g = torch.Generator().manual_seed(9)
    N, Cin, Cout, H, W = 2, 32, 24, 6, 10
    x = rnd(g, N, Cin, H, W)
    w = (rnd(g, Cin, Cout, 2, 2) * 0.1).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, stride=2)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    node = _mk_node(eng, x)
    dz = nhwc(gy).cuda()
    dW = torch.full((Cin, Cout, 2, 2), float("nan"), device="cuda")
    ws = eng.Workspace(torch.device("cuda"))
    eng.wgrad(node.src(), [eng.plain_src(dz, Cout)], dW, N, H, W, 2 * H, 2 * W, 2, 2, 2, 0, ws)
    assert rel_err(dW.cpu(), w.grad) < TOL


@pytest.mark.parametrize("norm", ["bn", "gn", "in"])
@pytest.mark.parametrize("act", ["relu", "mish", "elu", "leakyrelu", "none"])
@pytest.mark.parametrize("N,Cc,H,W", [(3, 16, 12, 20), (2, 64, 32, 32)])
def test_norm_forward_backward(eng, norm, act, N, Cc, H, W):
    """norm(act(z)) statistics, scale/shift tables, running stats, and the full backward (dz, dgamma, dbeta, dbias)."""
    from microbeseg_amd._lib import ACT, NORM
    g = torch.Generator().manual_seed(17 + Cc)
    z = (rnd(g, N, Cc, H, W) * 1.5 + 0.3).requires_grad_(True)
    gamma = (rnd(g, Cc) * 0.3 + 1).requires_grad_(True)
    beta = (rnd(g, Cc) * 0.1).requires_grad_(True)
    rm, rv = rnd(g, Cc) * 0.1, torch.rand(Cc, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    a = _act_cpu(z, act)
    if norm == "bn":
        y = F.batch_norm(a, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    elif norm == "gn":
        y = F.group_norm(a, 8, gamma, beta, 1e-5)
    else:
        y = F.instance_norm(a, eps=1e-5)
    gy = rnd(g, *y.shape)
    y.backward(gy)

    node = eng.Node(nhwc(z.detach()).cuda(), N, H, W, Cc)
    node.act = ACT[act]
    ws = eng.Workspace(torch.device("cuda"))
    has_aff = norm != "in"
    gm, bt = (gamma.detach().cuda(), beta.detach().cuda()) if has_aff else (None, None)
    rmd, rvd = (rm.cuda(), rv.cuda()) if norm == "bn" else (None, None)
    eng.norm_stats(node, NORM[norm], gm, bt, rmd, rvd, True, ws)
    # the tables reproduce y when applied to a
    sc = node.scale.cpu().reshape(-1, Cc)
    sh = node.shift.cpu().reshape(-1, Cc)
    y_hip = a.detach() * sc[:, :, None, None] + sh[:, :, None, None] if norm != "bn" else \
        a.detach() * sc[0][None, :, None, None] + sh[0][None, :, None, None]
    assert rel_err(y_hip, y.detach()) < TOL
    if norm == "bn":
        assert rel_err(rmd.cpu(), rm_ref) < 1e-5 and rel_err(rvd.cpu(), rv_ref) < 1e-5
    gyd = nhwc(gy).cuda()
    dgamma = torch.empty(Cc, device="cuda") if has_aff else None
    dbeta = torch.empty(Cc, device="cuda") if has_aff else None
    dbias = torch.empty(Cc, device="cuda")
    dz = eng.norm_bwd(node, gyd, gm, dgamma, dbeta, dbias, ws)
    scale_ref = z.grad.abs().max().item()
    assert rel_err(nchw(dz.cpu()), z.grad) < TOL
    assert rel_err(dbias.cpu(), z.grad.sum((0, 2, 3)), floor=1e-3 * scale_ref * H * W) < TOL
    if has_aff:
        assert rel_err(dgamma.cpu(), gamma.grad) < TOL
        assert rel_err(dbeta.cpu(), beta.grad) < TOL


@pytest.mark.parametrize("st", ["f32", "bf16"])
@pytest.mark.parametrize("norm", ["bn", "gn", "in"])
@pytest.mark.parametrize("N,Cc,H,W", [(4, 64, 96, 96), (9, 256, 40, 40), (32, 1024, 12, 10), (2, 72, 33, 31), (33, 128, 64, 64)])
def test_norm_tails_equal_separate_launches(eng, st, norm, N, Cc, H, W):
    """Four ways to get from a pass's partial sums to the tables: separate reduction + finalize launches; the reduction
    launch whose last workgroup also finalizes (mseg_norm_set_finish(1), <= 256 channels); one launch in which a workgroup
    does all of it for four channels (mseg_norm_set_finish(2), the default); the last workgroups of the pass itself
    (mseg_norm_set_tails(1)).  Every output — tables, saved statistics, running statistics, dz, dgamma,
    dbeta, dbias — is BIT-identical between them, call after call on one workspace (the arrival counters return to zero),
    for one to many chunks, channel slices and samples."""
    from microbeseg_amd import _lib
    from microbeseg_amd._lib import ACT, NORM
    lib = _lib.load()
    if st == "bf16" and Cc % 8:
        pytest.skip("bf16 storage: 8 channels per thread")
    dt = torch.bfloat16 if st == "bf16" else torch.float32
    g = torch.Generator().manual_seed(91 + Cc + N)
    z = (rnd(g, N, H, W, Cc) * 1.5 + 0.3).cuda().to(dt)
    gy = rnd(g, N, H, W, Cc).cuda().to(dt)
    gamma, beta = (rnd(g, Cc) * 0.3 + 1).cuda(), (rnd(g, Cc) * 0.1).cuda()
    has_aff = norm != "in"
    ws = eng.Workspace(torch.device("cuda"))
    results = []
    for tails, finish in ((0, 0), (0, 1), (1, 0), (0, 1), (0, 2), (0, 2)):
        assert lib.mseg_norm_set_tails(tails) == 0 and lib.mseg_norm_set_finish(finish) == 0
        try:
            node = eng.Node(z, N, H, W, Cc)
            node.act = ACT["relu"]
            rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
            eng.norm_stats(node, NORM[norm], gamma if has_aff else None, beta if has_aff else None,
                           rm if norm == "bn" else None, rv if norm == "bn" else None, True, ws)
            dgamma = torch.empty(Cc, device="cuda") if has_aff else None
            dbeta = torch.empty(Cc, device="cuda") if has_aff else None
            dbias = torch.empty(Cc, device="cuda")
            dz = eng.norm_bwd(node, gy.clone(), gamma if has_aff else None, dgamma, dbeta, dbias, ws)   # (dz may alias gy)
            torch.cuda.synchronize()
            out = [node.scale, node.shift, node.mean, node.rstd, rm, rv, dz.float(), dbias]
            if has_aff:
                out += [dgamma, dbeta]
            results.append([o.clone() for o in out])
        finally:
            lib.mseg_norm_set_tails(0)
            lib.mseg_norm_set_finish(-1)
    for k, (a, b, c, d, e, f) in enumerate(zip(*results)):
        assert torch.isfinite(a).all(), k
        assert torch.equal(a, b), f"output {k}: the finishing reduction differs from the separate launches"
        assert torch.equal(a, c), f"output {k}: tails differ from the separate launches"
        assert torch.equal(a, d), f"output {k}: second call on the same workspace differs"
        assert torch.equal(a, e), f"output {k}: the four-channels-per-workgroup launch differs from the separate launches"
        assert torch.equal(a, f), f"output {k}: its second call differs"
    # the counters are back at zero
    head = ws.buf["norm"][:65536].view(torch.int32)
    assert int(head.abs().max().item()) == 0


@pytest.mark.parametrize("dtype,H0,W0", [(np.uint16, 100, 90), (np.uint8, 64, 64), (np.uint16, 333, 257), (np.uint16, 256, 320)])
def test_frame_normalisation_on_the_device(eng, dtype, H0, W0):
    """K14 on the device: min / max of a raw frame, top / left padding to the tested shapes and
    2 * (f32(x) - min) / (max - min) - 1 (reference infer.py:346-348, utils.py:124-163) — the normalised frame equals the
    host formula BIT FOR BIT, and the first convolution fed with the raw frame equals, bit for bit, the same convolution fed
    with the host-normalised tensor."""
    from microbeseg_amd import _lib
    from microbeseg_amd.utils.utils import zero_pad_model_input
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(5 + H0))
    hi = 255 if dtype == np.uint8 else 60000
    frame = rng.integers(7, hi, size=(H0, W0)).astype(dtype)
    fmin, fmax = np.min(frame), np.max(frame)
    padded, pads = zero_pad_model_input(np.copy(frame), pad_val=fmin)
    want = 2 * (padded.astype(np.float32) - fmin) / (fmax - fmin) - 1          # the reference's expression, numpy scalars
    store = torch.from_numpy(frame.view(np.int16) if dtype == np.uint16 else frame).cuda()
    rf = eng.RawFrame(store, pads[0], pads[1])
    mm = rf.minmax.cpu().numpy().view(np.uint32)
    assert (~mm[0]) & 0xffffffff == int(fmin) and mm[1] == int(fmax)
    got = rf.normalized().cpu().numpy()[0, 0]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))
    # fused into the first convolution
    g = torch.Generator().manual_seed(3)
    Cout = 64
    w, b = (rnd(g, Cout, 1, 3, 3) * 0.3).cuda(), rnd(g, Cout).cuda()
    H, W = want.shape
    for st in (torch.float32, torch.bfloat16):
        z_raw = torch.full((1, H, W, Cout), float("nan"), device="cuda", dtype=st)
        _lib.check(lib.mseg_first_conv_fwd_raw(*rf.args(), w.data_ptr(), b.data_ptr(), Cout, z_raw.data_ptr(), eng._st(z_raw),
                                               torch.cuda.current_stream().cuda_stream), "first_conv_fwd_raw")
        x4 = torch.zeros((1, H, W, 4), device="cuda")
        x4[..., 0] = torch.from_numpy(want.astype(np.float32)).cuda()
        z_ref = torch.full((1, H, W, Cout), float("nan"), device="cuda", dtype=st)
        _lib.check(lib.mseg_first_conv_fwd(x4.data_ptr(), w.data_ptr(), b.data_ptr(), 1, H, W, 1, Cout, z_ref.data_ptr(),
                                           eng._st(z_ref), torch.cuda.current_stream().cuda_stream), "first_conv_fwd")
        assert torch.equal(z_raw.float(), z_ref.float())


def test_bn_eval_coeffs(eng):
    from microbeseg_amd._lib import NORM
    g = torch.Generator().manual_seed(1)
    N, Cc, H, W = 2, 32, 8, 8
    z = rnd(g, N, Cc, H, W)
    gamma, beta, rm, rv = rnd(g, Cc), rnd(g, Cc), rnd(g, Cc), torch.rand(Cc, generator=g) + 0.5
    ref = F.batch_norm(F.relu(z), rm, rv, gamma, beta, False, 0.1, 1e-5)
    node = eng.Node(nhwc(z).cuda(), N, H, W, Cc)
    node.act = 1
    eng.norm_stats(node, NORM["bn"], gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), False, eng.Workspace("cuda"))
    y = F.relu(z) * node.scale.cpu()[None, :, None, None] + node.shift.cpu()[None, :, None, None]
    assert rel_err(y, ref) < 1e-5


@pytest.mark.parametrize("Cc,Co", [(8, 1), (64, 3), (16, 2)])
def test_head_forward_backward(eng, Cc, Co):
    from microbeseg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(23 + Cc)
    N, H, W = 2, 20, 12
    z = rnd(g, N, Cc, H, W)
    scale, shift = rnd(g, Cc) * 0.3 + 1, rnd(g, Cc) * 0.1
    xin = _transform_cpu(z, "mish", scale, shift, False).requires_grad_(True)
    w = (rnd(g, Co, Cc, 1, 1) * 0.2).requires_grad_(True)
    b = rnd(g, Co).requires_grad_(True)
    y = F.conv2d(xin, w, b)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    node = _mk_node(eng, z, "mish", scale, shift, False)
    out = torch.empty((N, Co, H, W), device="cuda")
    s = node.src()
    wd, bd = w.detach().cuda(), b.detach().cuda()
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mseg_head_fwd(C.byref(s), N, H * W, wd.data_ptr(), bd.data_ptr(), Co, out.data_ptr(), stream))
    assert rel_err(out.cpu(), y.detach()) < TOL
    gx = torch.empty((N, H, W, Cc), device="cuda")
    dW, db = torch.empty((Co, Cc), device="cuda"), torch.empty(Co, device="cuda")
    wsb = torch.empty(lib.mseg_head_bwd_workspace_bytes(N, H * W, Cc, Co), dtype=torch.uint8, device="cuda")
    god = gy.cuda()
    _lib.check(lib.mseg_head_bwd(C.byref(s), N, H * W, wd.data_ptr(), Co, god.data_ptr(), gx.data_ptr(), 0, dW.data_ptr(),
                                 db.data_ptr(), wsb.data_ptr(), stream))
    assert rel_err(nchw(gx.cpu()), xin.grad) < TOL
    assert rel_err(dW.cpu(), w.grad.reshape(Co, Cc)) < TOL
    assert rel_err(db.cpu(), b.grad) < TOL


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 1, 64, 24, 40), (1, 3, 16, 17, 9), (3, 1, 8, 32, 32), (2, 1, 64, 19, 50),
                                           (1, 1, 32, 9, 70), (5, 1, 64, 64, 64)])
def test_first_layer_kernels(eng, N, Cin, Cout, H, W):
    """Conv2d(ch_in, Cout, 3, padding=1) on the raw (zero-padded to 4 channels) input + its weight gradient (ch_in 1)."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(77 + Cout)
    x = rnd(g, N, Cin, H, W)
    w = (rnd(g, Cout, Cin, 3, 3) * 0.3).requires_grad_(True)
    b = rnd(g, Cout)
    ref = F.conv2d(x, w, b, padding=1)
    x4 = torch.zeros((N, H, W, 4))
    x4[..., :Cin] = nhwc(x)
    x4 = x4.cuda()
    z = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    wd, bd = w.detach().cuda(), b.cuda()          # keep the device copies alive across the asynchronous launch
    _lib.check(lib.mseg_first_conv_fwd(x4.data_ptr(), wd.data_ptr(), bd.data_ptr(), N, H, W, Cin, Cout, z.data_ptr(), 0,
                                       st), "first_conv_fwd")
    if Cout % 8 == 0:                              # bf16 storage of z: the same values rounded to bf16
        z16 = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
        _lib.check(lib.mseg_first_conv_fwd(x4.data_ptr(), wd.data_ptr(), bd.data_ptr(), N, H, W, Cin, Cout, z16.data_ptr(),
                                           1, st), "first_conv_fwd")
        assert torch.equal(z16, z.to(torch.bfloat16))
    assert rel_err(nchw(z.cpu()), ref.detach()) < TOL
    if Cin == 1:
        gy = rnd(g, N, Cout, H, W)
        ref.backward(gy)
        ws = torch.empty(lib.mseg_first_wgrad_workspace_bytes(N, H, W, Cout), dtype=torch.uint8, device="cuda")
        dW = torch.full((Cout, 1, 3, 3), float("nan"), device="cuda")
        gyd = nhwc(gy).cuda()
        _lib.check(lib.mseg_first_wgrad(x4.data_ptr(), gyd.data_ptr(), 0, N, H, W, Cout, dW.data_ptr(), ws.data_ptr(), st),
                   "first_wgrad")
        assert rel_err(dW.cpu(), w.grad) < TOL
        g16 = gyd.to(torch.bfloat16)               # bf16 storage of dz: exact on the rounded operand
        ref2 = F.conv2d(x, w.detach().clone().requires_grad_(True), b, padding=1)
        w2 = torch.zeros_like(w).requires_grad_(True)
        F.conv2d(x, w2, None, padding=1).backward(nchw(g16.float().cpu()))
        _lib.check(lib.mseg_first_wgrad(x4.data_ptr(), g16.data_ptr(), 1, N, H, W, Cout, dW.data_ptr(), ws.data_ptr(), st),
                   "first_wgrad")
        assert rel_err(dW.cpu(), w2.grad) < TOL


def test_halo_kernels_beyond_2gib(eng):
    """Operands larger than 2 GiB (33 images of 512 x 512 x 64 fp32 = 2.2 GB): the halo kernels address one image at a
    time, so the batch may be arbitrarily large.  Checked against the same kernels run on the two halves of the batch
    (bit-identical forward; weight gradient = sum of the halves' gradients)."""
    N, Cc, Co, H, W = 33, 64, 64, 512, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((N, H, W, Cc), generator=g, device="cuda")
    assert x.numel() * 4 > 2 ** 31
    w = (torch.randn((Co, Cc, 3, 3), generator=g, device="cuda") * 0.05)
    b = torch.randn(Co, generator=g, device="cuda")
    wp = eng.pack_weight(w, 9, Co, Cc, 1, Cc * 9, 9)

    def fwd(xs):
        n = xs.shape[0]
        out = torch.empty((n, H, W, Co), device="cuda")
        eng.igemm([eng.plain_src(xs, Cc)], wp, b, n, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Co, out, Co)
        return out
    full = fwd(x)
    h0 = 16
    assert torch.equal(full[:h0], fwd(x[:h0].contiguous())) and torch.equal(full[h0:], fwd(x[h0:].contiguous()))
    # spot check of one far image against torch on the CPU
    ref = F.conv2d(nchw(x[N - 1:].cpu()), w.cpu(), b.cpu(), padding=1)
    assert rel_err(nchw(full[N - 1:].cpu()), ref) < TOL
    # weight gradient with dz := the forward output (any tensor of that shape does)
    ws = eng.Workspace(torch.device("cuda"))

    def wg(dz, xs):
        n = xs.shape[0]
        dW = torch.empty_like(w)
        eng.wgrad(eng.plain_src(dz, Co), [eng.plain_src(xs, Cc)], dW, n, H, W, H, W, 3, 3, 1, 1, ws)
        return dW
    dW = wg(full, x)
    parts = wg(full[:h0].contiguous(), x[:h0].contiguous()) + wg(full[h0:].contiguous(), x[h0:].contiguous())
    assert rel_err(dW.cpu(), parts.cpu()) < TOL
    del full, x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N,Cin,Cout,H,W,two_src", [(1, 256, 256, 16, 16, False), (2, 512, 128, 8, 16, False),
                                                    (1, 128, 64, 16, 32, True), (3, 1024, 1024, 16, 8, False)])
def test_split_k_halo_launches(eng, N, Cin, Cout, H, W, two_src):
    """Few tiles + many input channels: the 3x3 stride-1 launch is split over the channel chunks and reduced in a second
    kernel (bias, accumulate flag and split destinations applied there).  Forward with norm-on-load operands and data
    gradient with accumulate + two destinations, against torch on the CPU."""
    from microbeseg_amd import _lib
    g = torch.Generator().manual_seed(100 + Cin + H)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1.0, rnd(g, Cin) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_transform_cpu(z, "relu", scale, shift, False), w, b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    if two_src:
        c0 = Cin // 2
        n0 = _mk_node(eng, z[:, :c0], "relu", scale[:c0], shift[:c0])
        n1 = _mk_node(eng, z[:, c0:], "relu", scale[c0:], shift[c0:])
        srcs = [n0.src(), n1.src()]
    else:
        node = _mk_node(eng, z, "relu", scale, shift)
        srcs = [node.src()]
    # the launch must actually be a split one
    p = _lib.MsegIgemm()
    p.Ngemm, p.Cin, p.NB, p.Hi, p.Wi, p.Ho, p.Wo, p.KH, p.KW, p.stride, p.pad = Cout, Cin, N, H, W, H, W, 3, 3, 1, 1
    assert _lib.load().mseg_igemm_workspace_bytes(C.byref(p)) > 0
    eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout)
    assert rel_err(nchw(out.cpu()), ref) < TOL
    # data gradient: accumulate into destination 0, plain store into destination 1
    x = rnd(g, N, Cin, H, W).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    gy = rnd(g, *y.shape)
    y.backward(gy)
    refn = nhwc(x.grad)
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    base = rnd(g, N, H, W, Cin)
    c0 = Cin // 2
    d0, d1 = base[..., :c0].contiguous().cuda(), torch.full((N, H, W, Cin - c0), float("nan"), device="cuda")
    dz = nhwc(gy).cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, c0, acc0=1,
              dst1=d1, ld1=Cin - c0, acc1=0, split=c0)
    assert rel_err(d0.cpu(), refn[..., :c0] + base[..., :c0]) < TOL
    assert rel_err(d1.cpu(), refn[..., c0:]) < TOL


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("N,Cin,Cout,H,W,two_src", [(2, 64, 64, 16, 32, False), (1, 96, 160, 32, 40, False),
                                                     (2, 128, 64, 16, 16, True), (1, 256, 256, 16, 16, False),
                                                     (1, 40, 24, 32, 20, False), (3, 64, 128, 30, 64, False),
                                                     # weight tensor > 2 MB: pixel-tile-fastest workgroup order, with
                                                     # (16 tiles) and without (256 tiles) split-K
                                                     (2, 512, 512, 16, 16, False), (8, 512, 512, 32, 32, False)])
def test_bf16_halo_forward_and_dgrad(eng, N, Cin, Cout, H, W, two_src):
    """MSEG_PREC_BF16 launches (BASELINE configs[2]): operands rounded to bf16 (RNE) while staged, fp32 accumulate.
    Reference = torch fp32 convolution of the bf16-rounded operands; what is left is the accumulation order and, for
    norm-on-load sources, a rare different rounding of an operand whose fp32 value differs in the last bit -> 5e-4."""
    g = torch.Generator().manual_seed(300 + Cin + H)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1.0, rnd(g, Cin) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z, "relu", scale, shift, False)), _bf(w), b, padding=1)
    exact = F.conv2d(_transform_cpu(z, "relu", scale, shift, False), w, b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    if two_src:
        c0 = Cin // 2
        n0 = _mk_node(eng, z[:, :c0], "relu", scale[:c0], shift[:c0])
        n1 = _mk_node(eng, z[:, c0:], "relu", scale[c0:], shift[c0:])
        srcs = [n0.src(), n1.src()]
    else:
        node = _mk_node(eng, z, "relu", scale, shift)
        srcs = [node.src()]
    c = eng.igemm_query(srcs, wp.Kpad, wp.Npad, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, ld0=Cout, precision="bf16",
                        dst_dtype=eng._st(out))
    assert c is not None and c.bf16 and ("halo" in c.name or "p8" in c.name or "c64p" in c.name)
    eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
    got = nchw(out.cpu())
    assert rel_err(got, ref) < 5e-4
    assert 1e-4 < rel_err(got, exact) < 2e-2            # it really is the bf16 path, and a sane one
    # data gradient (plain operand): accumulate into destination 0, plain store into destination 1
    gy = rnd(g, N, Cout, H, W)
    refn = nhwc(F.conv_transpose2d(_bf(gy), _bf(w), None, padding=1))
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    base = rnd(g, N, H, W, Cin)
    c0 = (Cin // 2 + 3) // 4 * 4
    d0, d1 = base[..., :c0].contiguous().cuda(), torch.full((N, H, W, Cin - c0), float("nan"), device="cuda")
    dz = nhwc(gy).cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, c0, acc0=1,
              dst1=d1, ld1=Cin - c0, acc1=0, split=c0, precision="bf16")
    assert rel_err(d0.cpu(), refn[..., :c0] + base[..., :c0]) < 1e-4
    assert rel_err(d1.cpu(), refn[..., c0:]) < 1e-4


def test_bf16_rejected_for_generic_shapes(eng):
    """bf16 inputs exist for the halo and the gather kernel; a launch that needs the fully general kernel (here: a
    concat boundary that is no multiple of 32 channels) is refused by the C ABI instead of silently computing in another
    precision (the engine never asks for it: it keeps such layers in fp32)."""
    from microbeseg_amd import _lib
    g = torch.Generator().manual_seed(5)
    x0, x1 = rnd(g, 1, 16, 16, 16).cuda(), rnd(g, 1, 16, 16, 16).cuda()
    w = eng.pack_weight(rnd(g, 32, 32, 3, 3).cuda(), 9, 32, 32, 1, 32 * 9, 9)
    out = torch.empty((1, 16, 16, 32), device="cuda")
    srcs = [eng.plain_src(x0, 16), eng.plain_src(x1, 16)]
    assert eng.igemm_query(srcs, w.Kpad, w.Npad, 1, 16, 16, 16, 16, 3, 3, 1, 1, eng.MODE_CONV, 32, ld0=32,
                           precision="bf16") is None
    p = _lib.MsegIgemm()
    p.src[0], p.src[1] = srcs
    p.nsrc, p.Cin, p.Kpad, p.Npad, p.w = 2, 32, w.Kpad, w.Npad, w.bf16().data_ptr()
    p.dst0 = out.data_ptr()
    p.NB, p.Hi, p.Wi, p.Ho, p.Wo, p.KH, p.KW, p.stride, p.pad = 1, 16, 16, 16, 16, 3, 3, 1, 1
    p.Ngemm, p.split, p.ld0, p.precision = 32, 32, 32, 1
    assert _lib.load().mseg_igemm(C.byref(p), torch.cuda.current_stream().cuda_stream) == -1


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 128, 16, 16), (1, 128, 72, 24, 40), (4, 32, 64, 32, 32)])
def test_bf16_gather_stride2_forward_and_dgrad(eng, N, Cin, Cout, H, W):
    """MSEG_PREC_BF16 on the gather kernel: Conv2d 3x3 stride 2 (ConvPool) forward with a norm-on-load source and its data
    gradient in parity M-order, against torch fp32 on the bf16-rounded operands."""
    g = torch.Generator().manual_seed(700 + Cin)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1.0, rnd(g, Cin) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    xin = _transform_cpu(z, "relu", scale, shift, False)
    ref = F.conv2d(_bf(xin), _bf(w), b, stride=2, padding=1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    node = _mk_node(eng, z, "relu", scale, shift)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    c = eng.igemm_query([node.src()], wp.Kpad, wp.Npad, N, H, W, Ho, Wo, 3, 3, 2, 1, eng.MODE_CONV, Cout, ld0=Cout,
                        precision="bf16")
    assert c is not None and c.bf16 and c.name.startswith("igemm_fast_bf16_kernel")
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
    eng.igemm([node.src()], wp, b.cuda(), N, H, W, Ho, Wo, 3, 3, 2, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
    assert rel_err(nchw(out.cpu()), ref) < 5e-4
    assert rel_err(nchw(out.cpu()), F.conv2d(xin, w, b, stride=2, padding=1)) > 1e-4
    gy = rnd(g, N, Cout, Ho, Wo)
    refd = F.conv_transpose2d(_bf(gy), _bf(w), None, stride=2, padding=1, output_padding=1)
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    dz = nhwc(gy).cuda()
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, Ho, Wo, H, W, 3, 3, 2, 1, eng.MODE_TCONV, Cin, dx, Cin,
              morder=eng.MORDER_PARITY, precision="bf16")
    # parity classes that are no whole 128-row tiles need the general kernel: the engine keeps that launch in fp32
    kind = eng.igemm_query([eng.plain_src(dz, Cout)], wd.Kpad, wd.Npad, N, Ho, Wo, H, W, 3, 3, 2, 1, eng.MODE_TCONV, Cin,
                           ld0=Cin, morder=eng.MORDER_PARITY, precision="bf16", bias=False)
    assert (kind is not None) == ((N * H * W) % 512 == 0)
    if kind is None:
        refd = F.conv_transpose2d(gy, w, None, stride=2, padding=1, output_padding=1)
    assert rel_err(nchw(dx.cpu()), refd) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 128, 64, 8, 8), (1, 256, 128, 8, 12)])
def test_bf16_gather_conv_transpose(eng, N, Cin, Cout, H, W):
    """MSEG_PREC_BF16 ConvTranspose2d 2x2 stride 2 as a 1x1 GEMM with the scatter epilogue, and its data gradient."""
    g = torch.Generator().manual_seed(800 + Cin)
    x = rnd(g, N, Cin, H, W)
    w, b = rnd(g, Cin, Cout, 2, 2) * 0.1, rnd(g, Cout)
    ref = F.conv_transpose2d(_bf(x), _bf(w), b, stride=2)
    node = _mk_node(eng, x)
    wp = eng.pack_weight(w.cuda(), 4, Cout, Cin, 1, 4, Cout * 4, merge_taps=True)
    out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device="cuda")
    eng.igemm([node.src()], wp, b.cuda(), N, H, W, H, W, 1, 1, 1, 0, eng.MODE_CONV, 4 * Cout, out, Cout,
              epi=eng.EPI_SCATTER2X2, Cq=Cout, precision="bf16")
    assert rel_err(nchw(out.cpu()), ref) < 1e-4
    assert rel_err(nchw(out.cpu()), F.conv_transpose2d(x, w, b, stride=2)) > 1e-4
    gy = rnd(g, N, Cout, 2 * H, 2 * W)
    refd = F.conv2d(_bf(gy), _bf(w), None, stride=2)
    wd = eng.pack_weight(w.cuda(), 4, Cin, Cout, 1, Cout * 4, 4)
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    dz = nhwc(gy).cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, 2 * H, 2 * W, H, W, 2, 2, 2, 0, eng.MODE_CONV, Cin, dx, Cin,
              precision="bf16")
    assert rel_err(nchw(dx.cpu()), refd) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,H,W,two_src,act", [(2, 64, 64, 16, 16, False, "relu"), (3, 72, 40, 16, 12, False, "relu"),
                                                        (2, 128, 96, 24, 32, True, "relu"), (1, 64, 128, 40, 40, False, "mish"),
                                                        (2, 32, 64, 30, 20, False, "none"), (4, 64, 64, 32, 32, False, "relu")])
def test_bf16_halo_wgrad(eng, N, Cin, Cout, H, W, two_src, act):
    """MSEG_PREC_BF16 weight gradient of a 3x3 stride-1 conv: dz and the (normalised) conv input rounded to bf16,
    fp32 accumulate over all pixels.  Reference = torch fp32 weight gradient of the rounded operands."""
    g = torch.Generator().manual_seed(500 + Cin + W)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1, rnd(g, Cin) * 0.1
    plain = act == "none"
    xin = z if plain else _transform_cpu(z, act, scale, shift, False)
    gy = rnd(g, N, Cout, H, W)
    ref = torch.nn.grad.conv2d_weight(_bf(xin), (Cout, Cin, 3, 3), _bf(gy), padding=1)
    exact = torch.nn.grad.conv2d_weight(xin, (Cout, Cin, 3, 3), gy, padding=1)
    if two_src:
        c0 = 64
        keep = [_mk_node(eng, z[:, :c0], act, scale[:c0], shift[:c0]), _mk_node(eng, z[:, c0:], act, scale[c0:], shift[c0:])]
        qs = [k.src() for k in keep]
    elif plain:
        keep = nhwc(z).cuda()                            # the descriptors hold raw pointers: keep the tensors alive
        qs = [eng.plain_src(keep, Cin)]
    else:
        keep = _mk_node(eng, z, act, scale, shift, False)
        qs = [keep.src()]
    dz = nhwc(gy).cuda()
    P = eng.plain_src(dz, Cout)
    assert eng.wgrad_query(P, qs, N, H, W, H, W, 3, 3, 1, 1, precision="bf16") is not None
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    ws = eng.Workspace(torch.device("cuda"))
    eng.wgrad(P, qs, dW, N, H, W, H, W, 3, 3, 1, 1, ws, precision="bf16")
    assert rel_err(dW.cpu(), ref) < 5e-4
    assert 1e-5 < rel_err(dW.cpu(), exact) < 2e-2


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 128, 16, 16), (1, 128, 72, 24, 40), (3, 32, 64, 32, 24)])
def test_bf16_strided_wgrads(eng, N, Cin, Cout, H, W):
    """MSEG_PREC_BF16 weight gradients of the strided layers through the same transposed-read kernel: Conv2d 3x3 stride 2
    (ConvPool; norm-on-load input) and ConvTranspose2d 2x2 stride 2 (norm-on-load input as the P operand)."""
    g = torch.Generator().manual_seed(900 + Cin)
    z = rnd(g, N, Cin, H, W)
    scale, shift = rnd(g, Cin) * 0.3 + 1, rnd(g, Cin) * 0.1
    xin = _transform_cpu(z, "relu", scale, shift, False)
    node = _mk_node(eng, z, "relu", scale, shift, False)
    ws = eng.Workspace(torch.device("cuda"))
    # stride-2 conv: P = dz (H/2 x W/2), Q = input
    Ho, Wo = H // 2, W // 2
    gy = rnd(g, N, Cout, Ho, Wo)
    ref = torch.nn.grad.conv2d_weight(_bf(xin), (Cout, Cin, 3, 3), _bf(gy), stride=2, padding=1)
    dz = nhwc(gy).cuda()
    P = eng.plain_src(dz, Cout)
    assert eng.wgrad_query(P, [node.src()], N, Ho, Wo, H, W, 3, 3, 2, 1, precision="bf16") is not None
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    eng.wgrad(P, [node.src()], dW, N, Ho, Wo, H, W, 3, 3, 2, 1, ws, precision="bf16")
    assert rel_err(dW.cpu(), ref) < 5e-4
    assert rel_err(dW.cpu(), torch.nn.grad.conv2d_weight(xin, (Cout, Cin, 3, 3), gy, stride=2, padding=1)) > 1e-5
    # ConvTranspose2d(Cin, Cout, 2, 2): P = its (normalised) input, Q = gradient of its output (2H x 2W)
    gy2 = rnd(g, N, Cout, 2 * H, 2 * W)
    refT = torch.nn.grad.conv2d_weight(_bf(gy2), (Cin, Cout, 2, 2), _bf(xin), stride=2)
    dz2 = nhwc(gy2).cuda()
    Q = eng.plain_src(dz2, Cout)
    assert eng.wgrad_query(node.src(), [Q], N, H, W, 2 * H, 2 * W, 2, 2, 2, 0, precision="bf16") is not None
    dWT = torch.full((Cin, Cout, 2, 2), float("nan"), device="cuda")
    eng.wgrad(node.src(), [Q], dWT, N, H, W, 2 * H, 2 * W, 2, 2, 2, 0, ws, precision="bf16")
    assert rel_err(dWT.cpu(), refT) < 5e-4


def test_repack_all_equals_single_packs(eng):
    """engine.repack_all (ONE launch over every packed operand of a network, LDS-tiled transposition; called by FusedAdam
    after the update) against the per-operand pack kernel: forward and data-gradient operands of 3x3 convolutions, both
    operands of a 2x2 transposed convolution (merged taps), channel counts that are no multiples of 32, fp32 and bf16."""
    g = torch.Generator().manual_seed(12)
    convs = [torch.nn.Parameter(rnd(g, co, ci, 3, 3).cuda()) for co, ci in ((64, 64), (40, 72), (128, 1), (8, 16))]
    ups = [torch.nn.Parameter(rnd(g, ci, co, 2, 2).cuda()) for ci, co in ((128, 64), (24, 40))]
    entries = []
    for p in convs:
        co, ci = p.shape[:2]
        entries += [eng.pack_weight(p, 9, co, ci, 1, ci * 9, 9, kind="fwd"), eng.pack_weight(p, 9, ci, co, 1, 9, ci * 9, kind="dgrad")]
    for p in ups:
        ci, co = p.shape[:2]
        entries += [eng.pack_weight(p, 4, co, ci, 1, 4, co * 4, merge_taps=True, kind="fwd"),
                    eng.pack_weight(p, 4, ci, co, 1, co * 4, 4, kind="dgrad")]
    for e in entries[::2]:
        e.bf16()                                       # half of them also carry a bf16 operand
    with torch.no_grad():
        for p in convs + ups:
            p.mul_(-1.5).add_(0.25)                    # "optimizer step": bumps the version counters
    assert all(e.stale() for e in entries)
    assert eng.repack_all(convs + ups) == len(entries)
    assert not any(e.stale() for e in entries)
    for e in entries:
        T, R, rpad, Cc, kpad, st, sr, sc = e.job
        once = eng.pack_weight(e.param.detach(), T, R, Cc, st, sr, sc, merge_taps=e.merge)
        assert torch.equal(e.t, once.t)
        if e._t16 is not None:
            assert torch.equal(e._t16, once.t.to(torch.bfloat16))


# ---- bf16 tensor STORAGE (MSEG_ST_BF16: BASELINE configs[2], "bf16 forward / backward") -----------------------------------
def _b16(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("norm", ["bn", "gn", "in"])
@pytest.mark.parametrize("act", ["relu", "mish"])
def test_bf16_storage_norm_forward_backward(eng, norm, act):
    """normalisation statistics / backward on bf16-stored z, gy, dz (8 channels per thread): the kernel computes in fp32 on
    the values as stored; against torch on the SAME rounded inputs, output dz compared after rounding to bf16."""
    from microbeseg_amd import _lib
    from microbeseg_amd._lib import ACT, NORM
    lib = _lib.load()
    N, Cc, H, W = 3, 64, 12, 20
    g = torch.Generator().manual_seed(5)
    z16 = _b16(rnd(g, N, Cc, H, W))
    gy16 = _b16(rnd(g, N, Cc, H, W))
    gamma, beta = rnd(g, Cc) * 0.3 + 1, rnd(g, Cc) * 0.1
    z = z16.float().requires_grad_(True)
    a = _act_cpu(z, act)
    if norm == "bn":
        y = F.batch_norm(a, None, None, gamma, beta, True, 0.1, 1e-5)
    elif norm == "gn":
        y = F.group_norm(a, 8, gamma, beta, 1e-5)
    else:
        y = F.instance_norm(a, eps=1e-5)
    y.backward(gy16.float())
    zd, gyd = nhwc(z16).cuda(), nhwc(gy16).cuda()
    ss = 0 if norm == "bn" else Cc
    nsc = Cc if norm == "bn" else N * Cc
    ng = Cc if norm == "bn" else (N * 8 if norm == "gn" else N * Cc)
    scale, shift = torch.empty(nsc, device="cuda"), torch.empty(nsc, device="cuda")
    mean, rstd = torch.empty(ng, device="cuda"), torch.empty(ng, device="cuda")
    aout = torch.empty_like(zd) if act == "mish" else None
    ws = torch.zeros(lib.mseg_norm_workspace_bytes(N, H * W, Cc), dtype=torch.uint8, device="cuda")   # counters: zero once
    st = torch.cuda.current_stream().cuda_stream
    gd, bd = (gamma.cuda(), beta.cuda()) if norm != "in" else (None, None)
    P = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.mseg_norm_stats(zd.data_ptr(), N, H * W, Cc, 1, ACT[act], NORM[norm], P(gd), P(bd), 1e-5,
                                   scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None, 0.1,
                                   P(aout), ws.data_ptr(), st), "norm_stats")
    # y = a * scale + shift reproduces the normalised tensor (a as stored: bf16 for the materialised activation)
    a_used = nchw(aout.float().cpu()) if aout is not None else a.detach()
    sc = scale.cpu().reshape((1, Cc, 1, 1) if norm == "bn" else (N, Cc, 1, 1))
    sh = shift.cpu().reshape(sc.shape)
    tol_y = 1e-4 if aout is None else 2e-2       # a rounded to bf16 shifts the statistics by ~2^-9 relative
    assert rel_err(a_used * sc + sh, y.detach()) < tol_y
    dzd = torch.empty_like(zd)
    dgamma, dbeta, dbias = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    _lib.check(lib.mseg_norm_bwd(gyd.data_ptr(), zd.data_ptr(), N, H * W, Cc, 1, ACT[act], NORM[norm], P(gd),
                                 mean.data_ptr(), rstd.data_ptr(), dzd.data_ptr(), P(dgamma) if norm != "in" else None,
                                 P(dbeta) if norm != "in" else None, dbias.data_ptr(), P(aout), ws.data_ptr(), st), "norm_bwd")
    tol = 1e-2 if aout is not None else 4e-3      # one bf16 rounding of dz (2^-9) on top of the fp32 arithmetic
    assert rel_err(nchw(dzd.float().cpu()), z.grad) < tol
    assert rel_err(dbias.cpu(), nchw(dzd.float().cpu()).sum((0, 2, 3))) < 1e-4      # sum of dz AS STORED
    if norm != "in":
        assert rel_err(dbeta.cpu(), gy16.float().sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,H,W,two_src", [(2, 64, 64, 16, 32, False), (1, 96, 160, 32, 40, False),
                                                     (2, 128, 64, 16, 16, True), (1, 256, 256, 16, 16, False),
                                                     (3, 64, 128, 30, 64, False), (2, 512, 512, 16, 16, False)])
def test_bf16_storage_halo_forward_and_dgrad(eng, N, Cin, Cout, H, W, two_src):
    """bf16 sources AND bf16 destinations through the bf16 halo kernels (8 channels per staging thread): forward with
    norm-on-load sources, data gradient with a plain bf16 operand, accumulate into a bf16 destination."""
    g = torch.Generator().manual_seed(300 + Cin + H)
    z16 = _b16(rnd(g, N, Cin, H, W))
    scale, shift = rnd(g, Cin) * 0.3 + 1.0, rnd(g, Cin) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z16.float(), "relu", scale, shift, False)), _bf(w), b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)

    def node16(zz, sc, sh):
        n = _mk_node(eng, zz.float(), "relu", sc, sh)
        n.z = n.z.to(torch.bfloat16)
        return n
    if two_src:
        c0 = Cin // 2
        nodes = [node16(z16[:, :c0], scale[:c0], shift[:c0]), node16(z16[:, c0:], scale[c0:], shift[c0:])]
    else:
        nodes = [node16(z16, scale, shift)]
    srcs = [n.src() for n in nodes]
    assert all(s.dtype == 1 for s in srcs)
    eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
    got = nchw(out.float().cpu())
    assert rel_err(got, ref) < 6e-3                      # the bf16 rounding of the stored result
    assert rel_err(got, _bf(ref)) < 6e-3
    # data gradient: plain bf16 operand, accumulate into bf16 destination 0, plain store into bf16 destination 1
    gy16 = _b16(rnd(g, N, Cout, H, W))
    refn = nhwc(F.conv_transpose2d(gy16.float(), _bf(w), None, padding=1))
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    base = _b16(rnd(g, N, H, W, Cin))
    c0 = (Cin // 2 + 7) // 8 * 8
    d0 = base[..., :c0].contiguous().cuda()
    d1 = torch.full((N, H, W, Cin - c0), float("nan"), device="cuda", dtype=torch.bfloat16)
    dz = nhwc(gy16).cuda()
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, c0, acc0=1,
              dst1=d1, ld1=Cin - c0, acc1=0, split=c0, precision="bf16")
    assert rel_err(d0.float().cpu(), refn[..., :c0] + base[..., :c0].float()) < 6e-3
    assert rel_err(d1.float().cpu(), refn[..., c0:]) < 6e-3


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 64, 128, 16, 16), (1, 128, 72, 24, 40), (4, 32, 64, 32, 32)])
def test_bf16_storage_strided_and_transposed(eng, N, Cin, Cout, H, W):
    """bf16 storage through the bf16 gather kernel: stride-2 convolution (norm-on-load source) and ConvTranspose2d as a
    1x1 GEMM with the scatter epilogue."""
    g = torch.Generator().manual_seed(41 + Cin)
    z16 = _b16(rnd(g, N, Cin, H, W))
    scale, shift = rnd(g, Cin) * 0.3 + 1.0, rnd(g, Cin) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    xin = _bf(_transform_cpu(z16.float(), "relu", scale, shift, False))
    ref = F.conv2d(xin, _bf(w), b, stride=2, padding=1)
    node = _mk_node(eng, z16.float(), "relu", scale, shift)
    node.z = node.z.to(torch.bfloat16)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    Ho, Wo = ref.shape[2:]
    out = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    eng.igemm([node.src()], wp, b.cuda(), N, H, W, Ho, Wo, 3, 3, 2, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
    assert rel_err(nchw(out.float().cpu()), ref) < 6e-3
    wt, bt = rnd(g, Cin, Cout, 2, 2) / (Cin ** 0.5), rnd(g, Cout)
    reft = F.conv_transpose2d(xin, _bf(wt), bt, stride=2)
    wpt = eng.pack_weight(wt.cuda(), 4, Cout, Cin, 1, 4, Cout * 4, merge_taps=True)
    outt = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    eng.igemm([node.src()], wpt, bt.cuda(), N, H, W, H, W, 1, 1, 1, 0, eng.MODE_CONV, 4 * Cout, outt, Cout,
              epi=eng.EPI_SCATTER2X2, Cq=Cout, precision="bf16")
    assert rel_err(nchw(outt.float().cpu()), reft) < 6e-3


@pytest.mark.parametrize("N,Cin,Cout,H,W,two_src,act", [(2, 64, 64, 16, 16, False, "relu"), (3, 72, 40, 16, 12, False, "relu"),
                                                        (2, 128, 64, 16, 16, True, "relu"), (2, 64, 64, 24, 32, False, "mish")])
def test_bf16_storage_wgrad(eng, N, Cin, Cout, H, W, two_src, act):
    """weight gradient of a 3x3 stride-1 conv from bf16-stored dz (plain) and z (norm-on-load): exact against torch on the
    rounded operands up to the fp32 accumulation order."""
    g = torch.Generator().manual_seed(9 + Cin + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    scale, shift = rnd(g, Cin) * 0.3 + 1, rnd(g, Cin) * 0.1
    xin = _bf(_transform_cpu(z16.float(), act, scale, shift, False))
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    gy16 = _b16(rnd(g, N, Cout, H, W))
    F.conv2d(xin, w, None, padding=1).backward(gy16.float())

    def node16(zz, sc, sh):
        n = _mk_node(eng, zz.float(), act, sc, sh)
        n.z = n.z.to(torch.bfloat16)
        return n
    if two_src:
        c0 = Cin // 2
        nodes = [node16(z16[:, :c0], scale[:c0], shift[:c0]), node16(z16[:, c0:], scale[c0:], shift[c0:])]
    else:
        nodes = [node16(z16, scale, shift)]
    qs = [n.src() for n in nodes]
    dz = nhwc(gy16).cuda()
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    eng.wgrad(eng.plain_src(dz, Cout), qs, dW, N, H, W, H, W, 3, 3, 1, 1, eng.Workspace(torch.device("cuda")),
              precision="bf16")
    assert rel_err(dW.cpu(), w.grad) < 5e-4


@pytest.mark.parametrize("N,H,W,act,per_sample,Cout,dst32", [
    (8, 128, 128, "relu", False, 64, False), (15, 120, 80, "relu", True, 64, False), (9, 122, 128, "mish", False, 64, False),
    (40, 60, 56, "none", False, 48, False), (8, 128, 128, "relu", False, 64, True)])
def test_bf16_storage_persistent_64_channel_layers(eng, N, H, W, act, per_sample, Cout, dst32):
    """64 -> 64 channel 3x3 layers on bf16 tensors with enough pixel tiles take the persistent kernel (weights of all nine
    taps resident in LDS, two pixel tiles per step): forward with norm-on-load sources (per-channel and per-sample tables,
    cheap and expensive activations), data gradient with a plain operand and accumulate, bf16 and fp32 destinations, tiles
    narrower than 32 pixels,
    image heights that are no multiple of the tile height, odd tile counts, fewer than 64 output channels.  Checked against
    torch on the rounded operands AND against the tile-per-workgroup kernel (mseg_igemm_set_persistent(0))."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    Cin = 64
    g = torch.Generator().manual_seed(900 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    nsc = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *nsc) * 0.3 + 1.0, rnd(g, *nsc) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(w), b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    node = _mk_node(eng, z16.float(), act, scale, shift, per_sample)
    node.z = node.z.to(torch.bfloat16)
    src = node.src()
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.float32 if dst32 else torch.bfloat16)
            eng.igemm([src], wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    tol = 6e-3 if act != "mish" else 8e-3               # one bf16 rounding of the stored result (+ the fast Mish flavour)
    assert rel_err(nchw(outs[0]), ref) < tol
    assert rel_err(outs[0], outs[1]) < 8e-3              # same operands, another fp32 accumulation order: at most one bf16 ulp
    if Cout != 64:
        return
    # data gradient (transposed convolution): plain bf16 operand, accumulate into a bf16 destination
    gy16 = _b16(rnd(g, N, Cout, H, W))
    refn = nhwc(F.conv_transpose2d(gy16.float(), _bf(w), None, padding=1))
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    base = _b16(rnd(g, N, H, W, Cin))
    if dst32:
        base = base.float()
    dz = nhwc(gy16).cuda()
    got = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            d0 = base.clone().cuda()
            eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, Cin, acc0=1,
                      precision="bf16")
            got.append(d0.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    assert rel_err(got[0], refn + base.float()) < 6e-3
    assert rel_err(got[0], got[1]) < 8e-3


@pytest.mark.parametrize("N,H,W,act,per_sample", [(16, 128, 128, "relu", False), (26, 120, 80, "relu", True), (18, 150, 96, "mish", False)])
def test_bf16_storage_512_pixel_tiles_concat_layers(eng, N, H, W, act, per_sample):
    """128 -> 64 channel concat convolutions on bf16 tensors (two 64-channel sources) with enough pixels take 512-pixel tiles
    (igemm_halo_bf16m512_kernel: four pixel tiles per weight stage): checked against torch on the rounded operands and
    against the 128-pixel-tile kernel (mseg_igemm_set_persistent(0) switches both level-0 specialisations off); image
    heights that are no multiple of the tile height, 16- and 32-pixel-wide tiles, per-sample tables, an expensive activation."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    Cin, Cout = 128, 64
    g = torch.Generator().manual_seed(1700 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    nsc = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *nsc) * 0.3 + 1.0, rnd(g, *nsc) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(w), b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    c0 = Cin // 2
    sl = (lambda t, a, b_: t[:, a:b_]) if per_sample else (lambda t, a, b_: t[a:b_])
    nodes = []
    for a, b_ in ((0, c0), (c0, Cin)):
        n = _mk_node(eng, z16[:, a:b_].float(), act, sl(scale, a, b_).contiguous(), sl(shift, a, b_).contiguous(), per_sample)
        n.z = n.z.to(torch.bfloat16)
        nodes.append(n)
    srcs = [n.src() for n in nodes]
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
            eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    assert rel_err(nchw(outs[0]), ref) < (6e-3 if act != "mish" else 8e-3)
    assert rel_err(outs[0], outs[1]) < 8e-3              # same operands, another fp32 accumulation order: at most one bf16 ulp
    # the shapes are chosen so that the 512-pixel tiles are really taken (>= 2 tiles per compute unit)
    tw = 32 if W % 32 == 0 else 16
    assert N * (-(-H // (512 // tw))) * (W // tw) >= 2 * torch.cuda.get_device_properties(0).multi_processor_count


@pytest.mark.parametrize("N,Cin,Cout,H,W,act,per_sample,two_src", [
    (16, 128, 128, 128, 128, "relu", False, False), (12, 256, 256, 80, 80, "relu", True, True),
    (16, 128, 256, 72, 64, "mish", False, False), (26, 192, 256, 64, 40, "none", False, False)])
def test_bf16_storage_256_pixel_tiles(eng, N, Cin, Cout, H, W, act, per_sample, two_src):
    """128-channel-tile layers on bf16 tensors with enough pixels take 256-pixel tiles (igemm_halo_bf16w4m_kernel: half-chunk
    weight stages, tables through LDS): forward with norm-on-load sources (one and two sources, per-channel and per-sample
    tables, cheap and expensive activations) and data gradient with a plain operand, 32- / 16- / 8-pixel-wide tiles, image
    heights that are no multiple of the tile height; against torch on the rounded operands and against the 128-pixel-tile
    kernel (mseg_igemm_set_wide_tiles(0))."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    tw = next(t for t in (32, 16, 8) if W % t == 0)      # the shapes are chosen so that the 256-pixel tiles are really taken
    th = 256 // tw
    assert N * (-(-H // th)) * (W // tw) * (-(-Cout // 128)) >= 2 * torch.cuda.get_device_properties(0).multi_processor_count
    assert H * 5 >= (-(-H // th)) * th * 4
    g = torch.Generator().manual_seed(2100 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    nsc = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *nsc) * 0.3 + 1.0, rnd(g, *nsc) * 0.1
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(w), b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    sl = (lambda t, a, b_: t[:, a:b_]) if per_sample else (lambda t, a, b_: t[a:b_])
    cuts = ((0, Cin // 2), (Cin // 2, Cin)) if two_src else ((0, Cin),)
    nodes = []
    for a, b_ in cuts:
        n = _mk_node(eng, z16[:, a:b_].float(), act, sl(scale, a, b_).contiguous(), sl(shift, a, b_).contiguous(), per_sample)
        n.z = n.z.to(torch.bfloat16)
        nodes.append(n)
    srcs = [n.src() for n in nodes]
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_wide_tiles(on) == 0
        try:
            out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
            eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_wide_tiles(1)
    assert rel_err(nchw(outs[0]), ref) < (6e-3 if act != "mish" else 8e-3)
    assert rel_err(outs[0], outs[1]) < 8e-3              # same operands, another fp32 accumulation order: at most one bf16 ulp
    # data gradient: plain bf16 operand, accumulate into a bf16 destination
    gy16 = _b16(rnd(g, N, Cout, H, W))
    refn = nhwc(F.conv_transpose2d(gy16.float(), _bf(w), None, padding=1))
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    base = _b16(rnd(g, N, H, W, Cin))
    dz = nhwc(gy16).cuda()
    got = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_wide_tiles(on) == 0
        try:
            d0 = base.clone().cuda()
            eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, Cin, acc0=1,
                      precision="bf16")
            got.append(d0.float().cpu())
        finally:
            lib.mseg_igemm_set_wide_tiles(1)
    assert rel_err(got[0], refn + base.float()) < 6e-3
    assert rel_err(got[0], got[1]) < 8e-3


@pytest.mark.parametrize("N,Cin,Cout,H,W,in_act,out_act,dst16,want", [
    (32, 128, 128, 96, 96, "relu", "relu", True, "igemm_p8_kernel<512, 128, 1, true>"),     # four wave rows per 512-pixel tile
    (40, 64, 128, 48, 48, "none", "none", True, "igemm_p8_kernel<256, 128, 0, true>"),      # plain operand, no activation
    (32, 128, 256, 72, 64, "relu", "relu", True, "igemm_p8_kernel<256, 256, 1, true>"),     # tile rows below the image
    (40, 256, 512, 40, 40, "relu", "relu", True, "igemm_p8_kernel<256, 256, 1, true>"),     # image-wide tiles: 16 dead GEMM rows
    (36, 512, 1024, 32, 32, "relu", "none", False, "igemm_p8_kernel<256, 256, 1, true>")])  # four channel tiles, fp32 output
def test_conv_epilogue_statistics(eng, N, Cin, Cout, H, W, in_act, out_act, dst16, want):
    """MsegIgemm.stats: the convolution's epilogue leaves per-tile sums of act(z) and act(z)^2 of the values AS STORED, and
    mseg_norm_stats_from_conv turns them into the BatchNorm tables without reading z again.  Against mseg_norm_stats on the
    stored tensor: the same sums in another order (fp32 over <= 128 pixels, then fp64), so tables, saved mean / rstd and
    running statistics agree to fp32 rounding; z itself is bit-identical to a launch without statistics; the query
    reports the partial rows, and a kernel without the feature reports none and leaves the buffer alone."""
    import ctypes as C
    from microbeseg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(4100 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    affine = in_act != "none"
    scale = rnd(g, Cin) * 0.3 + 1.0 if affine else None
    shift = rnd(g, Cin) * 0.1 if affine else None
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout) * 0.5
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    node = _mk_node(eng, z16.float(), in_act, scale, shift, False)
    node.z = node.z.to(torch.bfloat16)
    srcs = [node.src()]
    dt = torch.bfloat16 if dst16 else torch.float32
    ws = eng.Workspace(torch.device("cuda"))
    out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=dt)
    old = eng.set_conv_stats(True)
    try:
        taken = eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16",
                          stats=(eng.ACT[out_act], ws))
    finally:
        eng.set_conv_stats(old)
    torch.cuda.synchronize()
    assert lib.mseg_last_kernel().decode() == want
    assert taken is not None
    part, rows = taken
    tiles = {512: 4, 256: 2}[int(want.split("<")[1].split(",")[0])]          # wave rows per tile (P8Cfg::WM)
    assert rows % tiles == 0 and rows * 2 * Cout * 4 <= part.numel() * part.element_size()
    plain = torch.full_like(out, float("nan"))
    assert eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, plain, Cout,
                     precision="bf16") is None
    assert torch.equal(out, plain), "the statistics must not change what is stored"
    gamma, beta = (rnd(g, Cout) * 0.2 + 1.0).cuda(), (rnd(g, Cout) * 0.1).cuda()
    res = []
    for fused in (True, False):
        t = {k: torch.empty(Cout, device="cuda") for k in ("scale", "shift", "mean", "rstd")}
        rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
        nws = ws.get("norm", lib.mseg_norm_workspace_bytes(N, H * W, Cout), zero=True)
        if fused:
            rc = lib.mseg_norm_stats_from_conv(part.data_ptr(), rows, Cout, N * H * W, gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                               t["scale"].data_ptr(), t["shift"].data_ptr(), t["mean"].data_ptr(),
                                               t["rstd"].data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, nws.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
        else:
            rc = lib.mseg_norm_stats(out.data_ptr(), N, H * W, Cout, _lib.ST_BF16 if dst16 else _lib.ST_F32, eng.ACT[out_act],
                                     eng.NORM["bn"], gamma.data_ptr(), beta.data_ptr(), 1e-5, t["scale"].data_ptr(),
                                     t["shift"].data_ptr(), t["mean"].data_ptr(), t["rstd"].data_ptr(), rm.data_ptr(),
                                     rv.data_ptr(), 0.1, None, nws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        res.append({**{k: v.cpu() for k, v in t.items()}, "rm": rm.cpu(), "rv": rv.cpu()})
    a = out.float().cpu()
    a = torch.relu(a) if out_act == "relu" else a
    mean_ref = a.double().mean(dim=(0, 1, 2))
    assert rel_err(res[1]["mean"], mean_ref) < 1e-5                        # the pass itself, as a sanity anchor
    for k in res[0]:
        assert rel_err(res[0][k], res[1][k]) < 2e-6, k
    # a launch whose kernel has no statistics epilogue: reported as such, buffer untouched
    lib.mseg_igemm_set_p8(0)
    try:
        marker = torch.full((16,), 7.0, device="cuda")
        p = eng.MsegIgemm()
        eng._fill_igemm(p, srcs, wp.Kpad, wp.Npad, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, Cout, 0, 0, 0, None,
                        eng.EPI_PLAIN, 0, eng.MORDER_LINEAR)
        p.w, p.bias, p.dst0 = wp.bf16().data_ptr(), b.cuda().data_ptr(), plain.data_ptr()
        p.precision, p.dst_dtype = 1, _lib.ST_BF16 if dst16 else _lib.ST_F32
        p.stats, p.stats_act = marker.data_ptr(), eng.ACT[out_act]
        info = _lib.MsegKernelInfo()
        assert lib.mseg_igemm_query(C.byref(p), C.byref(info)) == 0 and info.stats_rows == 0
    finally:
        lib.mseg_igemm_set_p8(1)


@pytest.mark.parametrize("N,Cin,Cout,H,W,act,per_sample,two_src,want", [
    (32, 128, 128, 96, 96, "relu", False, False, "igemm_p8_kernel<512, 128, 1, false>"),    # 32-wide tiles, 512 x 128
    (20, 256, 256, 80, 80, "relu", True, True, "igemm_p8_kernel<256, 256, 1, false>"),       # 16-wide tiles, per-sample tables, concat
    (32, 128, 256, 72, 64, "mish", False, False, "igemm_p8_kernel<256, 256, 2, false>"),     # tile rows below the image (72 = 9 x 8)
    (40, 64, 128, 48, 48, "none", False, False, "igemm_p8_kernel<256, 128, 0, false>"),      # plain operand, 256 x 128 tiles
    (36, 512, 1024, 32, 32, "elu", False, False, "igemm_p8_kernel<256, 256, 2, false>"),     # four 256-channel tiles per pixel tile
    (40, 256, 512, 40, 40, "relu", False, True, "igemm_p8_kernel<256, 256, 1, false>"),      # image-wide tiles 40 x 6 (240 live rows)
    (80, 512, 1024, 20, 20, "mish", True, False, "igemm_p8_kernel<256, 256, 2, false>"),     # image-wide tiles 20 x 12
    (64, 128, 256, 44, 44, "none", False, False, "igemm_p8_kernel<256, 256, 0, false>")])    # 44 x 5 (220 live rows), 9 tile rows
def test_bf16_storage_p8_kernel(eng, N, Cin, Cout, H, W, act, per_sample, two_src, want):
    """Layers with >= 128 output channels on bf16 tensors and at least one tile per CU take the one-workgroup-per-CU kernel
    with DMA-streamed weights (igemm_p8.hip): persistent tile walk (more tiles than CUs, a ragged last round), transposed
    accumulators with 16-byte stores, tables through LDS.  Forward with norm-on-load sources against torch on the rounded
    operands and against the tile-per-workgroup kernels (mseg_igemm_set_p8(0)); data gradient into TWO accumulating bf16
    destinations (the concat split) and into an fp32 destination; the library reports the kernel it took."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3100 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    affine = act != "none"
    nsc = (N, Cin) if per_sample else (Cin,)
    scale = rnd(g, *nsc) * 0.3 + 1.0 if affine else None
    shift = rnd(g, *nsc) * 0.1 if affine else None
    w, b = rnd(g, Cout, Cin, 3, 3) / (3 * Cin ** 0.5), rnd(g, Cout)
    ref = F.conv2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(w), b, padding=1)
    wp = eng.pack_weight(w.cuda(), 9, Cout, Cin, 1, Cin * 9, 9)
    sl = (lambda t, a, b_: t[:, a:b_]) if per_sample else (lambda t, a, b_: t[a:b_])
    cuts = ((0, Cin // 2), (Cin // 2, Cin)) if two_src else ((0, Cin),)
    nodes = []
    for a, b_ in cuts:
        n = _mk_node(eng, z16[:, a:b_].float(), act, sl(scale, a, b_).contiguous() if affine else None,
                     sl(shift, a, b_).contiguous() if affine else None, per_sample)
        n.z = n.z.to(torch.bfloat16)
        nodes.append(n)
    srcs = [n.src() for n in nodes]
    outs = []
    for mode in (1, 0):
        assert lib.mseg_igemm_set_p8(mode) == 0
        try:
            out = torch.full((N, H, W, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
            eng.igemm(srcs, wp, b.cuda(), N, H, W, H, W, 3, 3, 1, 1, eng.MODE_CONV, Cout, out, Cout, precision="bf16")
            torch.cuda.synchronize()
            if mode == 1:
                assert lib.mseg_last_kernel().decode() == want
            else:
                assert "p8" not in lib.mseg_last_kernel().decode()
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_p8(1)
    assert rel_err(nchw(outs[0]), ref) < (6e-3 if act in ("relu", "none") else 8e-3)
    assert rel_err(outs[0], outs[1]) < 8e-3              # same operands, another fp32 accumulation order: at most one bf16 ulp
    # data gradient (plain operand dz; N = Cin >= 128 only): two accumulating bf16 destinations split at Cin / 2
    if Cin < 128:
        return
    gy16 = _b16(rnd(g, N, Cout, H, W))
    refn = nhwc(F.conv_transpose2d(gy16.float(), _bf(w), None, padding=1))
    wd = eng.pack_weight(w.cuda(), 9, Cin, Cout, 1, 9, Cin * 9)
    dz = nhwc(gy16).cuda()
    half = Cin // 2
    base0, base1 = _b16(rnd(g, N, H, W, half)), _b16(rnd(g, N, H, W, Cin - half))
    got = []
    for mode in (1, 0):
        assert lib.mseg_igemm_set_p8(mode) == 0
        try:
            d0, d1 = base0.clone().cuda(), base1.clone().cuda()
            eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d0, half, acc0=1,
                      dst1=d1, ld1=Cin - half, acc1=1, split=half, precision="bf16")
            torch.cuda.synchronize()
            if mode == 1:
                assert lib.mseg_last_kernel().decode().startswith("igemm_p8_kernel")
            got.append(torch.cat([d0.float().cpu(), d1.float().cpu()], dim=3))
        finally:
            lib.mseg_igemm_set_p8(1)
    want_d = refn + torch.cat([base0.float(), base1.float()], dim=3)
    assert rel_err(got[0], want_d) < 6e-3
    assert rel_err(got[0], got[1]) < 8e-3
    # fp32 destination, no accumulation
    d32 = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    eng.igemm([eng.plain_src(dz, Cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, eng.MODE_TCONV, Cin, d32, Cin, precision="bf16")
    assert lib.mseg_last_kernel().decode().startswith("igemm_p8_kernel")
    assert rel_err(d32.cpu(), refn) < 1e-4


def test_dispatch_query_equals_launch(eng):
    """mseg_igemm_query / mseg_wgrad_query run the dispatch code itself with the launches switched off: for a sweep of layer
    shapes (both precisions, tensor storages, strides, concat, ConvTranspose forms) the kernel a query names is the kernel
    the launch then takes (mseg_last_kernel)."""
    from microbeseg_amd import _lib
    from microbeseg_amd._lib import ST_F32, ST_BF16, ACT
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n_checked = 0
    for prec, st in (("fp32", torch.float32), ("bf16", torch.float32), ("bf16", torch.bfloat16)):
        for (N, cins, cout, H, W, stride) in [(2, (8,), 16, 16, 16, 1), (4, (64,), 64, 64, 64, 1), (32, (128,), 128, 96, 96, 1),
                                              (16, (128, 128), 256, 40, 40, 1), (3, (16,), 8, 20, 12, 1),
                                              (8, (64,), 64, 64, 64, 2), (2, (136,), 72, 10, 14, 1), (32, (64,), 64, 160, 160, 1)]:
            cin = sum(cins)
            if st == torch.bfloat16 and (cin % 8 or cout % 8 or any(c % 8 for c in cins)):
                continue
            Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
            nodes = []
            for c in cins:
                n = _mk_node(eng, rnd(g, N, c, H, W), "relu", rnd(g, c) * 0.1 + 1, rnd(g, c) * 0.1)
                n.z = n.z.to(st)
                nodes.append(n)
            srcs = [n.src() for n in nodes]
            w = rnd(g, cout, cin, 3, 3) * 0.05
            wp = eng.pack_weight(w.cuda(), 9, cout, cin, 1, cin * 9, 9)
            out = torch.empty((N, Ho, Wo, cout), device="cuda", dtype=st)
            q = eng.igemm_query(srcs, wp.Kpad, wp.Npad, N, H, W, Ho, Wo, 3, 3, stride, 1, eng.MODE_CONV, cout, ld0=cout,
                                precision=prec, dst_dtype=eng._st(out), bias=False)
            if q is None:                                # no bf16 kernel: the engine keeps the launch in fp32
                assert prec == "bf16"
                if st == torch.bfloat16:
                    continue
                q = eng.igemm_query(srcs, wp.Kpad, wp.Npad, N, H, W, Ho, Wo, 3, 3, stride, 1, eng.MODE_CONV, cout, ld0=cout,
                                    precision="fp32", dst_dtype=eng._st(out), bias=False)
            eng.igemm(srcs, wp, None, N, H, W, Ho, Wo, 3, 3, stride, 1, eng.MODE_CONV, cout, out, cout, precision=prec)
            assert lib.mseg_last_kernel().decode() == q.name, (prec, st, N, cins, cout, H, W, stride)
            # weight gradient of the same layer
            dz = torch.randn(N, Ho, Wo, cout, device="cuda").to(st)
            P = eng.plain_src(dz, cout)
            qw = eng.wgrad_query(P, srcs, N, Ho, Wo, H, W, 3, 3, stride, 1, precision=prec)
            if qw is None:
                if st == torch.bfloat16:
                    continue
                qw = eng.wgrad_query(P, srcs, N, Ho, Wo, H, W, 3, 3, stride, 1, precision="fp32")
            dW = torch.empty((cout, cin, 3, 3), device="cuda")
            eng.wgrad(P, srcs, dW, N, Ho, Wo, H, W, 3, 3, stride, 1, eng.Workspace(torch.device("cuda")), precision=prec)
            assert lib.mseg_last_kernel().decode() == qw.name, ("wgrad", prec, st, N, cins, cout, H, W, stride)
            n_checked += 1
    assert n_checked >= 12


@pytest.mark.parametrize("N,H,W,act,per_sample,dst32", [(8, 128, 128, "relu", False, False), (16, 96, 96, "mish", True, False),
                                                         (9, 128, 128, "none", False, True)])
def test_bf16_storage_persistent_conv_transpose(eng, N, H, W, act, per_sample, dst32):
    """ConvTranspose2d(128 -> 64, 2, stride 2) on bf16 tensors with enough pixels takes the persistent kernel (resident
    weights, two wave groups walking their own 128-pixel tiles, scatter to the four output positions): against torch on the
    rounded operands and against the gather kernel (mseg_igemm_set_persistent(0)); per-channel and per-sample tables, a
    cheap / an expensive / no activation, bf16 and fp32 destinations, an odd tile count."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    Cin, Cout = 128, 64
    assert N * H * W // 128 >= 4 * torch.cuda.get_device_properties(0).multi_processor_count and W % 32 == 0 and (H * W) % 128 == 0
    g = torch.Generator().manual_seed(3300 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    nsc = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *nsc) * 0.3 + 1.0, rnd(g, *nsc) * 0.1
    wt, bt = rnd(g, Cin, Cout, 2, 2) / (Cin ** 0.5), rnd(g, Cout)
    ref = F.conv_transpose2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(wt), bt, stride=2)
    node = _mk_node(eng, z16.float(), act, scale, shift, per_sample)
    node.z = node.z.to(torch.bfloat16)
    wpt = eng.pack_weight(wt.cuda(), 4, Cout, Cin, 1, 4, Cout * 4, merge_taps=True)
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=torch.float32 if dst32 else torch.bfloat16)
            eng.igemm([node.src()], wpt, bt.cuda(), N, H, W, H, W, 1, 1, 1, 0, eng.MODE_CONV, 4 * Cout, out, Cout,
                      epi=eng.EPI_SCATTER2X2, Cq=Cout, precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    assert rel_err(nchw(outs[0]), ref) < (6e-3 if act != "mish" else 8e-3)
    assert rel_err(outs[0], outs[1]) < 8e-3


@pytest.mark.parametrize("N,h,w,dst32", [(16, 64, 64, False), (11, 64, 96, False), (17, 64, 64, True)])
def test_bf16_storage_persistent_conv_transpose_dgrad(eng, N, h, w, dst32):
    """Data gradient of ConvTranspose2d(128 -> 64, 2, stride 2) = Conv2d(64 -> 128, 2, stride 2) over dz, on bf16 tensors
    with enough pixels: the persistent kernel (resident weights, 64-pixel tiles staged with all four taps) against torch on
    the rounded operands and against the gather kernel (mseg_igemm_set_persistent(0)); bf16 and fp32 destinations, image
    rows of 2 and 3 blocks, an odd tile count."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    Cin, Cout = 128, 64                                  # of the ConvTranspose2d; dz has Cout channels at (2h, 2w)
    assert N * h * w // 64 >= 4 * torch.cuda.get_device_properties(0).multi_processor_count and w % 32 == 0
    g = torch.Generator().manual_seed(3700 + N + w)
    wt = rnd(g, Cin, Cout, 2, 2) / (Cout ** 0.5)
    dz16 = _b16(rnd(g, N, Cout, 2 * h, 2 * w))
    ref = F.conv2d(dz16.float(), _bf(wt), None, stride=2)        # weight (Cin, Cout, 2, 2) read as (out = Cin, in = Cout, 2, 2)
    wp = eng.pack_weight(wt.cuda(), 4, Cin, Cout, 1, Cout * 4, 4)
    dz = nhwc(dz16).cuda()
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            out = torch.full((N, h, w, Cin), float("nan"), device="cuda", dtype=torch.float32 if dst32 else torch.bfloat16)
            eng.igemm([eng.plain_src(dz, Cout)], wp, None, N, 2 * h, 2 * w, h, w, 2, 2, 2, 0, eng.MODE_CONV, Cin, out, Cin,
                      precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    assert rel_err(nchw(outs[0]), ref) < 6e-3
    assert rel_err(outs[0], outs[1]) < 8e-3


@pytest.mark.parametrize("N,H,W,act,per_sample,dst32", [(4, 64, 64, "relu", False, False), (3, 96, 96, "mish", True, False),
                                                         (5, 64, 64, "none", False, True), (4, 80, 80, "relu", False, False),
                                                         (9, 48, 40, "relu", True, False)])
def test_bf16_storage_persistent_conv_transpose_level1(eng, N, H, W, act, per_sample, dst32):
    """ConvTranspose2d(256 -> 128, 2, stride 2) on bf16 tensors: the persistent kernel that keeps the weights of ONE output
    position per workgroup (four workgroups share a tile stream) against torch on the rounded operands and against the gather
    kernel (mseg_igemm_set_persistent(0))."""
    from microbeseg_amd import _lib
    lib = _lib.load()
    Cin, Cout = 256, 128
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert N * H * W // 64 >= 4 * (cus // 4) and W >= 32 and (H * W) % 64 == 0     # rows of 80 / 40 pixels: blocks wrap
    g = torch.Generator().manual_seed(4100 + N + W)
    z16 = _b16(rnd(g, N, Cin, H, W))
    nsc = (N, Cin) if per_sample else (Cin,)
    scale, shift = rnd(g, *nsc) * 0.3 + 1.0, rnd(g, *nsc) * 0.1
    wt, bt = rnd(g, Cin, Cout, 2, 2) / (Cin ** 0.5), rnd(g, Cout)
    ref = F.conv_transpose2d(_bf(_transform_cpu(z16.float(), act, scale, shift, per_sample)), _bf(wt), bt, stride=2)
    node = _mk_node(eng, z16.float(), act, scale, shift, per_sample)
    node.z = node.z.to(torch.bfloat16)
    wpt = eng.pack_weight(wt.cuda(), 4, Cout, Cin, 1, 4, Cout * 4, merge_taps=True)
    outs = []
    for on in (1, 0):
        assert lib.mseg_igemm_set_persistent(on) == 0
        try:
            out = torch.full((N, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=torch.float32 if dst32 else torch.bfloat16)
            eng.igemm([node.src()], wpt, bt.cuda(), N, H, W, H, W, 1, 1, 1, 0, eng.MODE_CONV, 4 * Cout, out, Cout,
                      epi=eng.EPI_SCATTER2X2, Cq=Cout, precision="bf16")
            outs.append(out.float().cpu())
        finally:
            lib.mseg_igemm_set_persistent(1)
    assert rel_err(nchw(outs[0]), ref) < (6e-3 if act != "mish" else 8e-3)
    assert rel_err(outs[0], outs[1]) < 8e-3
