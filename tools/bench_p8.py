"""GPU box: igemm_p8_kernel (one 8-wave workgroup per CU, DMA-streamed weights) against the tile-per-workgroup kernels.

  python tools/bench_p8.py check        correctness: p8 on == p8 off (to one bf16 ulp) and run-to-run bit-identical
  python tools/bench_p8.py bench [B]    per-layer timings of the 320^2 / 256^2 training shapes, p8 modes 0 / 1 / 2
"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd import engine as E, _lib
from microbeseg_amd._lib import ACT

dev = torch.device("cuda")
lib = _lib.load()
st = torch.bfloat16


def node(N, H, W, Cc, act="relu", per_sample=False, affine=True, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    n = E.Node(torch.randn(N, H, W, Cc, device=dev, generator=g).to(st), N, H, W, Cc)
    n.act = ACT[act]
    if affine:
        shp = (N, Cc) if per_sample else (Cc,)
        n.scale = torch.rand(*shp, device=dev, generator=g) + 0.5
        n.shift = torch.randn(*shp, device=dev, generator=g) * 0.1
        n.ss = Cc if per_sample else 0
    return n


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


def run_pair(fn):
    outs = []
    for mode in (1, 0):
        lib.mseg_igemm_set_p8(mode)
        try:
            outs.append(fn())
        finally:
            lib.mseg_igemm_set_p8(1)
    return outs


def check():
    torch.manual_seed(0)
    bad = 0
    cases = [  # N, H, W, cins, cout, act, per_sample, affine
        (32, 64, 64, (256,), 256, "relu", False, True),        # TW 32, BN 256
        (32, 80, 80, (128,), 256, "relu", False, True),        # TW 16
        (20, 80, 80, (256, 256), 256, "relu", True, True),     # two sources, per-sample tables
        (16, 72, 64, (128,), 256, "mish", False, True),        # generic activation, tile rows below the image (72 = 9 x 8)
        (32, 160, 160, (128,), 128, "relu", False, True),      # BN 128, 512-pixel tiles
        (32, 160, 160, (128, 128), 128, "none", False, False), # plain operand (TR 0), two sources
        (40, 48, 48, (64,), 128, "relu", False, True),         # TW 16, BN 128, 256-pixel tiles
        (64, 40, 32, (512,), 512, "elu", False, True),         # TW 32, 40 rows = 5 x 8
        (32, 32, 32, (512,), 1024, "relu", False, True),
        (32, 40, 40, (512,), 512, "relu", False, True),        # image-wide tiles: 40 x 6 pixels (240 of 256 GEMM rows live)
        (32, 20, 20, (512, 512), 1024, "relu", True, True),    # 20 x 12 pixels, two sources, per-sample tables
        (64, 44, 44, (128,), 256, "mish", False, True),        # 44 x 5 pixels
        (48, 52, 24, (256,), 256, "none", False, False),       # 24 x 10 pixels, plain operand
    ]
    for (N, H, W, cins, cout, act, ps, aff) in cases:
        cin = sum(cins)
        srcs = [node(N, H, W, c, act, ps, aff, seed=7 + i) for i, c in enumerate(cins)]
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        bias = torch.randn(cout, device=dev)
        wp = E.pack_weight(w, 9, cout, cin, 1, cin * 9, 9)

        def fwd():
            z = torch.full((N, H, W, cout), float("nan"), device=dev, dtype=st)
            E.igemm([n.src() for n in srcs], wp, bias, N, H, W, H, W, 3, 3, 1, 1, E.MODE_CONV, cout, z, cout, precision="bf16")
            torch.cuda.synchronize()
            return z
        a, b = run_pair(fwd)
        e = rel(a.float(), b.float())
        reps = [fwd() for _ in range(4)]
        same = all(torch.equal(reps[0].view(torch.int16), r.view(torch.int16)) for r in reps[1:]) and \
            torch.equal(reps[0].view(torch.int16), a.view(torch.int16))
        ok = e < 8e-3 and same and torch.isfinite(a.float()).all().item()
        bad += not ok
        print(f"fwd   N{N} {H}x{W} cin{cins} cout{cout} {act} ps{int(ps)} aff{int(aff)}: vs old {e:.2e} deterministic {same} {'ok' if ok else 'FAIL'} [{lib.mseg_last_kernel().decode()}]", flush=True)
        # data gradient: plain operand dz, accumulate into a bf16 destination; two destinations when there are two sources
        if cin >= 128:
            wd = E.pack_weight(w, 9, cin, cout, 1, 9, cin * 9)
            dz = torch.randn(N, H, W, cout, device=dev).to(st)
            base = torch.randn(N, H, W, cin, device=dev).to(st)

            def dgrad():
                d0 = base.clone()
                E.igemm([E.plain_src(dz, cout)], wd, None, N, H, W, H, W, 3, 3, 1, 1, E.MODE_TCONV, cin, d0, cin, acc0=1,
                        precision="bf16")
                torch.cuda.synchronize()
                return d0
            a, b = run_pair(dgrad)
            e = rel(a.float(), b.float())
            reps = [dgrad() for _ in range(3)]
            same = all(torch.equal(reps[0].view(torch.int16), r.view(torch.int16)) for r in reps[1:])
            ok = e < 8e-3 and same
            bad += not ok
            print(f"dgrad N{N} {H}x{W} cin{cin} cout{cout}: vs old {e:.2e} deterministic {same} {'ok' if ok else 'FAIL'} [{lib.mseg_last_kernel().decode()}]", flush=True)
    print("CHECK", "FAILED" if bad else "PASSED", bad)
    return bad


def timeit(fn, flops, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, flops / ms / 1e9


MODES = (0, 1, 2)


def bench(B):
    shapes = [((128,), 128, 160), ((256,), 256, 80), ((512,), 512, 40), ((1024,), 1024, 20), ((512, 512), 512, 40),
              ((256, 256), 256, 80), ((128, 128), 128, 160), ((128,), 256, 80), ((64,), 128, 160),
              ((128,), 128, 128), ((256,), 256, 64), ((512,), 512, 32), ((256, 256), 256, 64)]
    tot = {m: [0.0, 0.0] for m in MODES}
    for cins, cout, s in shapes:
        cin = sum(cins)
        srcs = [node(B, s, s, c) for c in cins]
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        z = torch.empty(B, s, s, cout, device=dev, dtype=st)
        wp = E.pack_weight(w, 9, cout, cin, 1, cin * 9, 9)
        wd = E.pack_weight(w, 9, cin, cout, 1, 9, cin * 9)
        dx = torch.empty(B, s, s, cin, device=dev, dtype=st)
        flops = 2.0 * B * s * s * cout * cin * 9
        line = f"cin{'+'.join(map(str, cins)):8s} cout{cout:5d} {s:3d}^2 |"
        for mode in MODES:
            lib.mseg_igemm_set_p8(mode)
            f = timeit(lambda: E.igemm([n.src() for n in srcs], wp, bias, B, s, s, s, s, 3, 3, 1, 1, E.MODE_CONV, cout, z, cout,
                                       precision="bf16"), flops)
            d = timeit(lambda: E.igemm([E.plain_src(z, cout)], wd, None, B, s, s, s, s, 3, 3, 1, 1, E.MODE_TCONV, cin, dx, cin,
                                       precision="bf16"), flops) if cin >= 128 else (0.0, 0.0)
            tot[mode][0] += f[0]; tot[mode][1] += d[0]
            line += f" m{mode}: fwd {f[0]*1e3:6.0f}us {f[1]:6.0f}TF dgrad {d[0]*1e3:6.0f}us {d[1]:6.0f}TF |"
        lib.mseg_igemm_set_p8(1)
        print(line, flush=True)
    for mode in MODES:
        print(f"mode {mode}: fwd {tot[mode][0]:.2f} ms dgrad {tot[mode][1]:.2f} ms")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        sys.exit(1 if check() else 0)
    if len(sys.argv) > 3:
        MODES = tuple(int(v) for v in sys.argv[3].split(","))
    bench(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
