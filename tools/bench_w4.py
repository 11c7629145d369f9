"""GPU box: the 128-channel-tile bf16 halo kernel (igemm_halo_bf16w4) on the 320x320 layer shapes, bf16 tensor storage."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd import engine as E
from microbeseg_amd._lib import ACT
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32


def node(N, H, W, Cc):
    n = E.Node(torch.randn(N, H, W, Cc, device=dev).to(st), N, H, W, Cc)
    n.act = ACT["relu"]
    n.scale = torch.rand(Cc, device=dev) + 0.5
    n.shift = torch.randn(Cc, device=dev) * 0.1
    n.ss = 0
    return n


def timeit(fn, flops, label, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:44s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s", flush=True)
    return ms


tf = td = 0.0
for cins, cout, s in [((128,), 128, 160), ((256,), 256, 80), ((512,), 512, 40), ((1024,), 1024, 20), ((512, 512), 512, 40),
                      ((256, 256), 256, 80), ((128, 128), 128, 160), ((128,), 256, 80)]:
    cin = sum(cins)
    srcs = [node(B, s, s, c) for c in cins]
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    z = torch.empty(B, s, s, cout, device=dev, dtype=st)
    wp = E.pack_weight(w, 9, cout, cin, 1, cin * 9, 9)
    flops = 2.0 * B * s * s * cout * cin * 9
    tag = f"cin{'+'.join(map(str, cins))} cout{cout} {s}x{s}"
    tf += timeit(lambda: E.igemm([n.src() for n in srcs], wp, bias, B, s, s, s, s, 3, 3, 1, 1, E.MODE_CONV, cout, z, cout,
                                 precision="bf16"), flops, "fwd   " + tag)
    wd = E.pack_weight(w, 9, cin, cout, 1, 9, cin * 9)
    dx = torch.empty(B, s, s, cin, device=dev, dtype=st)
    if cin >= 128:
        td += timeit(lambda: E.igemm([E.plain_src(z, cout)], wd, None, B, s, s, s, s, 3, 3, 1, 1, E.MODE_TCONV, cin, dx, cin,
                                     precision="bf16"), flops, "dgrad " + tag)
print(f"totals (ms): fwd {tf:.2f} dgrad {td:.2f}")
