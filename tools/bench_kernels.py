"""GPU box: per-shape throughput of the MFMA kernels on the layer shapes of DU [64,1024] at batch 32, 256x256."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd import engine as E
from microbeseg_amd._lib import ACT

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "all"
ws = E.Workspace(dev)


def node(N, H, W, Cc, act="relu", bn=True):
    n = E.Node(torch.randn(N, H, W, Cc, device=dev), N, H, W, Cc)
    n.act = ACT[act]
    if bn:
        n.scale = torch.rand(Cc, device=dev) + 0.5
        n.shift = torch.randn(Cc, device=dev) * 0.1
        n.ss = 0
    return n


def timeit(fn, flops, label, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:58s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s", flush=True)
    return ms


shapes = []  # (cins, cout, HW, stride)
lv = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]
shapes.append(((4,), 64, 256, 1))
for i, (c, s) in enumerate(lv):
    if i > 0:
        shapes.append(((c // 2,), c, s, 1))
    shapes.append(((c,), c, s, 1))
    if i < 4:
        shapes.append(((c,), c, s, 2))
for c, s in lv[:4][::-1]:
    shapes.append(((c, c), c, s, 1))

tot_f = tot_d = tot_w = 0.0
for cins, cout, s, stride in shapes:
    cin = sum(cins)
    srcs = [node(B, s, s, c) for c in cins]
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    so = s // stride
    z = torch.empty(B, so, so, cout, device=dev)
    wp = E.pack_weight(w, 9, cout, cin, 1, cin * 9, 9)
    flops = 2.0 * B * so * so * cout * cin * 9
    tag = f"cin{'+'.join(map(str, cins))} cout{cout} {s}x{s} s{stride}"
    if which in ("all", "fwd"):
        tot_f += timeit(lambda: E.igemm([n.src() for n in srcs], wp, bias, B, s, s, so, so, 3, 3, stride, 1, E.MODE_CONV,
                                        cout, z, cout), flops, "fwd   " + tag)
    if cin >= 64 and which in ("all", "dgrad"):
        wd = E.pack_weight(w, 9, cin, cout, 1, 9, cin * 9)
        dx = torch.empty(B, s, s, cin, device=dev)
        mo = E.MORDER_PARITY if stride == 2 else E.MORDER_LINEAR
        tot_d += timeit(lambda: E.igemm([E.plain_src(z, cout)], wd, None, B, so, so, s, s, 3, 3, stride, 1, E.MODE_TCONV,
                                        cin, dx, cin, morder=mo), flops, "dgrad " + tag)
    if which in ("all", "wgrad"):
        dW = torch.empty(cout, cin, 3, 3, device=dev)
        tot_w += timeit(lambda: E.wgrad(E.plain_src(z, cout), [n.src() for n in srcs], dW, B, so, so, s, s, 3, 3, stride,
                                        1, ws), flops, "wgrad " + tag)
print(f"totals (ms): fwd {tot_f:.1f} dgrad {tot_d:.1f} wgrad {tot_w:.1f}")
