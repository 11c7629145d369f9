"""GPU box: ablation of the igemm K-loop (needs microbeseg_amd/_build/libmseg_hip_ablate.so, built with -DMSEG_ABLATE)."""
import os, sys, pathlib
root = pathlib.Path(__file__).resolve().parents[1]
os.environ["MSEG_HIP_LIB"] = str(root / "microbeseg_amd" / "_build" / "libmseg_hip_ablate.so")
sys.path.insert(0, str(root))
import torch
from microbeseg_amd import engine as E
from microbeseg_amd._lib import ACT
import microbeseg_amd._lib as L
import ctypes as C
dev = torch.device("cuda")
B = 32
def run(cin, cout, s, flags, plain=False):
    n = E.Node(torch.randn(B, s, s, cin, device=dev), B, s, s, cin)
    if not plain:
        n.act = ACT["relu"]; n.scale = torch.rand(cin, device=dev) + 0.5; n.shift = torch.randn(cin, device=dev) * 0.1
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    wp = E.pack_weight(w, 9, cout, cin, 1, cin * 9, 9)
    z = torch.empty(B, s, s, cout, device=dev)
    bias = torch.randn(cout, device=dev)
    dbg = torch.zeros(1024 * 16, device=dev)
    f = lambda: E.igemm([n.src()], wp, bias, B, s, s, s, s, 3, 3, 1, 1, E.MODE_CONV, cout, z, cout, Cq=flags << 16,
                        dst1=dbg, split=1 << 30)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    if flags & 32:
        d = dbg.view(1024, 4, 4)[:min(1024, (B * s * s // 128) * ((cout + 127) // 128))].mean((0, 1)).tolist()
        steps = 9 * cin // 32
        print("   per-step cycles: issue %.0f  mfma %.0f  commit %.0f  barrier %.0f   (total %.0f)" % tuple([x / steps for x in d] + [sum(d) / steps]))
    return ms, 2.0 * B * s * s * cout * cin * 9 / ms / 1e9
names = {32: "stamped", 16: "stagger", 0: "full", 1: "no global loads", 2: "no barriers", 4: "no MFMA", 8: "no commit", 9: "no loads+commit", 11: "no loads/commit/barrier", 12: "no MFMA, no commit", 6: "no MFMA no barrier"}
for cin, cout, s in ((512, 512, 32), (128, 128, 128), (64, 64, 256)):
    for fl in (32,):
        ms, tf = run(cin, cout, s, fl)
        print(f"cin{cin} cout{cout} {s}x{s}  {names[fl]:26s} {ms:7.3f} ms  {tf:6.1f} TF/s(nominal)", flush=True)
