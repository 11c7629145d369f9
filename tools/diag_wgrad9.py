"""GPU box diagnostic: wgrad_halo9_kernel vs torch on a few shapes; prints NaN counts / errors per tap."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from microbeseg_amd import engine as eng

def run(N, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xq = x.permute(0, 2, 3, 1).contiguous().cuda()      # keep alive: MsegSrc holds a raw pointer
    q = eng.plain_src(xq, Cin)
    dz = gy.permute(0, 2, 3, 1).contiguous().cuda()
    P = eng.plain_src(dz, Cout)
    dW = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    name = eng.wgrad_query(P, [q], N, H, W, H, W, 3, 3, 1, 1).name
    eng.wgrad(P, [q], dW, N, H, W, H, W, 3, 3, 1, 1, eng.Workspace(torch.device("cuda")))
    torch.cuda.synchronize()
    d = dW.cpu()
    nan = torch.isnan(d)
    err = (d - w.grad).abs()
    err[nan] = 0
    print(N, Cin, Cout, H, W, name, "nan per tap", nan.sum((0, 1)).flatten().tolist(), "max err per tap",
          [round(v, 5) for v in err.amax((0, 1)).flatten().tolist()], "ref max", w.grad.abs().max().item())

for a in [(1, 64, 64, 4, 8), (4, 64, 64, 32, 32), (4, 64, 64, 32, 32), (1, 64, 64, 4, 8), (2, 64, 64, 16, 32), (1, 128, 128, 32, 32)]:
    run(*a)
