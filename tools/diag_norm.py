import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from microbeseg_amd import engine as eng, _lib
from microbeseg_amd._lib import ACT, NORM
lib = _lib.load()
N, Cc, H, W = 4, 64, 96, 96
g = torch.Generator().manual_seed(1)
z = (torch.randn(N, Cc, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
gamma = (torch.randn(Cc, generator=g) * 0.3 + 1).requires_grad_(True)
beta = (torch.randn(Cc, generator=g) * 0.1).requires_grad_(True)
y = F.batch_norm(F.relu(z), None, None, gamma, beta, True, 0.1, 1e-5)
gy = torch.randn(y.shape, generator=g)
y.backward(gy)
ws = eng.Workspace(torch.device("cuda"))
for mode in (1, 0, 1, 0):
    lib.mseg_norm_set_tails(mode)
    node = eng.Node(z.detach().permute(0, 2, 3, 1).contiguous().cuda(), N, H, W, Cc)
    node.act = ACT["relu"]
    eng.norm_stats(node, NORM["bn"], gamma.detach().cuda(), beta.detach().cuda(), None, None, True, ws)
    dg, db, dbias = (torch.empty(Cc, device="cuda") for _ in range(3))
    dz = eng.norm_bwd(node, gy.permute(0, 2, 3, 1).contiguous().cuda(), gamma.detach().cuda(), dg, db, dbias, ws)
    torch.cuda.synchronize()
    e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    print(mode, "dz", e(dz.cpu().permute(0, 3, 1, 2), z.grad), "dgamma", e(dg.cpu(), gamma.grad), "dbeta", e(db.cpu(), beta.grad),
          "dbias", e(dbias.cpu(), z.grad.sum((0, 2, 3))), "ctr", int(ws.buf["norm"][:65536].view(torch.int32).abs().max()))
