"""GPU box: time the normalisation passes (stats / backward) on level-0..2 shapes in fp32 and bf16 storage -> GB/s."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
from microbeseg_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
import os
lib.mseg_norm_set_tails(int(os.environ.get("TAILS", "1")))
for (N, H, C) in [(32, 320, 64), (32, 160, 128), (32, 80, 256), (32, 40, 512), (32, 20, 1024), (4, 256, 64), (4, 16, 1024)]:
    for dt, code in ((torch.bfloat16, 1),):
        HW = H * H
        z = torch.randn(N, HW, C, device="cuda").to(dt)
        gy = torch.randn(N, HW, C, device="cuda").to(dt)
        dz = torch.empty_like(z)
        scale, shift, mean, rstd = (torch.empty(C, device="cuda") for _ in range(4))
        gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        dg, db, dbias = (torch.empty(C, device="cuda") for _ in range(3))
        ws = torch.zeros(lib.mseg_norm_workspace_bytes(N, HW, C), dtype=torch.uint8, device="cuda")
        def fwd():
            _lib.check(lib.mseg_norm_stats(z.data_ptr(), N, HW, C, code, 1, 0, gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                           scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None,
                                           0.1, None, ws.data_ptr(), st))
        def bwd():
            _lib.check(lib.mseg_norm_bwd(gy.data_ptr(), z.data_ptr(), N, HW, C, code, 1, 0, gamma.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), dz.data_ptr(), dg.data_ptr(), db.data_ptr(), dbias.data_ptr(),
                                         None, ws.data_ptr(), st))
        for f, nbytes, name in ((fwd, 1, "stats"), (bwd, 5, "bwd")):
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): f()
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
            gb = nbytes * z.numel() * z.element_size() / 1e9
            print(f"N{N} {H}x{H} C{C} {str(dt)[6:]:9s} {name:5s} {dtm*1e3:7.3f} ms  {gb/dtm/1e3:5.2f} TB/s")
