"""GPU box: what a plain streaming kernel reaches on this HBM (torch copy / read-only reduction of 0.4-1.7 GB tensors) —
the yardstick for the normalisation passes of tools/bench_norm.py."""
import time
import torch

for mb in (420, 840, 1680):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n // 2, device="cuda").to(torch.bfloat16).repeat(2)
    b = torch.empty_like(a)
    af = a.view(torch.float32) if False else None

    def timeit(f, reps=20):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    t = timeit(lambda: b.copy_(a))
    print(f"copy   {mb:5d} MB: {t*1e6:8.1f} us  {2*a.numel()*2/t/1e12:5.2f} TB/s (read + write)")
    x = a.view(torch.int32)
    t = timeit(lambda: torch.sum(x))
    print(f"sum    {mb:5d} MB: {t*1e6:8.1f} us  {a.numel()*2/t/1e12:5.2f} TB/s (read only, int32 sum)")
    t = timeit(lambda: b.fill_(1.0))
    print(f"fill   {mb:5d} MB: {t*1e6:8.1f} us  {a.numel()*2/t/1e12:5.2f} TB/s (write only)")
