"""Write-only HBM bandwidth reference: torch fill / copy at the observation tensor sizes."""
import torch, json
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for mb in (257, 642, 1500):
    n = mb * 1000 * 1000
    x = torch.empty(n, dtype=torch.int8, device='cuda')
    y = torch.empty(n, dtype=torch.int8, device='cuda')
    xi = x.view(torch.int32)
    us = t(lambda: xi.fill_(1))
    us2 = t(lambda: x.zero_())
    us3 = t(lambda: y.copy_(x))
    print(json.dumps(dict(MB=mb, fill_us=round(us, 1), fill_TBps=round(n / us / 1e6, 2), memset_us=round(us2, 1), memset_TBps=round(n / us2 / 1e6, 2),
                          copy_us=round(us3, 1), copy_TBps_rw=round(2 * n / us3 / 1e6, 2))), flush=True)
