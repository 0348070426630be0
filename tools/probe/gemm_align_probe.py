"""Does the GRU input projection run faster with the feature rows padded from 610 to 612 floats (16-byte row pitch)?"""
import torch


def t(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for R in (16384, 81920):
    for K in (610, 612, 616, 640):
        x = torch.randn(R, K, device='cuda')
        w = torch.randn(384, K, device='cuda')
        g = torch.randn(R, 384, device='cuda')
        fwd = t(lambda: torch.matmul(x, w.t()))
        dgrad = t(lambda: torch.matmul(g, w))
        wgrad = t(lambda: torch.matmul(g.t(), x))
        fl = 2.0 * R * K * 384
        print('rows %6d K %3d: fwd %7.1f us (%5.1f TF)  dgrad %7.1f us  wgrad %7.1f us' % (R, K, fwd, fl / fwd / 1e6, dgrad, wgrad), flush=True)
    # K = 610 data inside a 612-pitch buffer (what a padded feature row would look like without padding the weights)
    xb = torch.randn(R, 612, device='cuda')
    xv = xb[:, :610]
    w = torch.randn(384, 610, device='cuda')
    print('rows %6d K 610 in a 612 pitch: fwd %7.1f us' % (R, t(lambda: torch.matmul(xv, w.t()))), flush=True)
