"""Times the one-launch GRU sequence kernels against the per-step aten path (learn shapes: T=40, R=2048)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.network.base_net import gru_sequence

T, R, H = 40, 2048, 128
cell = torch.nn.GRUCell(610, H).cuda()
ig = torch.randn(T, R, 3 * H, device='cuda', requires_grad=True)
h0 = torch.zeros(R, H, device='cuda')
g = torch.randn(T, R, H, device='cuda')
for impl in ('hip', 'aten'):
    for mode in ('fwd_nograd', 'fwd+bwd'):
        def run():
            if mode == 'fwd_nograd':
                with torch.no_grad():
                    gru_sequence(ig.detach(), h0, cell.weight_hh, cell.bias_ih, cell.bias_hh, impl)
            else:
                hs = gru_sequence(ig, h0, cell.weight_hh, cell.bias_ih, cell.bias_hh, impl)
                hs.backward(g)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        print('%s %s: %.3f ms' % (impl, mode, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
