"""fov-19 eval-net front end under autograd: im2col+GEMM path (conv_impl 'gemm') vs F.conv2d (MIOpen), first call and steady
state, forward + backward over `rows` rows.   python tools/probe/conv19_learn_probe.py [rows]"""
import sys
import time
import types

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from marl_dmfb_amd.network.base_net import CRNN  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
a = types.SimpleNamespace(obs_shape=(3, 19, 19, 2, 1085), hyper_hidden_dim=32, rnn_hidden_dim=128, n_actions=9, fov=19)
torch.manual_seed(0)
obs = torch.randint(0, 6, (rows, 1085), dtype=torch.int8, device='cuda')
la = torch.zeros((rows, 9), dtype=torch.int8, device='cuda')
for impl in ('gemm', 'conv2d'):
    net = CRNN(a).cuda()
    net.conv_impl = impl

    def run():
        x = torch.cat([obs.float(), la.float()], dim=1)
        y = net.features(x)
        y.sum().backward()
        return y

    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    print('%-7s first call %.2f s, steady %.2f ms per forward+backward of %d rows' % (impl, first, e0.elapsed_time(e1) / 5, rows), flush=True)
