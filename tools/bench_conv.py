import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.network.base_net import CRNN
for od in (24, 32):
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=5, fov=9)
    net = CRNN(a).cuda()
    for rows in (16384, 81920):
        obs = torch.randint(0, 5, (rows, 245), dtype=torch.int8, device='cuda')
        with torch.no_grad():
            for _ in range(5):
                net._pixel_features_hip(obs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                net._pixel_features_hip(obs)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        flop = rows * 2 * (49 * od * 27 + 25 * od * od * 9)
        print('od', od, 'rows', rows, 'us', round(us, 1), 'TFLOP/s', round(flop / us / 1e6, 1))

# training pair
from marl_dmfb_amd.network.base_net import _ConvFront9
for od in (24, 32):
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=5, fov=9)
    net = CRNN(a).cuda()
    rows = 81920
    obs = torch.randint(0, 5, (rows, 245), dtype=torch.int8, device='cuda')
    g = torch.randn(rows, od * 25, device='cuda')
    c1, c2 = net.convs
    for _ in range(3):
        pix = _ConvFront9.apply(obs, c1.weight, c1.bias, c2.weight, c2.bias)
        pix.backward(g)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        e[0].record()
        pix = _ConvFront9.apply(obs, c1.weight, c1.bias, c2.weight, c2.bias)
        e[1].record()
        pix.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print('od', od, 'rows', rows, 'train fwd ms', round(tf / 10, 3), 'bwd ms', round(tb / 10, 3))
