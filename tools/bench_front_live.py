"""crnn_front9_forward vs crnn_front9_forward_live (all chips live / half of them) and the head-select pair: what does the live-row
indirection cost?"""
import sys, os, types, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.network.base_net import CRNN
from marl_dmfb_amd import _lib

a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=24, rnn_hidden_dim=128, n_actions=5, fov=9)
net = CRNN(a).cuda()
E, n = 4096, 4
obs = torch.randint(0, 5, (E * n, 245), dtype=torch.int8, device='cuda')
la = torch.zeros((E * n, 5), dtype=torch.int8, device='cuda'); la[:, 1] = 1
lib = _lib.rollout_ops()
vp = C.c_void_p


def timeit(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


x = torch.empty((E * n, net.padded_cols()), device='cuda')
with torch.no_grad():
    print('front9_forward full            %.1f us' % timeit(lambda: net._front_features_hip(obs, la, padded=True)))
    for frac in (1.0, 0.75, 0.5, 0.25):
        alive = (torch.rand(E, device='cuda') < frac).to(torch.uint8) if frac < 1 else torch.ones(E, dtype=torch.uint8, device='cuda')
        lst = torch.empty(E, dtype=torch.int32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
        tc = timeit(lambda: lib.rollout_compact_alive(E, vp(alive.data_ptr()), vp(lst.data_ptr()), vp(cnt.data_ptr()), None))
        t = timeit(lambda: net.front_features_live(obs, la, lst, cnt, n, x))
        print('front9_forward_live %.0f%% live   %.1f us   (compact kernel %.1f us)' % (100 * frac, t, tc))
