"""Where does the run-to-run spread of the FOV-gather launch (0.58-0.67 of the roofline) come from?  Same process, same env handle:
the 642 MB observation buffer is re-allocated several times (a fresh hipMalloc each time: the allocator cache is emptied), and the
kernel is timed 30 times on each allocation.   python tools/probe/observe_variance.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

E = 655360
env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=3, device='cuda:0')
env.reset()
n, O = 4, env.obs_len
keep = []
for trial in range(6):
    obs = torch.empty((E, n, O), dtype=torch.int8, device='cuda')
    for _ in range(5):
        env.observe(obs=obs)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    ev[0].record()
    for i in range(30):
        env.observe(obs=obs)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(30))
    print('allocation %d at 0x%x: back-to-back launches median %.1f us, min %.1f, max %.1f' % (trial, obs.data_ptr(), ts[15], ts[0], ts[-1]), flush=True)
    if trial % 2 == 0:
        keep.append(obs)          # hold some buffers so that the next allocation lands elsewhere
    del obs
    torch.cuda.empty_cache()
