"""Where does the spread of the FOV-gather launch (0.58-0.67 of the roofline over the captures) come from?  Same process, same env handle:
(1) the 642 MB observation buffer is re-allocated several times (a fresh hipMalloc each time: the allocator cache is emptied) and the
kernel is timed 30 times on each allocation; (2) one allocation, shifted start offsets; (3) the FIRST buffer again at the end.
Finding (three runs): placement does not matter; the first second of a process is ~3 % slower; the level differs by ~5 % per run.
    python tools/probe/observe_variance.py"""
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

# ---- same allocation, different start offsets: is it the address phase (controllable) or the physical placement (not)?
print('--- one allocation, shifted starts')
torch.cuda.empty_cache()
big = torch.empty((E * n * O + (64 << 20),), dtype=torch.int8, device='cuda')
for off in (0, 4096, 65536, 1 << 20, 2 << 20, 16 << 20, 48 << 20):
    obs = big[off:off + E * n * O].view(E, n, O)
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
    print('offset %9d (0x%x): median %.1f us, min %.1f' % (off, obs.data_ptr(), ts[15], ts[0]), flush=True)

# ---- placement or time?  The very first buffer (kept alive) once more, now that the GPU has been busy for a few seconds
print('--- the first allocation again')
obs = keep[0]
for rep in range(3):
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
    print('allocation 0 at 0x%x again: median %.1f us, min %.1f' % (obs.data_ptr(), ts[15], ts[0]), flush=True)
