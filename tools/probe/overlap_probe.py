"""Do a VALU-bound transition launch and an HBM-bound observation launch of two independent chip sets overlap when they
sit on two HIP streams?  Two envs of E/2 chips each: (a) one stream, (b) two streams, second one offset by one
transition.   python tools/probe/overlap_probe.py <cfg A|D|E> <E total>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from marl_dmfb_amd import _lib  # noqa: E402
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

CFGS = {'A': dict(width=10, length=10, n_agents=4, fov=9), 'D': dict(width=50, length=50, n_agents=10, fov=9),
        'E': dict(width=20, length=20, n_agents=10, fov=9, b_degrade=True, per_degrade=1.0)}
name, E = sys.argv[1], int(sys.argv[2])
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = CFGS[name]
n = cfg['n_agents']
envs = [VecDMFB(n_envs=E // parts, seed=3 + i, env_id0=i * (E // parts), **cfg) for i in range(parts)]
whole = VecDMFB(n_envs=E, seed=3, **cfg)
g = torch.Generator(device='cuda').manual_seed(0)
acts = [torch.randint(0, 5, (E // parts, n), device='cuda', generator=g, dtype=torch.int8) for _ in range(4)]
acts_w = [torch.randint(0, 5, (E, n), device='cuda', generator=g, dtype=torch.int8) for _ in range(4)]
for e in envs + [whole]:
    e.reset()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def lockstep_one_stream(i):
    for e in envs:
        e.step(acts[i % 4], autoreset=True)


def lockstep_whole(i):
    whole.step(acts_w[i % 4], autoreset=True)


def lockstep_two_streams(i):
    # part p on stream p % 2; transition p+1 starts when transition p is done, so it runs next to observation p
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(cur)
    prev = None
    for p, e in enumerate(envs):
        s = streams[p % 2]
        s.wait_event(fork)
        if prev is not None:
            s.wait_event(prev)
        with torch.cuda.stream(s):
            o = _lib.DmfbVecStepOut(e.rewards.data_ptr(), e.dones.data_ptr(), e.constraints.data_ptr(), e.success.data_ptr(), None,
                                    e.team_reward.data_ptr(), e.terminated.data_ptr())
            e.step(acts[i % 4], autoreset=True, out=o)     # transition only
            prev = torch.cuda.Event()
            prev.record(s)
            e.observe()
    for s in streams:
        cur.wait_stream(s)


def timeit(fn, iters=40):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def graphed(fn, reps=8):
    """the same launches captured into one HIP graph (no host launch cost in the timed region)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(reps):
            fn(i)
    def run(i):
        gr.replay()
    return run, reps


best = {}
for k, fn in (('whole batch, graph', lockstep_whole), ('%d parts, one stream, graph' % parts, lockstep_one_stream),
              ('%d parts, two streams, graph' % parts, lockstep_two_streams)):
    try:
        run, reps = graphed(fn)
        best[k] = min(timeit(run, 10) / reps for _ in range(4))
    except Exception as ex:  # noqa: BLE001
        print('graph capture failed for', k, ':', str(ex)[:200])
for _ in range(4):
    for k, fn in (('whole batch, one launch pair', lockstep_whole), ('%d parts, one stream' % parts, lockstep_one_stream),
                  ('%d parts, two streams' % parts, lockstep_two_streams)):
        best[k] = min(best.get(k, 1e9), timeit(fn))
for k, v in best.items():
    print('%s %d chips  %-30s %8.1f us per lock-step' % (name, E, k, v))
