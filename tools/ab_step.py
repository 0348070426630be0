"""Same-box A/B timing of the lock-step (transition [+ observation]) of several builds (tools/build_variant.sh):
   python tools/ab_step.py <E> <cfg> <variant> [<variant> ...]     ('main' = the shipped library)"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marl_dmfb_amd import _lib  # noqa: E402
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

CFGS = {'A': dict(width=10, length=10, n_agents=4, fov=9), 'D': dict(width=50, length=50, n_agents=10, fov=9),
        'E': dict(width=20, length=20, n_agents=10, fov=9, b_degrade=True, per_degrade=1.0)}


def make(variant, cfg, E):
    variant, _, opt = variant.partition(':')   # '<lib>:packed' = usage log with n packed entries per step (DMFB_VEC_LOG_STRIDE knob)
    os.environ.pop('DMFB_VEC_LOG_STRIDE', None)
    if opt == 'packed':
        os.environ['DMFB_VEC_LOG_STRIDE'] = '0'
    name = 'dmfb_vec' if variant == 'main' else 'dmfb_vec_' + variant
    _lib._CACHE['dmfb_vec'] = C.CDLL(os.path.join(ROOT, 'marl_dmfb_amd', 'lib', 'lib%s.so' % name))
    return VecDMFB(n_envs=E, seed=3, **cfg)


def timeit(fn, iters=60):
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


if __name__ == '__main__':
    E = int(sys.argv[1])
    name = sys.argv[2]
    variants = sys.argv[3:]
    cfg = CFGS[name]
    envs = {v: make(v, cfg, E) for v in variants}
    g = torch.Generator(device='cuda').manual_seed(0)
    acts = [torch.randint(0, 5, (E, cfg['n_agents']), device='cuda', generator=g, dtype=torch.int8) for _ in range(8)]
    null_out = {}
    for v, e in envs.items():
        e.reset()
        for i in range(120):   # age the chips: episodes end, usage accumulates, cells degrade
            e.step(acts[i % 8], autoreset=True)
        o = _lib.DmfbVecStepOut(e.rewards.data_ptr(), e.dones.data_ptr(), e.constraints.data_ptr(), e.success.data_ptr(), None,
                                e.team_reward.data_ptr(), e.terminated.data_ptr())
        null_out[v] = o
    best = {v: [1e9, 1e9] for v in variants}
    for _ in range(5):
        for v in variants:
            e = envs[v]
            best[v][0] = min(best[v][0], timeit(lambda i: e.step(acts[i % 8], autoreset=True)))
            best[v][1] = min(best[v][1], timeit(lambda i: e.step(acts[i % 8], autoreset=True, out=null_out[v])))
    print(json.dumps({'cfg': name, 'E': E, 'lockstep_us': {v: round(t[0], 2) for v, t in best.items()},
                      'step_only_us': {v: round(t[1], 2) for v, t in best.items()}}), flush=True)
