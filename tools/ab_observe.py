"""Same-box A/B timing of the observation kernel of several builds (tools/build_variant.sh):
   python tools/ab_observe.py <E> <cfgs> <variant> [<variant> ...]     ('main' = the shipped library)
Variants are timed interleaved, several rounds; the MIN per variant is reported (boxes and clocks vary)."""
import ctypes as C
import types
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marl_dmfb_amd import _lib  # noqa: E402
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

CFGS = {'A': dict(width=10, length=10, n_agents=4, fov=9), 'D': dict(width=50, length=50, n_agents=10, fov=9),
        'E': dict(width=20, length=20, n_agents=10, fov=9)}


def make(variant, cfg, E):
    name = 'dmfb_vec' if variant == 'main' else 'dmfb_vec_' + variant
    lib = C.CDLL(os.path.join(ROOT, 'marl_dmfb_amd', 'lib', 'lib%s.so' % name))
    _lib._CACHE['dmfb_vec'] = lib
    return VecDMFB(n_envs=E, seed=3, **cfg)


def timeit(env, iters=40):
    for _ in range(3):
        env.observe()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        env.observe()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


if __name__ == '__main__':
    E = int(sys.argv[1])
    variants = sys.argv[3:]
    for name in sys.argv[2].split(','):
        cfg = CFGS[name]
        envs = {v: make(v, cfg, E) for v in variants}
        ref = None
        for v, e in envs.items():
            e.reset()
            o = e.observe().clone()
            if ref is None:
                ref = o
            elif not torch.equal(o, ref):
                print('MISMATCH', v)
        best = {v: 1e9 for v in variants}
        for _ in range(6):
            for v in variants:
                best[v] = min(best[v], timeit(envs[v]))
        n, fov = cfg['n_agents'], cfg['fov']
        fb = n * (3 * fov * fov + 2) + 5 * n + 8
        # write-only reference on this box: torch fill_ over a buffer of the observation tensor's size
        x = torch.empty(E * n * (3 * fov * fov + 2) // 4, dtype=torch.int32, device='cuda')
        fill = min(timeit(types.SimpleNamespace(observe=lambda: x.fill_(1))) for _ in range(4))
        best['torch_fill'] = fill
        print(json.dumps({'cfg': name, 'E': E, 'us': {v: round(t, 2) for v, t in best.items()},
                          'frac': {v: round(E * fb / t / 8e6, 3) for v, t in best.items()}}), flush=True)
