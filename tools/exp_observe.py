"""Time the observation kernel alone (development aid): python tools/exp_observe.py [E] [cfgs]"""
import os
import sys
import json
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

CFGS = {'A': dict(width=10, length=10, n_agents=4, fov=9), 'D': dict(width=50, length=50, n_agents=10, fov=9),
        'E': dict(width=20, length=20, n_agents=10, fov=9)}

if __name__ == '__main__':
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    for name in (sys.argv[2] if len(sys.argv) > 2 else 'A,D,E').split(','):
        cfg = CFGS[name]
        env = VecDMFB(n_envs=E, seed=3, **cfg)
        env.reset()
        n, fov = cfg['n_agents'], cfg['fov']
        fb = n * (3 * fov * fov + 2) + 5 * n + 8
        for _ in range(5):
            env.observe()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(60):
            env.observe()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 60
        print(json.dumps(dict(cfg=name, E=E, us=round(us, 2), TBps=round(E * fb / us / 1e6, 3), frac=round(E * fb / us / 8e6, 3),
                              tile=env.launch_shape())), flush=True)
        env.close()
