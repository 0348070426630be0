"""Kernel-level timing of the fused DMFB transition kernel (env-only tier): per-launch time,
env-steps/s and algorithmic GB/s at several batch sizes.  Development aid; bench.py is the
contract benchmark."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402
from marl_dmfb_amd.env.meda import VecMEDA  # noqa: E402

CFGS = {
    'A': dict(width=10, length=10, n_agents=4, fov=9),
    'D': dict(width=50, length=50, n_agents=10, fov=9),
    'E': dict(width=20, length=20, n_agents=10, fov=9, b_degrade=True, per_degrade=1.0),
    # BASELINE config 3 is 'MEDA 10x10': rejected by the reference (meda.py:151-154); smallest legal 4-droplet chip
    # and the CLI default chip instead
    'M30': dict(width=30, length=30, n_agents=4, fov=19, meda=True),
    'M60': dict(width=30, length=60, n_agents=4, fov=19, meda=True),
    # MEDAEnv_v0_2 observation (3 int8 layers + zoomed direction): what the MEDA training path uses
    'M30v2': dict(width=30, length=30, n_agents=4, fov=19, meda=True, version=2),
}


def algo_bytes(cfg, ext_uniforms=False):
    """algorithmic bytes per env-step."""
    n, fov = cfg['n_agents'], cfg['fov']
    if cfg.get('meda'):
        return n * ((3 if cfg.get('version') == 2 else 4) * fov * fov + 2) + 8 * n + n + 9 + n + 2 * (4 * n + 16)
    writes = n * (3 * fov * fov + 2) + 8 * n + n + 5
    reads = n + (8 * n if ext_uniforms else 0) + (8 * n if cfg.get('b_degrade') else 0)
    state = 2 * (2 * n + n + 8) + (4 * n if cfg.get('b_degrade') else 0)
    return writes + reads + state


def launch_labels(name, cfg, env, E):
    """(kernel name | grid size) of the launches this configuration makes -> key in profiles/traffic.json and the
    algorithmic bytes of one launch (tools/reduce_profiles.py traffic)."""
    n, fov, W, L = cfg['n_agents'], cfg['fov'], cfg['width'], cfg['length']
    wgs = lambda tile: (E + tile - 1) // tile * 256
    sh = env.launch_shape()
    out = {}
    if cfg['meda']:
        vtag = '_v0_2' if cfg.get('version') == 2 else ''
        ob = n * ((3 if cfg.get('version') == 2 else 4) * fov * fov + 2)
        out['medak::k_meda_step<%d>|%d' % (n, wgs(sh['step_tile']))] = {'key': 'k_meda_step%s_%dx%d_%dd_E%d' % (vtag, W, L, n, E), 'algo_bytes': (algo_bytes(cfg) - ob) * E}
        out['(anonymous namespace)::k_meda_observe<%d>|%d' % (4 if n <= 4 else 8 if n <= 8 else 16, sh['observe_workgroups'] * sh['observe_block'])] = {'key': 'k_meda_observe%s_%dx%d_%dd_E%d' % (vtag, W, L, n, E), 'algo_bytes': (ob + 5 * n + 8) * E}
        return out
    ob = n * (3 * fov * fov + 2)
    maps = 'true' if cfg.get('b_degrade') else 'false'
    if E >= sh['split_min_envs']:
        out['dmfbk::k_step<%d, %s, false>|%d' % (n, maps, wgs(sh['step_only_tile']))] = {'key': 'k_step_only_%dx%d_%dd_E%d' % (W, L, n, E), 'algo_bytes': (algo_bytes(cfg) - ob) * E}
    else:
        out['dmfbk::k_step<%d, %s, true>|%d' % (n, maps, wgs(sh['fused_tile']))] = {'key': 'k_step_%dx%d_%dd_E%d' % (W, L, n, E), 'algo_bytes': algo_bytes(cfg) * E}
    out['dmfbk::k_observe<%d>|%d' % (n, sh['observe_workgroups'] * sh['observe_block'])] = {'key': 'k_observe_%dx%d_%dd_E%d' % (W, L, n, E), 'algo_bytes': (ob + 5 * n + 8) * E}
    return out


def run(name, E, iters, autoreset=True, observe=False, labels=None):
    cfg = dict(CFGS[name])
    meda = cfg.pop('meda', False)
    env = (VecMEDA if meda else VecDMFB)(n_envs=E, seed=0, **cfg)
    cfg['meda'] = meda
    env.reset()
    if labels is not None:
        for k, v in launch_labels(name, cfg, env, E).items():
            labels.setdefault(k, []).append(v)
    g = torch.Generator(device='cuda').manual_seed(0)
    acts = [torch.randint(0, 9 if meda else 5, (E, cfg['n_agents']), device='cuda', generator=g, dtype=torch.int8) for _ in range(8)]
    for i in range(10):
        env.step(acts[i % 8], autoreset=autoreset)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if observe:
        env.observe_timing(True)   # dispatch time stamps of the observation launches inside the lock-step loop
    t0.record()
    for i in range(iters):
        env.step(acts[i % 8], autoreset=autoreset)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    b = algo_bytes(cfg)
    out = dict(cfg=name, E=E, us_per_launch=round(ms * 1e3, 2), env_steps_per_s=round(E / ms * 1e3),
               algo_bytes_per_env_step=b, algo_GBps=round(E * b / ms / 1e6, 1), frac_of_8TBps=round(E * b / ms / 1e6 / 8000, 4))
    if observe:
        # the FOV-gather kernel alone: the dispatch time stamps of its launches inside the lock-step loop above (what
        # rocprofv3 --kernel-trace reports); at batches where the transition and the observation are ONE fused launch
        # there is no such launch and the standalone kernel is timed back to back instead
        us_in_loop, launches = env.observe_timing_read()
        env.observe_timing(False)
        env.observe()
        torch.cuda.synchronize()
        t0.record()
        for i in range(iters):
            env.observe()
        t1.record()
        torch.cuda.synchronize()
        b2b = t0.elapsed_time(t1) / iters * 1e3
        us = us_in_loop / launches if launches else b2b
        n, fov = cfg['n_agents'], cfg['fov']
        fb = n * (((3 if cfg.get('version') == 2 else 4) if meda else 3) * fov * fov + 2) + 5 * n + 8
        out['observe'] = dict(us_per_launch=round(us, 2), timing='dispatch time stamps in the lock-step loop' if launches else 'back-to-back launches',
                              back_to_back_us_per_launch=round(b2b, 2), algo_bytes_per_env=fb, algo_GBps=round(E * fb / us / 1e3, 1),
                              frac_of_8TBps=round(E * fb / us / 1e3 / 8000, 4))
    return out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default='A,D,E,M30,M60,M30v2')
    ap.add_argument('--sizes', default='4096,65536,262144,1048576')
    ap.add_argument('--msizes', default=None, help='batch sizes for the MEDA configurations (default: --sizes up to 65536)')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--observe', action='store_true', help='also time the observation kernel alone')
    ap.add_argument('--labels', default=None, help='write the (kernel|grid) -> traffic.json key table of this run here')
    a = ap.parse_args()
    labels = {} if a.labels else None
    for name in a.cfg.split(','):
        sizes = a.msizes if (name.startswith('M') and a.msizes) else a.sizes
        for E in [int(s) for s in sizes.split(',')]:
            if name != 'A' and E > 262144 or name.startswith("M") and E > 163840:
                continue
            print(json.dumps(run(name, E, a.iters, observe=a.observe, labels=labels)), flush=True)
    if a.labels:
        json.dump(labels, open(a.labels, 'w'), indent=1)
