"""Phase time stamps of the observation kernel (diagnostic build: make -C marl_dmfb_amd/csrc stamps).
Prints, per configuration, the median cycles a workgroup spends in each phase and the average number of workgroups
resident per CU.  The diagnostic build's fences forbid overlaps the real kernel has: read the SHARES, not the length."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marl_dmfb_amd import _lib  # noqa: E402

_lib._CACHE['dmfb_vec'] = C.CDLL(os.path.join(ROOT, 'marl_dmfb_amd', 'lib', 'libdmfb_vec_stamps.so'))
from marl_dmfb_amd.env.dmfb import VecDMFB  # noqa: E402

CFGS = {'A': dict(width=10, length=10, n_agents=4, fov=9), 'D': dict(width=50, length=50, n_agents=10, fov=9),
        'E': dict(width=20, length=20, n_agents=10, fov=9)}
NAMES = ['top_barrier', 'zero_fill', 'barrier', 'bands+bar', 'rows+bar', 'copy_issue(|loader)', 'final_drain']  # cycles summed over a workgroup's tiles

if __name__ == '__main__':
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    for name in (sys.argv[2] if len(sys.argv) > 2 else 'A,D').split(','):
        env = VecDMFB(n_envs=E, seed=3, **CFGS[name])
        env.reset()
        T = env.launch_shape()['observe_tile']
        wgs = (E + T - 1) // T
        buf = torch.zeros((wgs, 8), dtype=torch.int64, device='cuda')
        lib = env.lib
        lib.dmfb_vec_dbg_stamps.argtypes = [C.c_void_p, C.c_void_p]
        assert lib.dmfb_vec_dbg_stamps(env.h, C.c_void_p(buf.data_ptr())) == 0
        for _ in range(3):
            env.observe()
        torch.cuda.synchronize()
        buf.zero_()
        env.observe()
        torch.cuda.synchronize()
        st = buf.cpu().numpy().astype(np.int64)
        st = st[st.sum(axis=1) != 0]  # the persistent grid has fewer workgroups than tiles
        tiles_per_wg = wgs / st.shape[0]
        out = {'cfg': name, 'E': E, 'tile': T, 'workgroups': int(st.shape[0]), 'tiles_per_wg': round(tiles_per_wg, 2),
               'median_cycles_per_tile': {n: int(np.median(st[:, k]) / tiles_per_wg) for k, n in enumerate(NAMES)},
               'total_per_tile': int(np.median(st[:, :7].sum(axis=1)) / tiles_per_wg)}
        print(json.dumps(out), flush=True)
        env.close()
