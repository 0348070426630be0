"""Same-box timing of the MEDA observation kernel:
   python tools/ab_meda_obs.py <E> <version> <variant> [<variant> ...]
variant: main (the shipped library), <name> (marl_dmfb_amd/lib/libmeda_vec_<name>.so), or ablate:<bits> -- the
diagnostic build (make -C marl_dmfb_amd/csrc meda_ablate) with phases switched off (1 layer-1 order+goals, 2 layer-2
bands, 4 layer 0, 8 copy-out, 16 zero fill; the outputs are then wrong by construction, only the time is of interest).
Use a batch large enough that the kernel outlasts the host's launch path (>= 262144 chips).  Variants are timed
interleaved, several rounds, minimum reported."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from marl_dmfb_amd import _lib  # noqa: E402
from marl_dmfb_amd.env.meda import VecMEDA  # noqa: E402

E, version = int(sys.argv[1]), int(sys.argv[2])
variants = sys.argv[3:]


def make(variant):
    name = 'meda_vec' if variant == 'main' else 'meda_vec_ablate' if variant.startswith('ablate') else 'meda_vec_' + variant
    lib = C.CDLL(os.path.join(ROOT, 'marl_dmfb_amd', 'lib', 'lib%s.so' % name))
    _lib._CACHE['meda_vec'] = lib
    env = VecMEDA(30, 30, 4, fov=19, n_envs=E, seed=3, version=version)
    env.reset()
    return env


def timeit(env, variant, iters=12):
    if variant.startswith('ablate'):
        os.environ['MEDA_ABLATE'] = variant.split(':')[1]
    for _ in range(2):
        env.observe()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        env.observe()
    b.record()
    torch.cuda.synchronize()
    os.environ.pop('MEDA_ABLATE', None)
    return a.elapsed_time(b) * 1e3 / iters


envs = {}
obs = None
for v in variants:
    envs[v] = make(v)
    if obs is not None:
        envs[v].obs = obs    # one output buffer for all (1+ GB each otherwise)
    obs = envs[v].obs
best = {v: 1e9 for v in variants}
for _ in range(5):
    for v in variants:
        best[v] = min(best[v], timeit(envs[v], v))
row = 4 * ((3 if version == 2 else 4) * 361 + 2)
for v in variants:
    print('%-12s %8.1f us  %6.0f GB/s' % (v, best[v], E * row / best[v] / 1e3), flush=True)
