"""Rollout round with and without the live-chip list (Evaluator.compact_every), random-init policy (every chip plays to the
end, so the list can only cost): ms per round, graph and eager.  `python tools/bench_rollout_live.py [graph|eager]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.common.arguments import make_args
from marl_dmfb_amd.env.dmfb import VecDMFB
from marl_dmfb_amd.train import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
E = 4096
env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=7, device='cuda:0')
args = make_args(device='cuda:0', n_envs=E, batch_size=256, train_time=1, buffer_size=2 * E, use_graph=(mode == 'graph'), **env.get_env_info())
tr = Trainer(env, args)
w = tr.rolloutWorker
for every, thr in ((0, 0.9), (4, 2.0), (1, 2.0), (0, 0.9)):
    w.compact_every, w.live_threshold = every, thr
    w._graphs = {}
    for _ in range(3):
        w.generate_episode()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        w.generate_episode()
    torch.cuda.synchronize()
    print('%s compact_every=%d: %.3f ms per round' % (mode, every, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
