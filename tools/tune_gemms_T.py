"""Extends the TunableOp results file with the learn shapes of EVERY episode length T = 1..T_max of a configuration (the
bench always sees T = T_max because an untrained policy never finishes early; a training run sees them all):
   MARL_DMFB_GEMM_TUNE_TO=gpurun_out/gemm_gfx950.csv python tools/tune_gemms_T.py [dmfb|meda]
(start from a copy of marl_dmfb_amd/tuning/gemm_gfx950.csv so that the existing entries are kept)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_dmfb_amd.common.arguments import make_args  # noqa: E402
from marl_dmfb_amd.train import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'dmfb'
assert os.environ.get('MARL_DMFB_GEMM_TUNE_TO'), 'set MARL_DMFB_GEMM_TUNE_TO'
if name == 'dmfb':
    from marl_dmfb_amd.env.dmfb import VecDMFB
    env = VecDMFB(10, 10, 4, fov=9, n_envs=4096, seed=7, device='cuda:0')
    args = make_args(device='cuda:0', n_envs=4096, batch_size=512, train_time=4, buffer_size=8192, **env.get_env_info())
else:
    from marl_dmfb_amd.env.meda import VecMEDA
    env = VecMEDA(30, 30, 4, fov=19, n_envs=4096, seed=7, device='cuda:0', version=2)
    args = make_args(name='meda', drop_num=4, width=30, length=30, fov=19, device='cuda:0', n_envs=4096, batch_size=256, train_time=2,
                     buffer_size=8192, **env.get_env_info())
tr = Trainer(env, args)
out = tr.rolloutWorker.generate_episode()
tr.buffer.store_episode(out[4])
T_max = args.episode_limit
for T in range(T_max, 0, -1):
    batch = tr.buffer.sample(args.batch_size)
    batch = {k: v[:, :T] for k, v in batch.items()}
    tr.agents.policy.learn(batch, T, 1)
    torch.cuda.synchronize()
    print('T', T, 'tuned', flush=True)
