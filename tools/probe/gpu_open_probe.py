"""Which host-side call opens the GPU device nodes?  (The GPU box allows 6 processes with the GPU open.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def gpu_fds():
    out = []
    for fd in os.listdir('/proc/self/fd'):
        try:
            t = os.readlink('/proc/self/fd/' + fd)
        except OSError:
            continue
        if 'kfd' in t or '/dri/' in t:
            out.append(t)
    return sorted(set(out))


print('start', gpu_fds(), flush=True)
import torch
print('import torch', gpu_fds(), flush=True)
torch.set_num_threads(1)
print('set_num_threads', gpu_fds(), flush=True)
torch.manual_seed(3)
print('manual_seed', gpu_fds(), flush=True)
import numpy as np
from oracle.dmfb_oracle import DmfbOracle
ora = DmfbOracle(10, 10, 4, fov=9, n_envs=1, seed=1)
print('oracle', gpu_fds(), flush=True)
from marl_dmfb_amd.agent.agent import Agents
from marl_dmfb_amd.common.arguments import make_args
print('import Agents', gpu_fds(), flush=True)
a = make_args(drop_num=4, width=10, length=10, fov=9, cuda=False, device='cpu', n_actions=5, n_agents=4,
              obs_shape=(3, 9, 9, 2, 245), episode_limit=40)
agents = Agents(a)
print('Agents()', gpu_fds(), flush=True)
agents.policy.init_hidden(1)
ora.reset()
obs = ora.observe()[0]
act = agents.choose_action(obs[0], np.zeros(5), 0, np.ones(5), 0.5)
print('choose_action', gpu_fds(), flush=True)
B, T, n, O = 4, 40, 4, 245
batch = {'o': torch.zeros((B, T, n, O), dtype=torch.int8), 'o_next': torch.zeros((B, T, n, O), dtype=torch.int8),
         'u': torch.zeros((B, T, n, 1), dtype=torch.int8), 'r': torch.zeros((B, T, 1)), 'avail_u': torch.ones((B, T, n, 5), dtype=torch.int8),
         'avail_u_next': torch.ones((B, T, n, 5), dtype=torch.int8), 'u_onehot': torch.zeros((B, T, n, 5), dtype=torch.int8),
         'padded': torch.zeros((B, T, 1), dtype=torch.bool), 'terminated': torch.zeros((B, T, 1), dtype=torch.bool)}
batch['terminated'][:, 5] = True
agents.train(batch, 0)
print('train', gpu_fds(), flush=True)
