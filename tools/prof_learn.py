"""Op-level GPU time of VDN.learn (or the rollout) with input shapes: python tools/prof_learn.py [learn|rollout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from marl_dmfb_amd.common.arguments import make_args
from marl_dmfb_amd.env.dmfb import VecDMFB
from marl_dmfb_amd.train import Trainer
E = 4096
mode = sys.argv[1] if len(sys.argv) > 1 else 'learn'
env = VecDMFB(n_envs=E, seed=1, device='cuda:0', width=10, length=10, n_agents=4, fov=9)
args = make_args(device='cuda:0', n_envs=E, batch_size=512, train_time=1, buffer_size=4 * E, use_graph=False, **env.get_env_info())
tr = Trainer(env, args)
out = tr.rolloutWorker.generate_episode()
tr.buffer.store_episode(out[4])
for i in range(2):
    tr.agents.train(tr.buffer.sample(512), i)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for r in range(N):
        if mode == 'rollout':
            out = tr.rolloutWorker.generate_episode()
            tr.buffer.store_episode(out[4])
        else:
            tr.agents.train(tr.buffer.sample(512), r + 2)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
