import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.common.arguments import make_args
from marl_dmfb_amd.env.meda import VecMEDA
from marl_dmfb_amd.train import Trainer
E = 4096
env = VecMEDA(30, 30, 4, fov=19, n_envs=E, seed=1, device='cuda:0', version=2)
args = make_args(name='meda', drop_num=4, width=30, length=30, fov=19, device='cuda:0', n_envs=E, batch_size=256, train_time=2, buffer_size=8192, **env.get_env_info())
tr = Trainer(env, args)
w, buf = tr.rolloutWorker, tr.buffer
w.use_graph = False
terms = []
w.stream_step_hook = lambda s, a, term: terms.append(int(term.sum().item()))
from marl_dmfb_amd import _lib
lib = w._ops()
orig = lib.rollout_stream_step
evs = []
def timed(*a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = orig(*a); e1.record(); evs.append((e0, e1)); return rc
lib.rollout_stream_step = timed
w.generate_steps(buf, 130)
torch.cuda.synchronize()
us = [round(a.elapsed_time(b) * 1e3) for a, b in evs]
print('terminated per step:', terms)
print('stream_step us:', us)
