import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_dmfb_amd.common.arguments import make_args
from marl_dmfb_amd.env.dmfb import VecDMFB
from marl_dmfb_amd.train import Trainer

def log(*a):
    print('[%.2f]' % (time.time() - T0), *a, flush=True)

T0 = time.time()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = dict(width=10, length=10, n_agents=4, fov=9)
env = VecDMFB(n_envs=E, seed=1, device='cuda:0', **cfg)
G = int(sys.argv[3]) if len(sys.argv) > 3 else 0
args = make_args(device='cuda:0', n_envs=E, batch_size=B, train_time=1, buffer_size=4 * E, use_graph=bool(G), **env.get_env_info())
tr = Trainer(env, args)
log('built')
w = tr.rolloutWorker
for r in range(4):
    t = time.time()
    out = w.generate_episode()
    torch.cuda.synchronize()
    log('rollout', r, 'took %.3f s' % (time.time() - t), 'played', int((~out[4]['padded']).sum()))
    t = time.time()
    tr.buffer.store_episode(out[4])
    torch.cuda.synchronize()
    log('store took %.3f' % (time.time() - t))
    t = time.time()
    mb = tr.buffer.sample(B)
    torch.cuda.synchronize()
    log('sample took %.3f' % (time.time() - t))
    t = time.time()
    tr.agents.train(mb, r)
    torch.cuda.synchronize()
    log('learn took %.3f' % (time.time() - t), 'loss', float(tr.agents.policy.last_loss))
