import sys, torch, json
sys.path.insert(0,'/root/repo')
from marl_dmfb_amd.env.dmfb import VecDMFB
E=262144
env = VecDMFB(n_envs=E, seed=1, width=10, length=10, n_agents=4, fov=9)
env.reset()
g = torch.Generator(device='cuda').manual_seed(0)
acts = [torch.randint(0, 5, (E, 4), device='cuda', generator=g, dtype=torch.int8) for _ in range(8)]
def t(fn, iters=100):
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters): fn(i)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b)*1e3/iters,2)
for rep in range(3):
    print('observe', t(lambda i: env.observe()), 'step', t(lambda i: env.step(acts[i%8], autoreset=True)), 'observe', t(lambda i: env.observe()),
          'observe300', t(lambda i: env.observe(), 300), flush=True)
