"""Which output of k_meda_step accounts for its WRITE_SIZE?  Launches the transition with different sets of output pointers (any may
be NULL, include/meda_vec.h) in blocks of 10; run under `rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv` and split
the k_meda_step dispatches by order (tools/probe/meda_step_writes.py --reduce <dir>)."""
import ctypes as C, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = ['all', 'no_rewards', 'no_dones', 'no_fail_team', 'no_flags', 'record_only']
if len(sys.argv) > 2 and sys.argv[1] == '--reduce':
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], '**', '*_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_meda_step' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
                rows.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    rows.sort()
    vals = [v for _, v in rows][10:]     # the warm-up block
    for k, name in enumerate(VARIANTS):
        blk = vals[k * 10:(k + 1) * 10]
        if blk:
            print('%-14s WRITE_SIZE %.1f KB per launch = %.1f B per chip' % (name, sum(blk) / len(blk), sum(blk) / len(blk) * 1024 / 65536))
    sys.exit(0)
import torch
from marl_dmfb_amd import _lib
from marl_dmfb_amd.env.meda import VecMEDA
E = 65536
env = VecMEDA(30, 30, 4, fov=19, n_envs=E, seed=0, device='cuda:0', version=2)
env.reset()
g = torch.Generator(device='cuda').manual_seed(0)
acts = [torch.randint(0, 9, (E, 4), device='cuda', generator=g, dtype=torch.int8) for _ in range(4)]
def out(**drop):
    o = _lib.MedaVecStepOut()
    C.memmove(C.byref(o), C.byref(env._out), C.sizeof(o))
    o.d_obs = None                       # step-only launch: the observation kernel is not the subject
    for k in drop:
        setattr(o, k, None)
    return o
outs = {'all': out(), 'no_rewards': out(d_rewards=1), 'no_dones': out(d_dones=1), 'no_fail_team': out(d_fail=1, d_team_reward=1),
        'no_flags': out(d_success=1, d_terminated=1),
        'record_only': out(d_rewards=1, d_dones=1, d_fail=1, d_team_reward=1, d_success=1, d_terminated=1)}
for i in range(10):
    env.step(acts[i % 4], out=outs['all'])
for name in VARIANTS:
    for i in range(10):
        env.step(acts[i % 4], out=outs[name])
torch.cuda.synchronize()
print('done')
