"""Data-parallel learner ON THE GPU path (two ranks sharing one GPU over gloo, gradients staged through the host): the
minibatch holds the replay buffer's device dtypes, so the HIP front end, the GRU sequence kernels and the fused TD block
are what runs.  Sharded learn with ONE flat all-reduce == one rank on the whole minibatch."""
import glob
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vdn_helpers import det_init

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'vdn_learn_4d_od24.npz')))[0]
DTYPES = {'o': torch.int8, 'u': torch.int8, 'r': torch.float32, 'o_next': torch.int8, 'avail_u': torch.int8,
          'avail_u_next': torch.int8, 'u_onehot': torch.int8, 'padded': torch.bool, 'terminated': torch.bool}


def _agents(dist):
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    args = make_args(device='cuda:0', dist=dist, n_actions=5, n_agents=4, obs_shape=(3, 9, 9, 2, 245), episode_limit=40)
    torch.manual_seed(0)
    return Agents(args)


def _batch(sl):
    g = np.load(GOLDEN)
    b = {k: torch.as_tensor(g[k][sl]).to(dt).cuda() for k, dt in DTYPES.items()}
    b['padded'][0, 25:] = True      # uneven shards in valid steps: the mask-count all-reduce matters
    b['terminated'][0, 24:] = True
    return b


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ag = _agents(dist=True)
    det_init(ag.policy.eval_rnn, salt=0.0 if rank == 0 else 3.0)      # rank 1 wrong on purpose: broadcast must fix it
    ag.policy.broadcast_parameters()
    det_init(ag.policy.target_rnn, salt=0.5)
    assert ag.policy.dist
    full = _batch(slice(0, 6))
    shard = {k: v[rank * 3:(rank + 1) * 3].contiguous() for k, v in full.items()}
    assert ag.policy._td_fused_ok(shard)
    for step in range(2):
        ag.policy.learn({k: v.clone() for k, v in shard.items()}, 40, step)
    sd = {k: v.cpu().clone() for k, v in ag.policy.eval_rnn.state_dict().items()}
    torch.save({'sd': sd, 'norm': float(ag.policy.last_grad_norm)}, os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_sharded_gpu_learn_equals_big_batch(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), 'ranks diverged on %s' % k
    ag = _agents(dist=False)
    det_init(ag.policy.eval_rnn)
    det_init(ag.policy.target_rnn, salt=0.5)
    full = _batch(slice(0, 6))
    assert ag.policy._td_fused_ok(full)
    for step in range(2):
        ag.policy.learn({k: v.clone() for k, v in full.items()}, 40, step)
    assert abs(float(ag.policy.last_grad_norm) - r0['norm']) <= 2e-4 * r0['norm']
    for k, v in ag.policy.eval_rnn.state_dict().items():
        np.testing.assert_allclose(r0['sd'][k].numpy(), v.cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
