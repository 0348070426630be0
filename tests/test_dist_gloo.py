"""Data-parallel learner on CPU (gloo, world_size 2): sharding a minibatch over two ranks with ONE
flat all-reduce of [gradients of the un-normalised loss, mask count] must give the same clipped
gradient and the same updated weights as one rank learning on the whole minibatch."""
import glob
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from vdn_helpers import det_init

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'vdn_learn_4d_od24.npz')))[0]
KEYS = ['o', 'u', 'r', 'o_next', 'avail_u', 'avail_u_next', 'u_onehot', 'padded', 'terminated']


def _agents(dist):
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    args = make_args(cuda=False, device='cpu', dist=dist, n_actions=5, n_agents=4, obs_shape=(3, 9, 9, 2, 245),
                     episode_limit=40)
    torch.manual_seed(0)
    ag = Agents(args)
    return ag


def _batch(sl):
    g = np.load(GOLDEN)
    b = {k: torch.as_tensor(g[k][sl]) for k in KEYS}
    b['padded'] = b['padded'].bool()
    b['terminated'] = b['terminated'].bool()
    # make the shards uneven in valid steps so that the mask-count all-reduce matters
    b['padded'][0, 25:] = True
    b['terminated'][0, 24:] = True
    return b


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    ag = _agents(dist=True)
    if rank == 0:
        det_init(ag.policy.eval_rnn)
    else:
        det_init(ag.policy.eval_rnn, salt=3.0)      # wrong on purpose: broadcast must fix it
    ag.policy.broadcast_parameters()
    det_init(ag.policy.target_rnn, salt=0.5)
    assert ag.policy.dist
    full = _batch(slice(0, 6))
    shard = {k: v[rank * 3:(rank + 1) * 3] for k, v in full.items()}
    for step in range(2):
        ag.policy.learn({k: v.clone() for k, v in shard.items()}, 40, step)
    sd = {k: v.clone() for k, v in ag.policy.eval_rnn.state_dict().items()}
    torch.save({'sd': sd, 'norm': float(ag.policy.last_grad_norm)}, os.path.join(out_dir, 'rank%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_sharded_learn_equals_big_batch(tmp_path):
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
    for step in range(2):
        ag.policy.learn({k: v.clone() for k, v in full.items()}, 40, step)
    assert abs(float(ag.policy.last_grad_norm) - r0['norm']) <= 2e-4 * r0['norm']
    for k, v in ag.policy.eval_rnn.state_dict().items():
        np.testing.assert_allclose(r0['sd'][k].numpy(), v.numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
