"""Rollout book-keeping parity on the GPU: the vectorised RolloutWorker playing the golden tasks
greedily with the golden weights must produce the reference's padded episode dict
(common/rollout.py:101-150 run in this container) - observations, actions, one-hots, avail masks,
padded/terminated flags bit-exact, team reward equal to float32(reference float64) - and the same
per-episode stats including the failure-inflated step count."""
import os

import numpy as np
import pytest
import torch

from vdn_helpers import det_init

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'rollout_greedy_4d.npz')


def test_vectorised_rollout_matches_reference_episodes():
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.common.rollout import RolloutWorker
    from marl_dmfb_amd.env.dmfb import VecDMFB
    g = np.load(GOLDEN)
    assert float(g['min_gap']) > 1e-3       # greedy decisions are far from ties, so fp32 noise cannot flip them
    E = g['starts'].shape[0]
    env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=0, device='cuda:0')
    args = make_args(device='cuda:0', **env.get_env_info())
    agents = Agents(args)
    det_init(agents.policy.eval_rnn, salt=0.25)
    worker = RolloutWorker(env, agents, args)
    worker.epsilon = torch.tensor(0.0, device='cuda:0')
    worker.min_epsilon = 0.0
    worker.anneal_epsilon = 0.0
    env.set_task(g['starts'], g['ends'])
    worker.reset_fn = lambda: env.restart()
    reward, steps, cons, succ, ep = worker.generate_episode()
    for k in ('o', 'u', 'o_next', 'avail_u', 'avail_u_next', 'u_onehot'):
        np.testing.assert_array_equal(ep[k].cpu().numpy(), g[k], err_msg=k)
    np.testing.assert_array_equal(ep['padded'].cpu().numpy().astype(np.uint8), g['padded'])
    np.testing.assert_array_equal(ep['terminated'].cpu().numpy().astype(np.uint8), g['terminated'])
    np.testing.assert_array_equal(ep['r'].cpu().numpy(), g['r'].astype(np.float32))
    st = g['stats']
    np.testing.assert_array_equal(reward.cpu().numpy().view(np.int64), st[:, 0].copy().view(np.int64))
    np.testing.assert_array_equal(steps.cpu().numpy(), st[:, 1].astype(np.int64))
    np.testing.assert_array_equal(cons.cpu().numpy(), st[:, 2].astype(np.int64))
    np.testing.assert_array_equal(succ.cpu().numpy(), st[:, 3].astype(np.int64))
