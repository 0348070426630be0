"""evaDegre-style sweep on the vectorised env (SURVEY 8 f1 / BASELINE config 5 shape): output layout,
per-epoch health snapshots, and the qualitative behaviour of the reference curves in DegreData/
(health starts at 1.0 and only decays; steps are capped at episode_limit; success in [0,1])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_degrade_sweep_layout_and_ageing(tmp_path):
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.evaDegre import Degre_evaluator, save_results
    E, W, n = 6, 20, 10
    env = VecDMFB(W, W, n, fov=9, b_degrade=True, per_degrade=1.0, n_envs=E, seed=1)
    env.set_map('usage', torch.full((E, W, W), 45.0, dtype=torch.float64))     # close to the >50 threshold
    args = make_args(drop_num=10, width=W, length=W, fov=9, device='cuda:0', evaluate_epoch=4, evaluate_task=3,
                     **env.get_env_info())
    torch.manual_seed(0)
    ev = Degre_evaluator(env, Agents(args), args)
    rewards, steps, success, health = ev.evaluate_process()
    assert rewards.shape == steps.shape == success.shape == (E, 4)
    assert health.shape == (E, 4, W, W)
    assert np.all(health[:, 0] == 1.0)                         # fresh chips (DegreData/*/health.npy epoch 0)
    assert np.all(np.diff(health, axis=1) <= 0)                # electrodes only degrade
    assert health[:, -1].min() < 1.0                           # and some did
    assert np.all((health > 0) & (health <= 1.0))
    assert np.all(steps <= 2 * (W + W)) and np.all(steps >= 1)
    assert np.all((success >= 0) & (success <= 1))
    deg = env.get_map('degrade').cpu().numpy()
    assert np.all((deg >= 0.6) & (deg < 1.0))                  # per_degrade=1.0: every cell in [0.6, 1) (dmfb.py:159-163)
    path = save_results(args, rewards, steps, success, health, root=str(tmp_path))
    assert np.load(path + '/health.npy').shape == (E, 4, W, W)
