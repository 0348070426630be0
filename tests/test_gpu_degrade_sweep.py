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


def test_degrade_sweep_replays_reference_harness():
    """The reference's Degre_evaluator.evaluate_process (evaDegre.py:8-26) + Evaluator.evaluate (common/rollout.py:69-85)
    on one ageing chip, captured with injected tasks and move draws (tools/oracle/gen_degre_golden.py), replayed through
    the vectorised harness: per-epoch health snapshots, steps and success bit-exact, rewards to float64 rounding."""
    import os
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.evaDegre import Degre_evaluator
    from vdn_helpers import det_init
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'degre_harness_10x10_4d.npz'))
    W, L, n, fov, epochs, tasks = [int(v) for v in g['cfg']]
    assert float(g['min_gap']) > 1e-3          # the greedy choices are not decided by float rounding
    env = VecDMFB(W, L, n, fov=fov, stall=True, b_degrade=True, per_degrade=1.0, n_envs=1, seed=9)
    env.set_map('degrade', g['degrade'][None])
    env.set_map('usage', g['usage0'][None])
    args = make_args(drop_num=n, width=W, length=L, fov=fov, device='cuda:0', evaluate_epoch=epochs, evaluate_task=tasks,
                     **env.get_env_info())
    agents = Agents(args)
    det_init(agents.policy.eval_rnn, salt=0.25)
    ev = Degre_evaluator(env, agents, args)
    ev.sync_every = 1
    episode = {'k': -1}

    def reset_fn():          # env.reset() of the reference = refresh(new=False): updateHealth + the (injected) new task
        episode['k'] += 1
        env.reset()
        env.set_task(g['starts'][episode['k']][None], g['ends'][episode['k']][None])
        return env.observe()

    def uniforms_fn(t):
        u = g['uniforms'][episode['k'], t]
        return torch.as_tensor(np.where(np.isnan(u), 2.0, u)[None], dtype=torch.float64, device='cuda:0')
    ev.reset_fn, ev.uniforms_fn = reset_fn, uniforms_fn
    rewards, steps, success, health = ev.evaluate_process()
    assert episode['k'] == epochs * tasks - 1
    np.testing.assert_array_equal(health[0].view(np.int64), g['health'].view(np.int64))       # bit-exact float64 maps
    np.testing.assert_array_equal(steps[0], g['steps'])
    np.testing.assert_array_equal(success[0], g['success'])
    np.testing.assert_allclose(rewards[0], g['rewards'], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(env.get_map('health')[0].cpu().numpy().view(np.int64), g['health_end'].view(np.int64))
    np.testing.assert_array_equal(env.get_map('usage')[0].cpu().numpy(), g['usage_end'])
