"""The reference-shaped single-chip facades (`DMFBenv`, `MEDAEnv`, `MEDAEnv_v0_2`) driven exactly the
way the reference's callers drive the real envs (common/rollout.py:32-38, evaDegre.py:18-21): list /
dict actions in, (list of 1-D obs, rewards dict, dones dict, info dict) out, checked against golden
episodes captured from the reference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_dmfb_facade_protocol_and_golden_episode():
    from marl_dmfb_amd.env.dmfb import DMFBenv
    g = np.load(os.path.join(GOLDEN, 'dmfb_A_10x10_4d_fov9_health.npz'))
    env = DMFBenv(10, 10, 4, fov=9)
    assert env.agents == ['player_0', 'player_1', 'player_2', 'player_3'] and env.max_step == 40
    assert env.get_env_info() == {'n_actions': 5, 'n_agents': 4, 'obs_shape': (3, 9, 9, 2, 245), 'episode_limit': 40}
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 4 and obs[0].dtype == np.int8 and obs[0].shape == (245,)
    s = 0
    for k in range(4):
        T = int(g['ep_len'][k])
        env.routing_manager.m_health = g['health'][k]
        env.routing_manager.set_task(g['starts'][k], g['ends'][k])
        obs = env.restart()
        np.testing.assert_array_equal(np.stack(obs), g['obs0'][k])
        np.testing.assert_array_equal(env.routing_manager.distances, np.abs(g['starts'][k].astype(int) - g['ends'][k]).sum(1))
        for t in range(T):
            # the facade draws from the Philox stream; for golden replay the draws are injected below the facade
            u = np.where(np.isnan(g['uniforms'][s]), 2.0, g['uniforms'][s])
            acts = [int(a) for a in g['actions'][s]]
            o, r, d, info = env._vec.step(np.asarray(acts, np.int32)[None], u[None])
            np.testing.assert_array_equal(o[0].cpu().numpy(), g['obs'][s])
            np.testing.assert_array_equal(r[0].cpu().numpy().view(np.int64), g['rewards'][s].view(np.int64))
            s += 1
    # protocol of step(): list and dict actions, python types of the outputs
    env.reset()
    obs, rewards, dones, info = env.step([1, 2, 3, 4])
    assert set(rewards) == set(env.agents) and isinstance(rewards['player_0'], np.float64)
    assert isinstance(dones['player_0'], bool) and set(info) == {'constraints', 'success'}
    obs2, *_ = env.step({a: 0 for a in env.agents})
    assert len(obs2) == 4
    with pytest.raises(TypeError):
        env.step((0, 0, 0, 0))          # TypeError('wrong actions') dmfb.py:568
    with pytest.raises(RuntimeError):
        env.step([0, 0, 0])             # dmfb.py:272-274
    assert env.routing_manager.m_health.shape == (10, 10)   # evaDegre.py:21


def test_dmfb_facade_constructor_guards():
    from marl_dmfb_amd.env.dmfb import DMFBenv
    with pytest.raises(AssertionError):
        DMFBenv(4, 10, 2)
    with pytest.raises(RuntimeError):
        DMFBenv(10, 10, 4, fov=11)      # 'Fov is too large' dmfb.py:139-140
    with pytest.raises(TypeError):
        DMFBenv(10, 10, 14, fov=5)      # 'Too many droplets for DMFB' dmfb.py:144-146


def test_meda_facades():
    from marl_dmfb_amd.env.meda import MEDAEnv, MEDAEnv_v0_2
    env = MEDAEnv(30, 30, 4, fov=19)
    obs = env.reset()
    assert len(obs) == 4 and obs[0].dtype == np.float64 and obs[0].shape == (1446,)
    assert env.get_env_info()['obs_shape'] == 1446 and env.get_env_info()['episode_limit'] == 60
    o, r, d, info = env.step([8, 8, 8, 8])
    assert set(info) == {'constraints', 'success'} and info['constraints'] <= 0
    v2 = MEDAEnv_v0_2(30, 30, 4, fov=19)
    assert v2.reset()[0].dtype == np.int8 and v2.reset()[0].shape == (1085,)
    with pytest.raises(RuntimeError):
        MEDAEnv(10, 10, 4)
