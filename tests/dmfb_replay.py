"""Replay committed DMFB golden episodes (tests/golden/dmfb_*.npz, captured from the
reference by tools/oracle/gen_dmfb_golden.py) through any backend with the DmfbOracle
method set (oracle.dmfb_oracle.DmfbOracle on CPU, marl_dmfb_amd.env.dmfb.VecDMFB on GPU).

All comparisons are bit-exact: float64 rewards are compared through their int64 bit
patterns, everything else is integer."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_files(pattern='dmfb_*.npz'):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def replay_independent(g, make_backend):
    """Episodes that each start from an injected task: run all of them as one batch of
    E = n_episodes lock-step envs."""
    W, L, n, fov, stall, b_degrade = [int(v) for v in g['cfg']]
    ep_len = g['ep_len'].astype(int)
    E = len(ep_len)
    has_health = 'health' in g
    nb = g['blocks'].shape[1] if 'blocks' in g else 0
    B = make_backend(width=W, length=L, n_agents=n, fov=fov, stall=bool(stall), n_envs=E, with_maps=has_health, n_blocks=nb)
    if has_health:
        B.set_map('health', g['health'])
    if nb:
        B.set_blocks(g['blocks'])
    B.set_task(g['starts'], g['ends'])
    np.testing.assert_array_equal(B.observe(), g['obs0'])
    first = np.concatenate([[0], np.cumsum(ep_len)[:-1]])
    for t in range(ep_len.max()):
        active = np.nonzero(t < ep_len)[0]
        idx = first[active] + t
        actions = np.zeros((E, n), np.int32)
        uniforms = np.full((E, n), 2.0)
        actions[active] = g['actions'][idx]
        u = g['uniforms'][idx]
        uniforms[active] = np.where(np.isnan(u), 2.0, u)
        rewards, dones, cons, succ = B.step(actions, uniforms)
        np.testing.assert_array_equal(_bits(rewards[active]), _bits(g['rewards'][idx]), err_msg='rewards t=%d' % t)
        np.testing.assert_array_equal(dones[active], g['dones'][idx], err_msg='dones t=%d' % t)
        np.testing.assert_array_equal(cons[active], g['constraints'][idx], err_msg='constraints t=%d' % t)
        np.testing.assert_array_equal(succ[active], g['success'][idx], err_msg='success t=%d' % t)
        np.testing.assert_array_equal(B.get_state()['pos'][active], g['pos'][idx], err_msg='pos t=%d' % t)
        np.testing.assert_array_equal(B.observe()[active], g['obs'][idx], err_msg='obs t=%d' % t)
    return int(ep_len.sum())


def replay_chain(g, make_backend):
    """evaDegre-style chain on ONE ageing chip: reset(new=False) between episodes
    (updateHealth), task overridden with the one the reference generated."""
    W, L, n, fov, stall, b_degrade = [int(v) for v in g['cfg']]
    B = make_backend(width=W, length=L, n_agents=n, fov=fov, stall=bool(stall), n_envs=1, b_degrade=True,
                     per_degrade=1.0)
    B.set_map('degrade', g['degrade'][None])
    B.set_map('usage', g['usage_init'][None])
    s = 0
    for k, T in enumerate(g['ep_len'].astype(int)):
        B.reset(new=False)
        B.set_task(g['starts'][k][None], g['ends'][k][None])
        np.testing.assert_array_equal(_bits(B.get_map('health')[0]), _bits(g['health'][k]), err_msg='health ep=%d' % k)
        np.testing.assert_array_equal(B.get_map('usage')[0], g['usage'][k], err_msg='usage ep=%d' % k)
        np.testing.assert_array_equal(B.observe()[0], g['obs0'][k])
        for t in range(T):
            u = g['uniforms'][s]
            rewards, dones, cons, succ = B.step(g['actions'][s][None].astype(np.int32), np.where(np.isnan(u), 2.0, u)[None])
            np.testing.assert_array_equal(_bits(rewards[0]), _bits(g['rewards'][s]), err_msg='rewards ep=%d t=%d' % (k, t))
            np.testing.assert_array_equal(dones[0], g['dones'][s])
            assert cons[0] == g['constraints'][s] and succ[0] == g['success'][s]
            np.testing.assert_array_equal(B.get_state()['pos'][0], g['pos'][s])
            np.testing.assert_array_equal(B.observe()[0], g['obs'][s], err_msg='obs ep=%d t=%d' % (k, t))
            s += 1
        np.testing.assert_array_equal(B.get_map('usage')[0], g['usage_end'][k], err_msg='usage_end ep=%d' % k)
    return s


def replay(path, make_backend):
    g = dict(np.load(path))
    if 'degrade' in g:
        return replay_chain(g, make_backend)
    return replay_independent(g, make_backend)
