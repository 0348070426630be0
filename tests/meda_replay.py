"""Replay committed MEDA golden episodes (tests/golden/meda_*.npz, captured from the reference by
tools/oracle/gen_meda_golden.py) through any backend with the MedaOracle method set.  Bit-exact:
float64 rewards and info['constraints'] are compared through their bit patterns."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_files(pattern='meda_*.npz'):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


def _compare(B, g, idx, active, t):
    rewards, dones, fail, succ = B.last
    np.testing.assert_array_equal(_bits(rewards[active]), _bits(g['rewards'][idx]), err_msg='rewards t=%d' % t)
    np.testing.assert_array_equal(dones[active], g['dones'][idx], err_msg='dones t=%d' % t)
    np.testing.assert_array_equal(_bits(fail[active] + 0.0), _bits(g['fail'][idx] + 0.0), err_msg='fail t=%d' % t)
    np.testing.assert_array_equal(succ[active], g['success'][idx], err_msg='success t=%d' % t)
    st = B.get_state()
    np.testing.assert_array_equal(st['pos'][active], g['pos'][idx], err_msg='pos t=%d' % t)
    np.testing.assert_array_equal(st['status'][active], g['status'][idx], err_msg='status t=%d' % t)
    np.testing.assert_array_equal(B.observe()[active], g['obs'][idx], err_msg='obs t=%d' % t)


def replay_independent(g, make_backend):
    W, L, n, fov, has_health, chain = [int(v) for v in g['cfg'][:6]]
    version = int(g['cfg'][6]) if len(g['cfg']) > 6 else 0
    ep_len = g['ep_len'].astype(int)
    E = len(ep_len)
    B = make_backend(width=W, length=L, n_agents=n, fov=fov, n_envs=E, with_maps=bool(has_health), version=version)
    if has_health:
        B.set_map('health', g['health'])
    B.set_task(g['starts'], g['ends'])
    np.testing.assert_array_equal(B.observe(), g['obs0'])
    first = np.concatenate([[0], np.cumsum(ep_len)[:-1]])
    for t in range(ep_len.max()):
        active = np.nonzero(t < ep_len)[0]
        idx = first[active] + t
        actions = np.full((E, n), 8, np.int32)
        uniforms = np.full((E, n), 2.0)
        actions[active] = g['actions'][idx]
        u = g['uniforms'][idx]
        uniforms[active] = np.where(np.isnan(u), 2.0, u)
        B.last = B.step(actions, uniforms)
        _compare(B, g, idx, active, t)
    return int(ep_len.sum())


def replay_chain(g, make_backend):
    W, L, n, fov, has_health, chain = [int(v) for v in g['cfg'][:6]]
    B = make_backend(width=W, length=L, n_agents=n, fov=fov, n_envs=1, b_degrade=True, per_degrade=1.0)
    B.set_map('degrade', g['degrade'][None])
    B.set_map('usage', g['usage_init'][None])
    B.set_map('health', g['health_init'][None])
    s = 0
    one = np.array([0])
    for k, T in enumerate(g['ep_len'].astype(int)):
        B.reset()
        B.set_task(g['starts'][k][None], g['ends'][k][None])
        np.testing.assert_array_equal(_bits(B.get_map('health')[0]), _bits(g['health'][k]), err_msg='health ep=%d' % k)
        np.testing.assert_array_equal(B.get_map('usage')[0], g['usage'][k], err_msg='usage ep=%d' % k)
        np.testing.assert_array_equal(B.observe()[0], g['obs0'][k])
        for t in range(T):
            u = g['uniforms'][s]
            B.last = B.step(g['actions'][s][None].astype(np.int32), np.where(np.isnan(u), 2.0, u)[None])
            _compare(B, g, np.array([s]), one, t)
            s += 1
        np.testing.assert_array_equal(B.get_map('usage')[0], g['usage_end'][k], err_msg='usage_end ep=%d' % k)
    return s


def replay(path, make_backend):
    g = dict(np.load(path))
    if int(g['cfg'][5]):
        return replay_chain(g, make_backend)
    return replay_independent(g, make_backend)
