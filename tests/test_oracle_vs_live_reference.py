"""Differential test of the CPU oracles against the LIVE reference (only where /root/reference is
mounted, i.e. in the build container; skipped on the GPU box).  Fresh random tasks, actions, draws,
health maps and blocks every run of the seed list, beyond what the committed goldens hold.

The reference is imported from where it lies through tools/oracle/ref_shim.py (inert stand-ins for
the absent gym / pettingzoo / cv2 packages); nothing is copied."""
import os
import sys

import numpy as np
import pytest

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'env')), reason='reference not mounted')
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'oracle')


@pytest.fixture(scope='module')
def ref():
    import random
    sys.path.insert(0, TOOLS)
    import ref_shim
    ref_shim.install()
    q = ref_shim.DrawQueue()
    original = random.random
    ref_shim.patch_random(q)
    from env.DMFB import dmfb as rd
    from env.MEDA import meda as rm
    yield rd, rm, q
    random.random = original


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)


@pytest.mark.parametrize('W,L,n,fov,stall,nb,seed', [
    (10, 10, 4, 9, True, 0, 1), (10, 10, 4, 9, False, 0, 2), (14, 11, 5, 7, True, 3, 3), (20, 20, 10, 9, True, 6, 4),
    (50, 50, 10, 9, True, 0, 5), (9, 13, 3, 5, True, 2, 6), (10, 10, 4, 8, True, 0, 7)])
def test_dmfb_oracle_matches_live_reference(ref, W, L, n, fov, stall, nb, seed):
    rd, _, Q = ref
    from oracle.dmfb_oracle import DmfbOracle
    rng = np.random.default_rng(seed)
    env = rd.DMFBenv(W, L, n, nb, fov=fov, stall=stall)
    rmgr = env.routing_manager
    ora = DmfbOracle(W, L, n, n_blocks=nb, fov=fov, stall=stall, n_envs=1, with_maps=True)
    for ep in range(6):
        pts = np.stack([rng.integers(0, W, 2 * n), rng.integers(0, L, 2 * n)], axis=1)
        while len({(x, y) for x, y in pts[:n]}) < n:
            pts = np.stack([rng.integers(0, W, 2 * n), rng.integers(0, L, 2 * n)], axis=1)
        blocks = []
        while len(blocks) < nb:
            x0, y0 = int(rng.integers(0, W - 3)), int(rng.integers(0, L - 3))
            if any(x0 <= px <= x0 + 1 and y0 <= py <= y0 + 1 for px, py in pts):
                continue
            if any(not (x0 > b[1] or b[0] > x0 + 1) and not (y0 > b[3] or b[2] > y0 + 1) for b in blocks):
                continue
            blocks.append((x0, x0 + 1, y0, y0 + 1))
        h = np.where(rng.random((W, L)) < 0.4, 1.0, rng.random((W, L)) * 0.8 + 0.2)
        rmgr.blocks = [rd.Block(*b) for b in blocks]
        rmgr.m_health = h.copy()
        rmgr.m_usage = np.zeros((W, L))
        rmgr.starts, rmgr.ends = pts[:n].copy(), pts[n:].copy()
        obs = env.restart()
        ora.set_map('health', h[None])
        ora.set_map('usage', np.zeros((1, W, L)))
        if nb:
            ora.set_blocks(np.array(blocks, np.int32)[None])
        ora.set_task(pts[:n][None], pts[n:][None])
        np.testing.assert_array_equal(ora.observe()[0], np.stack(obs))
        for t in range(2 * (W + L) + 3):
            acts = [int(a) for a in rng.integers(0, 5, n)]
            drawing = [not (stall and rmgr.distances[i] == 0) for i in range(n)]
            u = rng.random(n)
            Q.feed([u[i] for i in range(n) if drawing[i]])
            o, r, d, info = env.step(list(acts))
            ro, do, co, so = ora.step(np.array(acts, np.int32)[None], u[None])
            np.testing.assert_array_equal(_bits(ro[0]), _bits([r[a] for a in env.agents]))
            np.testing.assert_array_equal(do[0], [d[a] for a in env.agents])
            assert co[0] == info['constraints'] and so[0] == info['success']
            np.testing.assert_array_equal(ora.observe()[0], np.stack(o))
            np.testing.assert_array_equal(ora.get_map('usage')[0], rmgr.m_usage)
            if all(d[a] for a in env.agents) and t > W + L:
                break


@pytest.mark.parametrize('W,L,n,fov,version,seed', [
    (30, 30, 4, 19, 0, 1), (30, 60, 8, 19, 2, 2), (45, 30, 6, 9, 0, 3), (60, 75, 14, 19, 2, 4), (80, 80, 10, 19, 0, 5)])
def test_meda_oracle_matches_live_reference(ref, W, L, n, fov, version, seed):
    _, rm, Q = ref
    from oracle.meda_oracle import MedaOracle
    rng = np.random.default_rng(seed)
    cls = rm.MEDAEnv_v0_2 if version == 2 else rm.MEDAEnv
    env = cls(W, L, n, fov=fov)
    mgr = env.routing_manager
    ora = MedaOracle(W, L, n, fov=fov, n_envs=1, with_maps=True, version=version)
    for ep in range(4):
        pts = np.stack([rng.integers(2, L - 2, 2 * n), rng.integers(2, W - 2, 2 * n)], axis=1)
        h = np.where(rng.random((W, L)) < 0.5, 1.0, rng.random((W, L)) * 0.7 + 0.3)
        mgr.starts = [rm.Droplet(x - 2, x + 2, y - 2, y + 2) for x, y in pts[:n]]
        mgr.destinations = [rm.Droplet(x - 2, x + 2, y - 2, y + 2) for x, y in pts[n:]]
        env.m_health, env.m_usage, env.fails = h.copy(), np.zeros((W, L)), 0
        obs = env.restart()
        ora.set_map('health', h[None]); ora.set_map('usage', np.zeros((1, W, L)))
        ora.set_task(pts[:n][None], pts[n:][None])
        np.testing.assert_array_equal(ora.observe()[0], np.stack(obs).astype(np.int8))
        for t in range(W + L + 2):
            acts = [int(a) for a in rng.integers(0, 9, n)]
            drawing = [(not mgr.status[i]) and not (mgr.distances[i] < 4) for i in range(n)]
            u = rng.random(n)
            Q.feed([u[i] for i in range(n) if drawing[i]])
            o, r, d, info = env.step(list(acts))
            ro, do, fo, so = ora.step(np.array(acts, np.int32)[None], u[None])
            np.testing.assert_array_equal(_bits(ro[0]), _bits([r[a] for a in env.agents]))
            np.testing.assert_array_equal(do[0], [d[a] for a in env.agents])
            np.testing.assert_array_equal(_bits(fo[0] + 0.0), _bits(float(info['constraints']) + 0.0))
            assert so[0] == info['success']
            np.testing.assert_array_equal(ora.observe()[0], np.stack(o).astype(np.int8))
            np.testing.assert_array_equal(ora.get_map('usage')[0], env.m_usage)
