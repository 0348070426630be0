"""Generate MEDA golden vectors by RUNNING THE REFERENCE (container-only): tests/golden/meda_*.npz.

Episodes on the real `MEDAEnv` (env/MEDA/meda.py) with tasks injected through
routing_manager.starts/destinations + restart() (meda.py:170-173,552-561), every
`random.random()` draw injected (meda.py:280), health/usage/degrade maps set directly
(meda.py:494-504).  Recorded per step: actions, per-agent draws, float64 rewards, dones,
info['constraints'] (float), info['success'], observation (float64 in the reference, all values
are small integers -> stored int8), centres, status."""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
from env.MEDA.meda import MEDAEnv, MEDAEnv_v0_2, Droplet  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
QUEUE = ref_shim.DrawQueue()
ref_shim.patch_random(QUEUE)
R = 2


def box(cx, cy):
    return Droplet(cx - R, cx + R, cy - R, cy + R)


def inject(env, starts, ends):
    rm = env.routing_manager
    rm.starts = [box(int(x), int(y)) for x, y in starts]
    rm.destinations = [box(int(x), int(y)) for x, y in ends]
    env.fails = 0
    return env.restart()


def random_task(rng, W, L, n, mode):
    """mode 0: far apart like the reference generator; mode 1: arbitrary (overlapping footprints,
    goals next to starts) to exercise punish / snap / clipped-goal smearing."""
    while True:
        pts = np.stack([rng.integers(R, L - R, 2 * n), rng.integers(R, W - R, 2 * n)], axis=1)
        if mode == 0:
            s, e = pts[:n], pts[n:]
            ds = ((s[:, None] - s[None]) ** 2).sum(-1) + np.eye(n, dtype=int) * 999
            de = ((e[:, None] - e[None]) ** 2).sum(-1) + np.eye(n, dtype=int) * 999
            if ds.min() < 81 or de.min() < 81:
                continue
        return pts[:n].copy(), pts[n:].copy()


def policy(rng, env, greedy):
    rm = env.routing_manager
    acts = []
    for d, g in zip(rm.droplets, rm.destinations):
        if rng.random() < greedy:
            dx, dy = g.x_center - d.x_center, g.y_center - d.y_center
            if abs(dx) >= 2 and abs(dy) >= 2:
                a = {(1, -1): 4, (1, 1): 5, (-1, 1): 6, (-1, -1): 7}[(int(np.sign(dx)), int(np.sign(dy)))]
            elif abs(dx) >= abs(dy):
                a = 1 if dx > 0 else (3 if dx < 0 else 8)
            else:
                a = 2 if dy > 0 else 0
            acts.append(a)
        else:
            acts.append(int(rng.integers(0, 9)))
    return acts


def run_episode(rng, env, rec, greedy, exact_prob):
    rm = env.routing_manager
    n = len(env.agents)
    steps = 0
    while True:
        acts = policy(rng, env, greedy)
        drawing = [(not rm.status[i]) and not (rm.distances[i] < 4) for i in range(n)]
        u = np.full(n, np.nan)
        for i in range(n):
            if drawing[i]:
                if rng.random() < exact_prob:
                    u[i] = min(rm.getMoveProb(rm.droplets[i], env.m_health), np.nextafter(1.0, 0.0))
                else:
                    u[i] = rng.random()
        QUEUE.feed([u[i] for i in range(n) if drawing[i]])
        obs, rewards, dones, info = env.step(list(acts))
        assert QUEUE.consumed == sum(drawing) and not QUEUE.q
        o = np.stack(obs)
        assert np.array_equal(o, np.round(o)) and np.abs(o).max() <= 127
        rec['actions'].append(np.array(acts, np.int8))
        rec['uniforms'].append(u)
        rec['rewards'].append(np.array([rewards[a] for a in env.agents], np.float64))
        rec['dones'].append(np.array([dones[a] for a in env.agents], np.uint8))
        rec['fail'].append(np.float64(info['constraints']))
        rec['success'].append(np.uint8(info['success']))
        rec['obs'].append(o.astype(np.int8))
        rec['pos'].append(np.array([[d.x_center, d.y_center] for d in rm.droplets], np.int16))
        rec['status'].append(np.array(rm.status, np.uint8))
        steps += 1
        if all(dones[a] for a in env.agents):
            break
    return steps


def gen(name, W, L, n, fov, n_episodes, seed, with_health=False, degrade_chain=False, version=0):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    env = (MEDAEnv_v0_2 if version == 2 else MEDAEnv)(W, L, n, fov=fov, b_degrade=degrade_chain, per_degrade=1.0)
    rec = {k: [] for k in ['actions', 'uniforms', 'rewards', 'dones', 'fail', 'success', 'obs', 'pos', 'status']}
    ep = {k: [] for k in ['starts', 'ends', 'ep_len', 'obs0', 'health', 'usage', 'usage_end']}
    if degrade_chain:
        env.m_usage = rng.integers(30, 52, (W, L)).astype(np.float64)
        env.m_health = rng.random((W, L)) * 0.5 + 0.5
    extra = dict(degrade=env.m_degrade.copy(), usage_init=env.m_usage.copy(), health_init=env.m_health.copy())
    for k in range(n_episodes):
        if degrade_chain:
            obs0 = env.reset()          # reference task generator + updateHealth (b_degrade)
            rm = env.routing_manager
            s = np.array([[d.x_center, d.y_center] for d in rm.starts])
            e = np.array([[d.x_center, d.y_center] for d in rm.destinations])
        else:
            s, e = random_task(rng, W, L, n, mode=k % 2)
            if with_health:
                h = rng.random((W, L)) * 0.7 + 0.3
                h[rng.random((W, L)) < 0.4] = 1.0
            else:
                h = np.ones((W, L))
            env.m_health = h.copy()
            env.m_usage = np.zeros((W, L))
            obs0 = inject(env, s, e)
        ep['starts'].append(np.array(s, np.int16)); ep['ends'].append(np.array(e, np.int16))
        ep['obs0'].append(np.stack(obs0).astype(np.int8))
        ep['health'].append(env.m_health.copy()); ep['usage'].append(env.m_usage.copy())
        ep['ep_len'].append(np.int32(run_episode(rng, env, rec, greedy=(0.9 if k % 3 else 0.3),
                                                 exact_prob=0.1 if (with_health or degrade_chain) else 0.0)))
        ep['usage_end'].append(env.m_usage.copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update({k: np.stack(v) for k, v in ep.items()})
    out.update(extra)
    out['cfg'] = np.array([W, L, n, fov, int(with_health or degrade_chain), int(degrade_chain), version], np.int32)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'meda_%s.npz' % name)
    np.savez_compressed(path, **out)
    print('%-30s episodes=%d steps=%d bytes=%d success=%d punished_steps=%d' % (
        os.path.basename(path), n_episodes, len(rec['actions']), os.path.getsize(path), int(out['success'].sum()),
        int((out['fail'] < 0).sum())))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'v02':
        gen('v02_30x30_4d_fov19', 30, 30, 4, 19, n_episodes=16, seed=21, version=2)
        gen('v02_30x60_4d_fov19', 30, 60, 4, 19, n_episodes=12, seed=22, version=2)
        gen('v02_80x80_10d_fov19', 80, 80, 10, 19, n_episodes=5, seed=23, version=2)
        gen('v02_60x75_12d_fov19', 60, 75, 12, 19, n_episodes=4, seed=24, version=2)
        sys.exit(0)
    gen('30x30_4d_fov19', 30, 30, 4, 19, n_episodes=40, seed=11)
    gen('30x60_4d_fov19_health', 30, 60, 4, 19, n_episodes=24, seed=12, with_health=True)
    gen('30x60_8d_fov19', 30, 60, 8, 19, n_episodes=10, seed=13)
    gen('80x80_10d_fov19_health', 80, 80, 10, 19, n_episodes=6, seed=14, with_health=True)
    gen('30x30_4d_fov9', 30, 30, 4, 9, n_episodes=16, seed=15)
    gen('30x30_4d_degrade_chain', 30, 30, 4, 19, n_episodes=16, seed=16, degrade_chain=True)
