"""Generate DMFB golden vectors by RUNNING THE REFERENCE (container-only).

Writes tests/golden/dmfb_<name>.npz.  Each file holds episodes replayed on the real
`DMFBenv` (env/DMFB/dmfb.py) with
  * tasks injected through routing_manager.starts/ends + restart()  (dmfb.py:185-190, 599-605)
  * every `random.random()` draw injected through a queue          (dmfb.py:335)
  * health / usage / degrade maps set directly where used           (dmfb.py:147-151)
and records, per step, the inputs (actions, per-agent draws) and everything the env
returns (obs int8, per-agent float64 rewards, dones, constraints, success) plus the
post-step positions and maps.  tests/test_oracle_dmfb_golden.py replays them through the
C oracle; the GPU parity tests replay them through the HIP kernels.

Run:  python tools/oracle/gen_dmfb_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
from env.DMFB.dmfb import DMFBenv, Block  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
QUEUE = ref_shim.DrawQueue()
ref_shim.patch_random(QUEUE)


def random_task(rng, W, L, n, spaced):
    """2n points; `spaced` enforces the reference generator's d^2>2 rule, otherwise only
    distinct start cells (adjacent starts, goals on top of other goals/starts allowed)."""
    while True:
        pts = np.stack([rng.integers(0, W, 2 * n), rng.integers(0, L, 2 * n)], axis=1)
        d = pts[:, None, :] - pts[None, :, :]
        d2 = (d ** 2).sum(-1) + np.eye(2 * n, dtype=int) * 1000
        if spaced and d2.min() <= 2:
            continue
        if not spaced and d2[:n, :n].min() == 0:
            continue
        return pts[:n].copy(), pts[n:].copy()


def inject_task(env, starts, ends):
    rm = env.routing_manager
    rm.starts = np.array(starts, dtype=int)
    rm.ends = np.array(ends, dtype=int)
    return env.restart()


def policy(rng, env, mode):
    """mode 0: uniform random; mode 1: mostly greedy toward the goal (reaches goals, so
    finished droplets / all-done bonus / success are exercised)."""
    rm = env.routing_manager
    acts = []
    for d in rm.droplets:
        if mode == 1 and rng.random() < 0.8:
            dx, dy = d.des_x - d.x, d.des_y - d.y
            cand = []
            if dx > 0: cand.append(1)
            if dx < 0: cand.append(2)
            if dy < 0: cand.append(3)
            if dy > 0: cand.append(4)
            acts.append(int(rng.choice(cand)) if cand else 0)
        else:
            acts.append(int(rng.integers(0, 5)))
    return acts


def run_episode(rng, env, rec, mode, exact_prob):
    """Step until all dones (what common/rollout.py:46 does) recording everything."""
    rm = env.routing_manager
    n = len(env.agents)
    steps = 0
    while True:
        acts = policy(rng, env, mode)
        drawing = [not (rm.stall and rm.distances[i] == 0) for i in range(n)]
        u = np.full(n, np.nan)
        for i in range(n):
            if drawing[i]:
                if rng.random() < exact_prob:      # u == health exactly -> the droplet moves (<=)
                    u[i] = min(rm.m_health[rm.droplets[i].x][rm.droplets[i].y], np.nextafter(1.0, 0.0))
                else:
                    u[i] = rng.random()
        QUEUE.feed([u[i] for i in range(n) if drawing[i]])
        obs, rewards, dones, info = env.step(list(acts))
        assert QUEUE.consumed == sum(drawing) and not QUEUE.q
        rec['actions'].append(np.array(acts, np.int8))
        rec['uniforms'].append(u)
        rec['rewards'].append(np.array([rewards[a] for a in env.agents], np.float64))
        rec['dones'].append(np.array([dones[a] for a in env.agents], np.uint8))
        rec['constraints'].append(np.int32(info['constraints']))
        rec['success'].append(np.uint8(info['success']))
        rec['obs'].append(np.stack(obs).astype(np.int8))
        rec['pos'].append(np.array([[d.x, d.y] for d in rm.droplets], np.int16))
        steps += 1
        if all(dones[a] for a in env.agents):
            break
    return steps


def new_rec():
    return {k: [] for k in ['actions', 'uniforms', 'rewards', 'dones', 'constraints', 'success', 'obs', 'pos']}


def finish(name, cfg, rec, ep, extra=None):
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update({k: np.stack(v) for k, v in ep.items()})
    out['cfg'] = np.array([cfg['W'], cfg['L'], cfg['n'], cfg['fov'], int(cfg['stall']), int(cfg['b_degrade'])], np.int32)
    if extra:
        out.update(extra)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'dmfb_%s.npz' % name)
    np.savez_compressed(path, **out)
    print('%-28s episodes=%d steps=%d bytes=%d success=%d' % (
        os.path.basename(path), len(ep['ep_len']), len(rec['actions']), os.path.getsize(path), int(np.sum(out['success']))))


def gen_plain(name, W, L, n, fov, n_episodes, seed, stall=True, with_health=False, crafted=()):
    """Independent episodes on a fresh task each (health 1.0 unless with_health)."""
    rng = np.random.default_rng(seed)
    env = DMFBenv(W, L, n, 0, fov=fov, stall=stall)
    rm = env.routing_manager
    rec, ep = new_rec(), {k: [] for k in ['starts', 'ends', 'ep_len', 'obs0', 'health']}
    tasks = list(crafted)
    while len(tasks) < n_episodes:
        tasks.append(random_task(rng, W, L, n, spaced=(len(tasks) % 3 != 0)))
    for k, (s, e) in enumerate(tasks):
        if with_health:
            h = rng.random((W, L)) * 0.7 + 0.3
            h[rng.random((W, L)) < 0.3] = 1.0
        else:
            h = np.ones((W, L))
        rm.m_health = h.copy()
        rm.m_usage = np.zeros((W, L))
        obs0 = inject_task(env, s, e)
        ep['starts'].append(np.array(s, np.int16)); ep['ends'].append(np.array(e, np.int16))
        ep['obs0'].append(np.stack(obs0).astype(np.int8)); ep['health'].append(h)
        ep['ep_len'].append(np.int32(run_episode(rng, env, rec, mode=k % 2, exact_prob=0.1 if with_health else 0.0)))
    if not with_health:
        ep.pop('health')
    finish(name, dict(W=W, L=L, n=n, fov=fov, stall=stall, b_degrade=False), rec, ep)


def random_blocks(rng, W, L, nb, starts, ends):
    """nb 2x2 blocks that respect GenRandomBlocks' rules (dmfb.py:243-251): no start/end inside, no overlap."""
    pts = np.vstack((starts, ends))
    blocks = []
    while len(blocks) < nb:
        x0, y0 = int(rng.integers(0, W - 3)), int(rng.integers(0, L - 3))
        if any(x0 <= px <= x0 + 1 and y0 <= py <= y0 + 1 for px, py in pts):
            continue
        if any(not (x0 > b[1] or b[0] > x0 + 1) and not (y0 > b[3] or b[2] > y0 + 1) for b in blocks):
            continue
        blocks.append((x0, x0 + 1, y0, y0 + 1))
    return blocks


def gen_blocks(name, W, L, n, fov, nb, n_episodes, seed):
    """Episodes with obstacle blocks assigned to routing_manager.blocks (dmfb.py:138): exercises
    _isTouchingBlocks (dmfb.py:301-308,338-340) and the global-coordinates block layer of the
    observation (dmfb.py:422-426)."""
    rng = np.random.default_rng(seed)
    env = DMFBenv(W, L, n, nb, fov=fov)
    rm = env.routing_manager
    rec, ep = new_rec(), {k: [] for k in ['starts', 'ends', 'ep_len', 'obs0', 'blocks']}
    for k in range(n_episodes):
        s, e = random_task(rng, W, L, n, spaced=True)
        blocks = random_blocks(rng, W, L, nb, s, e)
        rm.blocks = [Block(*b) for b in blocks]
        rm.m_health = np.ones((W, L))
        obs0 = inject_task(env, s, e)
        ep['starts'].append(np.array(s, np.int16)); ep['ends'].append(np.array(e, np.int16))
        ep['blocks'].append(np.array(blocks, np.int16))
        ep['obs0'].append(np.stack(obs0).astype(np.int8))
        ep['ep_len'].append(np.int32(run_episode(rng, env, rec, mode=k % 2, exact_prob=0.0)))
    finish(name, dict(W=W, L=L, n=n, fov=fov, stall=True, b_degrade=False), rec, ep)


def gen_degrade_chain(name, W, L, n, fov, n_episodes, seed):
    """evaDegre.py-style chain: ONE ageing chip, b_degrade=True, per_degrade=1.0,
    reset(new=False) between episodes so updateHealth (dmfb.py:465-471) fires.  Usage is
    pre-loaded near the >50 threshold so that degradation happens within a few episodes."""
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    env = DMFBenv(W, L, n, 0, fov=fov, stall=True, b_degrade=True, per_degrade=1.0)
    rm = env.routing_manager
    rm.m_usage = rng.integers(30, 52, (W, L)).astype(np.float64)
    degrade = rm.m_degrade.copy()
    usage_init = rm.m_usage.copy()
    rec = new_rec()
    ep = {k: [] for k in ['starts', 'ends', 'ep_len', 'obs0', 'health', 'usage', 'usage_end']}
    for k in range(n_episodes):
        obs0 = env.reset()                      # reference task generator + updateHealth
        ep['starts'].append(np.array(rm.starts, np.int16)); ep['ends'].append(np.array(rm.ends, np.int16))
        ep['obs0'].append(np.stack(obs0).astype(np.int8))
        ep['health'].append(rm.m_health.copy()); ep['usage'].append(rm.m_usage.copy())
        ep['ep_len'].append(np.int32(run_episode(rng, env, rec, mode=1 if k % 4 else 0, exact_prob=0.05)))
        ep['usage_end'].append(rm.m_usage.copy())
    finish(name, dict(W=W, L=L, n=n, fov=fov, stall=True, b_degrade=True), rec, ep,
           extra=dict(degrade=degrade, usage_init=usage_init))


def crafted_A():
    """Hand-made 10x10/4-droplet tasks: head-on swap, wall pushes, goal adjacent to start,
    start == goal (finished from step 0), coinciding starts (violates the generator invariant:
    _isinvalidaction then reverts every move, dmfb.py:310-323,341-343)."""
    return [
        (np.array([[0, 0], [1, 0], [9, 9], [5, 5]]), np.array([[1, 0], [0, 0], [9, 8], [5, 5]])),
        (np.array([[0, 0], [9, 0], [0, 9], [9, 9]]), np.array([[9, 9], [0, 9], [9, 0], [0, 0]])),
        (np.array([[4, 4], [4, 5], [5, 4], [5, 5]]), np.array([[5, 5], [5, 4], [4, 5], [4, 4]])),
        (np.array([[2, 2], [2, 2], [7, 7], [0, 5]]), np.array([[2, 4], [4, 2], [7, 9], [0, 7]])),
        (np.array([[3, 3], [6, 6], [3, 6], [6, 3]]), np.array([[3, 3], [6, 6], [3, 6], [6, 3]])),
    ]


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'blocks':
        gen_blocks('F_10x10_4d_fov9_5blocks', 10, 10, 4, 9, 5, n_episodes=60, seed=401)
        gen_blocks('F_20x20_10d_fov9_12blocks', 20, 20, 10, 9, 12, n_episodes=12, seed=402)
        sys.exit(0)
    gen_plain('A_10x10_4d_fov9', 10, 10, 4, 9, n_episodes=120, seed=101, crafted=crafted_A())
    gen_plain('A_10x10_4d_fov9_health', 10, 10, 4, 9, n_episodes=60, seed=102, with_health=True)
    gen_plain('A_10x10_4d_fov9_nostall', 10, 10, 4, 9, n_episodes=40, seed=103, stall=False)
    gen_plain('A_10x10_4d_fov5', 10, 10, 4, 5, n_episodes=30, seed=104)
    gen_plain('A_12x9_3d_fov7', 12, 9, 3, 7, n_episodes=30, seed=105, with_health=True)
    gen_plain('A_10x10_4d_fov6_even', 10, 10, 4, 6, n_episodes=20, seed=107)
    gen_plain('D_50x50_10d_fov9', 50, 50, 10, 9, n_episodes=16, seed=201)
    gen_plain('E_20x20_10d_fov9_health', 20, 20, 10, 9, n_episodes=24, seed=301, with_health=True)
    gen_degrade_chain('E_20x20_10d_degrade_chain', 20, 20, 10, 9, n_episodes=24, seed=1)
    gen_blocks('F_10x10_4d_fov9_5blocks', 10, 10, 4, 9, 5, n_episodes=60, seed=401)
    gen_blocks('F_20x20_10d_fov9_12blocks', 20, 20, 10, 9, 12, n_episodes=12, seed=402)
