"""Generate the degradation-sweep harness golden by RUNNING THE REFERENCE (container-only).

  tests/golden/degre_harness_10x10_4d.npz  the reference's Degre_evaluator.evaluate_process (evaDegre.py:8-26) ->
        Evaluator.evaluate / _generate_episode (common/rollout.py:41-85) on ONE ageing chip (b_degrade, per_degrade 1.0,
        usage pre-loaded close to the threshold 50), greedy policy with det_init weights, tasks and move draws injected:
        per-epoch rewards / steps / success and the (epochs, W, L) health snapshots.

Run: python tools/oracle/gen_degre_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
from env.DMFB.dmfb import DMFBenv  # noqa: E402
import evaDegre as ref_evaDegre  # noqa: E402
from gen_vdn_golden import det_init, ref_args  # noqa: E402  (imports install the same shim; __main__ guard keeps it inert)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
QUEUE = ref_shim.DrawQueue()
ref_shim.patch_random(QUEUE)


def main(seed=4, W=10, L=10, n=4, fov=9, epochs=4, tasks=3):
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    env = DMFBenv(W, L, n, 0, fov=fov, stall=True, b_degrade=True, per_degrade=1.0)
    rm = env.routing_manager
    degrade = rm.m_degrade.copy()
    usage0 = rng.integers(40, 51, (W, L)).astype(np.float64)
    rm.m_usage = usage0.copy()
    args = ref_args(n, W, L, fov, env)
    args.evaluate_epoch, args.evaluate_task = epochs, tasks
    ev = ref_evaDegre.Degre_evaluator(env, args)
    det_init(ev.agents.policy.eval_rnn, salt=0.25)
    ref_evaDegre.env = env  # evaluate_process reads the module-level name `env` (evaDegre.py:18)

    # ---- injection: tasks through _Generate_Start_End, move draws through random.random
    task_log, draw_log, len_log = [], [], []

    def gen_points():
        while True:
            pts = np.stack([rng.integers(0, W, 2 * n), rng.integers(0, L, 2 * n)], axis=1)
            d = pts[:, None, :] - pts[None, :, :]
            if ((d ** 2).sum(-1) + np.eye(2 * n, dtype=int) * 99).min() > 2:
                k = len(task_log)
                if k // tasks == 1 or k == 3 * tasks:        # epoch 1 (and the first task of epoch 3: a mixed epoch, success 1/3): every droplet starts on its goal -> each episode SUCCEEDS after one step
                    pts[n:] = pts[:n]      # (success = 1, steps < limit: the early-exit branches of rollout.py:41-67 inside the ageing loop)
                elif k % tasks == 1:       # mixed epochs: three on their goals, one a single cell away (the greedy net decides)
                    pts[n:] = pts[:n]
                    pts[n, 0] = pts[0, 0] + (1 if pts[0, 0] < W - 1 else -1)
                task_log.append(pts.copy())
                draw_log.append([])
                return pts
    rm._Generate_Start_End = gen_points
    real_step = env.step

    def step(actions, record=True):
        drawing = [not (rm.stall and rm.distances[i] == 0) for i in range(n)]
        u = rng.random(n)
        QUEUE.feed([u[i] for i in range(n) if drawing[i]])
        out = real_step(actions, record)
        assert not QUEUE.q
        draw_log[-1].append(u)
        return out
    env.step = step
    gaps = []
    orig = ev.agents.policy.eval_rnn.forward

    def fwd(inputs, hidden):
        q, h = orig(inputs, hidden)
        s = torch.sort(q.detach().reshape(-1), descending=True).values
        gaps.append(float(s[0] - s[1]))
        return q, h
    ev.agents.policy.eval_rnn.forward = fwd

    rewards, steps, success, health = ev.evaluate_process()
    n_ep = len(task_log)
    assert n_ep == epochs * tasks
    T = env.max_step
    uni = np.full((n_ep, T, n), np.nan)
    for k, d in enumerate(draw_log):
        uni[k, :len(d)] = np.array(d)
        len_log.append(len(d))
    out = {'cfg': np.array([W, L, n, fov, epochs, tasks]), 'degrade': degrade, 'usage0': usage0,
           'starts': np.array([p[:n] for p in task_log], np.int32), 'ends': np.array([p[n:] for p in task_log], np.int32),
           'uniforms': uni, 'ep_len': np.array(len_log, np.int32),
           'rewards': np.array(rewards, np.float64), 'steps': np.array(steps, np.float64), 'success': np.array(success, np.float64),
           'health': np.asarray(health, np.float64), 'health_end': rm.m_health.copy(), 'usage_end': rm.m_usage.copy(),
           'min_gap': np.array(min(gaps))}
    path = os.path.join(OUT, 'degre_harness_%dx%d_%dd.npz' % (W, L, n))
    np.savez_compressed(path, **out)
    print(os.path.basename(path), 'episodes=%d lens=%s rewards=%s steps=%s success=%s mean_health=%s min_gap=%.2e bytes=%d' % (
        n_ep, len_log, np.round(rewards, 3), steps, success, np.round(np.asarray(health).mean(axis=(1, 2)), 4), min(gaps), os.path.getsize(path)))


if __name__ == '__main__':
    main()
