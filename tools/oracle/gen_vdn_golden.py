"""Generate VDN / rollout golden vectors by RUNNING THE REFERENCE (container-only).

  tests/golden/vdn_learn_<tag>.npz   one fixed minibatch, deterministic initial weights ->
        loss-side quantities of two consecutive `Agents.train` calls of the reference
        (policy/vdn.py:79-132): clipped gradients, grad norms and weights, sampled.
  tests/golden/rollout_greedy_4d.npz greedy `RolloutWorker.generate_episode` of the reference
        (common/rollout.py:101-150) on injected tasks: the full padded episode dict + stats.

Weights are set by `det_init` (a closed formula, shared with the tests) so they do not have to be
stored.  Run: python tools/oracle/gen_vdn_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
from env.DMFB.dmfb import DMFBenv  # noqa: E402
from agent.agent import Agents  # noqa: E402
from common.rollout import RolloutWorker  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
QUEUE = ref_shim.DrawQueue()


def det_init(module, salt=0.0):
    """Deterministic weights: w.flat[i] = scale * sin(0.37*i + 1.7*k + salt), k = parameter index."""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            i = torch.arange(p.numel(), dtype=torch.float64)
            scale = 0.08 if p.dim() > 1 else 0.02
            p.copy_((scale * torch.sin(0.37 * i + 1.7 * k + salt)).to(torch.float32).view_as(p))


def ref_args(drop_num, W, L, fov, env):
    with open('/root/reference/data-dmfb/TrainParas/{}d.yaml'.format(drop_num)) as f:
        net, train = yaml.safe_load_all(f.read())
    a = types.SimpleNamespace(alg='vdn', net='crnn', last_action=True, reuse_network=True, cuda=False, optimizer='ADAM',
                              gamma=0.99, model_dir='/tmp/model', load_model=False, load_model_name='', ith_run=0,
                              fov=fov, width=W, length=L, drop_num=drop_num, block_num=0, stall=True)
    a.__dict__.update(net)
    a.__dict__.update(train)
    a.__dict__.update(env.get_env_info())
    return a


def sample_idx(numel, k=512):
    return np.unique(np.linspace(0, numel - 1, min(k, numel)).astype(np.int64))


def near_task(rng, W, L, n, max_gap):
    """Starts anywhere, every goal 1..max_gap cells (Manhattan) from its own start; all 2n points pairwise d^2 > 2 except a
    droplet's own (start, goal) pair -- the rule of _Generate_Start_End (dmfb.py:200-226) with that one exception."""
    while True:
        st = np.stack([rng.integers(0, W, n), rng.integers(0, L, n)], axis=1)
        en = st.copy()
        for i in range(n):
            gap = int(rng.integers(1, max_gap + 1))
            dx = int(rng.integers(0, gap + 1))
            en[i] = st[i] + np.array([dx * rng.choice([-1, 1]), (gap - dx) * rng.choice([-1, 1])])
        if en.min() < 0 or en[:, 0].max() >= W or en[:, 1].max() >= L:
            continue
        pts = np.concatenate([st, en])
        d = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1) + np.eye(2 * n, dtype=int) * 99
        for i in range(n):
            d[i, n + i] = d[n + i, i] = 99
        if d.min() > 2:
            return st, en


def gen_learn(tag, drop_num, W, L, fov, B, seed, max_gap=6, p_seek=0.7, far_every=3):
    """Episodes that END EARLY: two of three tasks are injected with every goal 1..max_gap cells from its start
    (env.reset -> env.restart() on the injected task, as gen_rollout does) and every agent plays a goal-seeking action with
    probability p_seek, the reference's own epsilon = 1 choice (np.random.choice, agent/agent.py:44-45) otherwise -- so episode
    lengths differ, most episodes carry padded steps, and `terminated` fires before the limit.  Every third task is a random
    far one played with a seek probability of 0.3 (most of those run to the episode limit: no padding, terminated at the last step).  The episode dicts are still the reference's own
    RolloutWorker.generate_episode output (padding rules of common/rollout.py:131-141).  far_every=0: near tasks only, so that the
    batch's own length (agent/agent.py:51-61) is below the episode limit -- the case where a host-side length bound
    (Agents.train(max_len=)) makes the learn run over steps that are padded in EVERY episode."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    import random
    random.seed(seed)
    rng = np.random.default_rng(seed)
    env = DMFBenv(W, L, drop_num, 0, fov=fov)
    args = ref_args(drop_num, W, L, fov, env)
    agents = Agents(args)
    det_init(agents.policy.eval_rnn)
    det_init(agents.policy.target_rnn, salt=0.5)     # different target weights: exercises both nets
    worker = RolloutWorker(env, agents, args)
    worker.epsilon = 1.0
    rm = env.routing_manager
    real_reset = env.reset
    mode = {'inject': False}
    env.reset = lambda new=False: env.restart() if mode['inject'] else real_reset(new)
    ref_choice = agents.choose_action

    def choose(obs, last_action, agent_num, avail_actions, epsilon, evaluate=False):
        a = ref_choice(obs, last_action, agent_num, avail_actions, epsilon, evaluate)   # keeps the hidden-state bookkeeping
        if rng.random() < (p_seek if mode['inject'] else 0.3):
            dx, dy = int(obs[-2]), int(obs[-1])          # goal - position per axis (dmfb.py:441-453; the zoom keeps the sign)
            opts = ([1] if dx > 0 else [2] if dx < 0 else []) + ([4] if dy > 0 else [3] if dy < 0 else [])
            a = int(rng.choice(opts)) if opts else 0
        return a
    agents.choose_action = choose
    episodes = []
    for k in range(B):
        mode['inject'] = (far_every == 0 or k % far_every != 0)
        if mode['inject']:
            st, en = near_task(rng, W, L, drop_num, max_gap)
            rm.starts, rm.ends = st.copy(), en.copy()
        episodes.append(worker.generate_episode()[4])
    batch = {k: np.concatenate([e[k] for e in episodes], axis=0) for k in episodes[0]}
    out = {k: (v.astype(np.int8) if k not in ('r',) else v.astype(np.float64)) for k, v in batch.items()}
    out['padded'] = batch['padded'].astype(np.uint8)
    out['terminated'] = batch['terminated'].astype(np.uint8)
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def clip(params, max_norm, *a, **k):
        n = orig_clip(params, max_norm, *a, **k)
        norms.append(float(n))
        return n
    torch.nn.utils.clip_grad_norm_ = clip
    names = [n for n, _ in agents.policy.eval_rnn.named_parameters()]
    for step in range(2):
        agents.train({k: v.copy() for k, v in batch.items()}, step)
        for n, p in agents.policy.eval_rnn.named_parameters():
            idx = sample_idx(p.numel())
            out['idx/%s' % n] = idx
            out['grad%d/%s' % (step, n)] = p.grad.detach().reshape(-1)[idx].numpy().copy()
            out['w%d/%s' % (step, n)] = p.detach().reshape(-1)[idx].numpy().copy()
    torch.nn.utils.clip_grad_norm_ = orig_clip
    out['grad_norm'] = np.array(norms)
    out['names'] = np.array(names)
    out['cfg'] = np.array([W, L, drop_num, fov, args.hyper_hidden_dim, args.grad_norm_clip])
    path = os.path.join(OUT, 'vdn_learn_%s.npz' % tag)
    np.savez_compressed(path, **out)
    lens = (1 - out['padded'][:, :, 0].astype(int)).sum(1)
    print(os.path.basename(path), 'B=%d valid steps per episode=%s grad_norms=%s bytes=%d' % (B, lens.tolist(), norms, os.path.getsize(path)))


def gen_rollout(seed, n_tasks):
    """Greedy reference rollouts on injected tasks (health 1.0 => no randomness at all)."""
    W = L = 10
    n, fov = 4, 9
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    env = DMFBenv(W, L, n, 0, fov=fov)
    args = ref_args(n, W, L, fov, env)
    agents = Agents(args)
    det_init(agents.policy.eval_rnn, salt=0.25)
    worker = RolloutWorker(env, agents, args)
    worker.epsilon = 0.0
    worker.min_epsilon = 0.0
    worker.anneal_epsilon = 0.0
    rm = env.routing_manager
    tasks = {}

    def fake_reset(new=False):       # generate_episode calls env.reset(): replay the injected task instead
        return env.restart()
    env.reset = fake_reset
    # record the smallest gap between the best and second-best Q over all decisions
    gaps = []
    orig = agents.policy.eval_rnn.forward

    def fwd(inputs, hidden):
        q, h = orig(inputs, hidden)
        s = torch.sort(q.detach().reshape(-1), descending=True).values
        gaps.append(float(s[0] - s[1]))
        return q, h
    agents.policy.eval_rnn.forward = fwd
    eps, stats, starts, ends = [], [], [], []
    for k in range(n_tasks):
        while True:
            pts = np.stack([rng.integers(0, W, 2 * n), rng.integers(0, L, 2 * n)], axis=1)
            d = pts[:, None, :] - pts[None, :, :]
            if ((d ** 2).sum(-1) + np.eye(2 * n, dtype=int) * 99).min() > 2:
                break
        if k % 4 == 1:      # every droplet already on its goal: the episode ends after one step (padding path)
            pts[n:] = pts[:n]
        elif k % 4 == 2:    # three on their goals, one a single cell away
            pts[n:] = pts[:n]
            pts[n, 0] = pts[0, 0] + (1 if pts[0, 0] < W - 1 else -1)
        rm.starts, rm.ends = pts[:n].copy(), pts[n:].copy()
        reward, step, cons, succ, ep = worker.generate_episode()
        eps.append(ep); stats.append([reward, step, cons, succ]); starts.append(pts[:n]); ends.append(pts[n:])
    out = {k: np.concatenate([e[k] for e in eps], axis=0) for k in eps[0]}
    for k in ('o', 'o_next', 'u', 'avail_u', 'avail_u_next', 'u_onehot'):
        out[k] = out[k].astype(np.int8)
    out['padded'] = out['padded'].astype(np.uint8)
    out['terminated'] = out['terminated'].astype(np.uint8)
    out['stats'] = np.array(stats, dtype=np.float64)
    out['starts'] = np.array(starts, np.int32)
    out['ends'] = np.array(ends, np.int32)
    out['min_gap'] = np.array(min(gaps))
    path = os.path.join(OUT, 'rollout_greedy_4d.npz')
    np.savez_compressed(path, **out)
    print(os.path.basename(path), 'episodes=%d min_q_gap=%.3e success=%d bytes=%d' % (
        n_tasks, min(gaps), int(out['stats'][:, 3].sum()), os.path.getsize(path)))


if __name__ == '__main__':
    which = sys.argv[1:] or ['small', 'b64', 'rollout', 'meda']
    if 'small' in which:
        gen_learn('4d_od24', 4, 10, 10, 9, B=6, seed=5)
        gen_learn('10d_od32', 10, 20, 20, 9, B=5, seed=6)
        gen_learn('4d_od24_short', 4, 10, 10, 9, B=6, seed=8, far_every=0)
    if 'b64' in which:
        # 64 episodes x T 40 x 4 droplets = 10 240 rows: crosses the build's split-K weight-gradient (>= 1024 rows), two-stage
        # column-sum (>= 2048 rows) and multi-block-per-workgroup conv-backward (> 2560 rows) thresholds
        gen_learn('4d_od24_b64', 4, 10, 10, 9, B=64, seed=7)
    if 'rollout' in which:
        gen_rollout(seed=2, n_tasks=16)


def gen_learn_meda(tag, seed):
    """VDN.learn on the MEDA network shape (fov 19: stride-2 conv1, then the tied conv3 twice; 9 actions).  The reference's MEDA
    TRAINING path is broken (get_env_info returns an int where a tuple is indexed, meda.py:676-681), so the batch is assembled
    here from the reference's own MEDAEnv_v0_2 (observations, rewards, dones of random play) in the replay buffer's layout
    (common/replay_buffer.py:10-28) and handed to the reference's Agents.train / VDN.learn with the tuple obs_shape the network
    needs."""
    import random
    from env.MEDA.meda import MEDAEnv_v0_2
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
    W = L = 30
    n, fov, A, T = 4, 19, 9, 16
    O = 3 * fov * fov + 2
    with open('/root/reference/data-meda/TrainParas/4d.yaml') as f:
        net, train = yaml.safe_load_all(f.read())
    a = types.SimpleNamespace(alg='vdn', net='crnn', last_action=True, reuse_network=True, cuda=False, optimizer='ADAM',
                              gamma=0.99, model_dir='/tmp/model', load_model=False, load_model_name='', ith_run=0,
                              fov=fov, width=W, length=L, drop_num=n, block_num=0, stall=True)
    a.__dict__.update(net)
    a.__dict__.update(train)
    a.__dict__.update(n_actions=A, n_agents=n, obs_shape=(3, fov, fov, 2, O), episode_limit=T)
    agents = Agents(a)
    det_init(agents.policy.eval_rnn)
    det_init(agents.policy.target_rnn, salt=0.5)
    assert agents.policy.eval_rnn.convs[1] is agents.policy.eval_rnn.convs[2]     # the tied conv3
    env = MEDAEnv_v0_2(W, L, n, fov=fov)
    lens = [12, 7, 16, 9, 5]
    B = len(lens)
    batch = {'o': np.zeros((B, T, n, O)), 'u': np.zeros((B, T, n, 1)), 'r': np.zeros((B, T, 1)), 'o_next': np.zeros((B, T, n, O)),
             'avail_u': np.zeros((B, T, n, A)), 'avail_u_next': np.zeros((B, T, n, A)), 'u_onehot': np.zeros((B, T, n, A)),
             'padded': np.ones((B, T, 1)), 'terminated': np.ones((B, T, 1))}
    for b, ln in enumerate(lens):
        obs = env.reset()
        for t in range(ln):
            acts = [int(np.random.randint(0, A)) for _ in range(n)]
            nxt, rew, dones, info = env.step(acts)
            batch['o'][b, t] = np.stack(obs)
            batch['o_next'][b, t] = np.stack(nxt)
            batch['u'][b, t, :, 0] = acts
            batch['u_onehot'][b, t] = np.eye(A)[acts]
            batch['r'][b, t, 0] = np.sum([rew[k] for k in env.agents]) / n
            batch['avail_u'][b, t] = 1
            batch['avail_u_next'][b, t] = 1
            batch['padded'][b, t] = 0
            batch['terminated'][b, t] = 1.0 if t == ln - 1 else 0.0
            obs = nxt
    assert np.abs(batch['o']).max() < 127 and batch['o'].shape[-1] == O
    out = {k: (v.astype(np.int8) if k not in ('r',) else v.astype(np.float64)) for k, v in batch.items()}
    out['padded'] = batch['padded'].astype(np.uint8)
    out['terminated'] = batch['terminated'].astype(np.uint8)
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def clip(params, max_norm, *a_, **k_):
        nn_ = orig_clip(params, max_norm, *a_, **k_)
        norms.append(float(nn_))
        return nn_
    torch.nn.utils.clip_grad_norm_ = clip
    names = [nm for nm, _ in agents.policy.eval_rnn.named_parameters()]
    for step in range(2):
        agents.train({k: v.copy() for k, v in batch.items()}, step)
        for nm, p in agents.policy.eval_rnn.named_parameters():
            idx = sample_idx(p.numel())
            out['idx/%s' % nm] = idx
            out['grad%d/%s' % (step, nm)] = p.grad.detach().reshape(-1)[idx].numpy().copy()
            out['w%d/%s' % (step, nm)] = p.detach().reshape(-1)[idx].numpy().copy()
    torch.nn.utils.clip_grad_norm_ = orig_clip
    out['grad_norm'] = np.array(norms)
    out['names'] = np.array(names)
    out['cfg'] = np.array([W, L, n, fov, a.hyper_hidden_dim, a.grad_norm_clip, A, T])
    path = os.path.join(OUT, 'vdn_learn_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print(os.path.basename(path), 'B=%d grad_norms=%s bytes=%d names=%s' % (B, norms, os.path.getsize(path), names))


if __name__ == '__main__' and ('meda' in sys.argv[1:] or not sys.argv[1:]):
    gen_learn_meda('meda_4d_od32', seed=9)
