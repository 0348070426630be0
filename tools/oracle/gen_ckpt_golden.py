"""Generate checkpoint / Q-value golden vectors by RUNNING THE REFERENCE (container-only).

  tests/golden/crnn_state_<tag>.npz   state_dict key names + shapes of the reference CRNN (network/base_net.py:35-71), a
        few input rows (int8 observation values + last-action one-hot) with hidden states, and the reference's Q-values /
        next hidden states for them; weights come from det_init (closed formula shared with the tests).  Tags: 4d_od24
        (fov 9), 10d_od32 (fov 9), meda_fov19 (the tied conv2/conv3 stack of network/base_net.py:23-33), plus the greedy
        choice of the reference's Agents.choose_action (agent/agent.py:22-48) for every row.
  tests/golden/ckpt_ref/              files written by the reference's VDN.save_model (policy/vdn.py:205-218) for the
        4d_od24 network: {i}_{k}_rnn_net_params.pkl, {i}_{k}_vdn_net_params.pkl, {i}_rnn_net_params.pkl, ...

Run: python tools/oracle/gen_ckpt_golden.py
"""
import os
import shutil
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
from network.base_net import CRNN  # noqa: E402
from agent.agent import Agents  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')


def det_init(module, salt=0.0):
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            i = torch.arange(p.numel(), dtype=torch.float64)
            scale = 0.08 if p.dim() > 1 else 0.02
            p.copy_((scale * torch.sin(0.37 * i + 1.7 * k + salt)).to(torch.float32).view_as(p))


def args_for(n, fov, od, model_dir='/tmp/ref_model'):
    obs = 3 * fov * fov + 2
    return types.SimpleNamespace(alg='vdn', net='crnn', last_action=True, reuse_network=True, cuda=False, optimizer='ADAM', gamma=0.99,
                                 model_dir=model_dir, load_model=False, load_model_name='', ith_run=0, fov=fov, n_actions=5,
                                 n_agents=n, obs_shape=(3, fov, fov, 2, obs), episode_limit=40, rnn_hidden_dim=128,
                                 hyper_hidden_dim=od, lr=5e-4, grad_norm_clip=9, target_update_cycle=200)


def rows(rng, R, n, fov):
    """Observation-like int8 rows: sparse droplet ids in layers 0/1, 0/1 in layer 2, direction in [-10, 10]."""
    ff = fov * fov
    o = np.zeros((R, 3 * ff + 2), np.int8)
    for r in range(R):
        for layer in (0, 1):
            k = rng.integers(1, n + 1)
            o[r, layer * ff + rng.choice(ff, k, replace=False)] = rng.integers(1, n + 1, k)
        o[r, 2 * ff:3 * ff] = (rng.random(ff) < 0.2)
        o[r, 3 * ff:] = rng.integers(-10, 11, 2)
    onehot = np.eye(5, dtype=np.int8)[rng.integers(0, 5, R)]
    onehot[0] = 0
    return o, onehot


def gen_state(tag, n, fov, od, seed):
    rng = np.random.default_rng(seed)
    a = args_for(n, fov, od)
    net = CRNN(a)
    det_init(net, salt=0.25)
    sd = net.state_dict()
    R = 8
    o, onehot = rows(rng, R, n, fov)
    h0 = (rng.standard_normal((R, 128)) * 0.3).astype(np.float32)
    x = torch.from_numpy(np.hstack([o.astype(np.float32), onehot.astype(np.float32)]))
    with torch.no_grad():
        q, h = net(x, torch.from_numpy(h0))
        q1, h1 = net(x[:1], torch.from_numpy(h0[:1]))  # batch of one: what Agents.choose_action feeds
    # the reference's own greedy choice, one agent at a time (epsilon 0)
    agents = Agents(a)
    agents.policy.eval_rnn.load_state_dict(sd)
    chosen = []
    for r in range(R):
        agents.policy.init_hidden(1)
        agents.policy.eval_hidden[:, 0, :] = torch.from_numpy(h0[r])
        act = agents.choose_action(o[r], onehot[r].astype(np.float64), 0, [1] * 5, 0.0, evaluate=True)
        chosen.append(int(act))
    srt = np.sort(q.numpy(), axis=1)
    out = {'keys': np.array(list(sd.keys())), 'shapes': np.array([str(tuple(v.shape)) for v in sd.values()]),
           'param_names': np.array([k for k, _ in net.named_parameters()]), 'n_params': np.array(sum(p.numel() for p in net.parameters())),
           'cfg': np.array([n, fov, od]), 'obs': o, 'onehot': onehot, 'h0': h0, 'q': q.numpy(), 'h': h.numpy(),
           'q_b1': q1.numpy(), 'chosen': np.array(chosen), 'min_gap': np.array((srt[:, -1] - srt[:, -2]).min())}
    path = os.path.join(OUT, 'crnn_state_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print(os.path.basename(path), 'keys=%d params=%d min_q_gap=%.3e bytes=%d' % (len(sd), out['n_params'], out['min_gap'], os.path.getsize(path)))
    return a, sd


def gen_ckpt_files():
    """Files the reference itself writes (policy/vdn.py:205-218), for the 4-droplet network."""
    model_dir = '/tmp/ref_model_ckpt'
    shutil.rmtree(model_dir, ignore_errors=True)
    a = args_for(4, 9, 24, model_dir)
    agents = Agents(a)
    det_init(agents.policy.eval_rnn, salt=0.25)
    agents.policy.save_model(3)
    agents.policy.save_model()
    src = os.path.join(model_dir, 'vdn', 'fov9')
    dst = os.path.join(OUT, 'ckpt_ref')
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    names = sorted(os.listdir(src))
    for f in names:
        if f.startswith('0_3_') or f == '0_vdn_net_params.pkl':   # one weight file (1.2 MB) is enough; keep every NAME
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    with open(os.path.join(dst, 'NAMES.txt'), 'w') as fh:
        fh.write('\n'.join(names) + '\n')
    print('ckpt_ref:', names)


if __name__ == '__main__':
    gen_state('4d_od24', 4, 9, 24, seed=1)
    gen_state('10d_od32', 10, 9, 32, seed=2)
    gen_state('meda_fov19', 4, 19, 32, seed=3)
    gen_ckpt_files()
