"""Trainer.run on the real HIP env + rollout (reference train.py:32-94): checkpoint cadence, file names, online evaluation,
and the HBM-resident replay ring on the device (common/replay_buffer.py:33-75)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_trainer_run_cadence_on_gpu(tmp_path):
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.train import Trainer
    E = 64
    env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=3, device='cuda:0')
    cycle = 3 * E * 40                      # a checkpoint every ~3 rounds of (failure-inflated) 64 x 40 steps
    args = make_args(device='cuda:0', n_envs=E, batch_size=32, train_time=1, buffer_size=4 * E, n_steps=3 * cycle,
                     evaluate_cycle=cycle, evaluate_task=E, model_dir=str(tmp_path / 'model'),
                     result_dir=str(tmp_path / 'TrainResult'), **env.get_env_info())
    torch.manual_seed(0)
    tr = Trainer(env, args)
    tr.run(online_evaluate=True)
    # train.py:39-58: checkpoint k when time_steps first reaches k * evaluate_cycle, then the final one
    ks = [k for _, k in tr.saves]
    assert ks[0] == 0 and ks[-1] is None and ks[:-1] == list(range(len(ks) - 1))
    for (ts, k) in tr.saves[:-1]:
        assert ts // cycle >= k and (k == 0 or ts - E * 40 < k * cycle)     # saved in the first round that crossed k * cycle
    assert tr.time_steps >= args.n_steps and tr.trained_times > 0
    mdir = tmp_path / 'model' / 'vdn' / 'fov9'
    names = sorted(os.listdir(mdir))
    want = sorted(['0_%d_%s_net_params.pkl' % (k, net) for k in ks[:-1] for net in ('rnn', 'vdn')] +
                  ['0_rnn_net_params.pkl', '0_vdn_net_params.pkl'])
    assert names == want
    sd = torch.load(mdir / '0_rnn_net_params.pkl', map_location='cpu', weights_only=True)
    for k, v in tr.agents.policy.eval_rnn.state_dict().items():
        assert torch.equal(sd[k], v.cpu())
    rdir = tmp_path / 'TrainResult' / 'vdn' / 'fov9' / '10by10-4d0b'
    pre = 'vdn_env(10,10,4,0,9,True)'
    assert sorted(os.listdir(rdir)) == sorted(pre + n + '_0.npy' for n in ('Rewards', 'steps', 'constraints', 'success_rate', 'runtime'))
    r = np.load(rdir / (pre + 'Rewards_0.npy'))
    assert len(r) == len(ks) and np.all(np.isfinite(r))               # one evaluation per checkpoint incl. the final one
    s = np.load(rdir / (pre + 'steps_0.npy'))
    assert np.all((s >= 1) & (s <= 40))


INCS = [3, 5, 4, 6, 2, 7, 1, 9, 10, 4]    # same sequence as tests/test_replay_buffer.py (wraps the 16-slot ring several times)


def test_replay_ring_on_device_matches_reference_rule():
    """The ring on `cuda`: contiguous slice fast path, wrap-around (indexed put) and the restart-at-slot-0 branch must
    leave the same episodes in the same slots as the reference's _get_storage_idx rule (common/replay_buffer.py:58-75),
    restated here on the host."""
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.common.replay_buffer import ReplayBuffer
    args = make_args(cuda=True, device='cuda:0', n_actions=5, n_agents=4, obs_shape=(3, 9, 9, 2, 245), episode_limit=6, buffer_size=16)
    buf = ReplayBuffer(args, device='cuda:0')
    assert all(v.is_cuda for v in buf.buffers.values())
    size, idx, cur = 16, 0, 0
    slots = np.full(size, -1)
    tag = 0
    for inc in INCS:
        ep = {k: torch.zeros((inc,) + tuple(v.shape[1:]), dtype=v.dtype, device='cuda:0') for k, v in buf.buffers.items()}
        tags = torch.arange(tag, tag + inc, device='cuda:0')
        ep['r'][:, 0, 0] = tags.float()
        ep['u'][:, 0, 0, 0] = (tags % 100).to(torch.int8)
        ep['padded'][:] = True
        buf.store_episode(ep)
        if idx + inc <= size:                                   # reference rule, three branches
            where = np.arange(idx, idx + inc); idx += inc
        elif idx < size:
            over = inc - (size - idx)
            where = np.concatenate([np.arange(idx, size), np.arange(0, over)]); idx = over
        else:
            where = np.arange(0, inc); idx = inc
        cur = min(size, cur + inc)
        slots[where] = np.arange(tag, tag + inc)
        tag += inc
        assert buf.current_idx == idx and buf.current_size == cur
        got = buf.buffers['r'][:, 0, 0].cpu().numpy()
        filled = slots >= 0
        np.testing.assert_array_equal(got[filled], slots[filled].astype(np.float32))
        np.testing.assert_array_equal(buf.buffers['u'][:, 0, 0, 0].cpu().numpy()[filled], (slots[filled] % 100).astype(np.int8))
    g = torch.Generator(device='cuda:0').manual_seed(1)
    buf.generator = g
    mb = buf.sample(64)
    assert all(v.is_cuda and v.shape[0] == 64 for v in mb.values())
    assert set(mb['r'][:, 0, 0].cpu().numpy().tolist()) <= set(slots.astype(np.float32).tolist())   # with replacement, from filled slots
