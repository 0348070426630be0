"""Host logic of Trainer.run (reference train.py:32-94): checkpoint/evaluation cadence, file names, the global
stop decision under data parallelism.  The rollout is replaced by a stub that hands out recorded reference
episodes (tests/golden/vdn_learn_4d_od24.npz) with a chosen step count, so the loop runs on the CPU; the real
env + rollout under Trainer.run is covered by tests/test_gpu_trainer_run.py."""
import glob
import os
import socket
import types

import numpy as np
import torch
import torch.multiprocessing as mp

from vdn_helpers import det_init

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'vdn_learn_4d_od24.npz')
KEYS = ['o', 'u', 'r', 'o_next', 'avail_u', 'avail_u_next', 'u_onehot', 'padded', 'terminated']


def _episode_batch():
    g = np.load(GOLDEN)
    b = {k: torch.as_tensor(g[k]) for k in KEYS}
    b['padded'] = b['padded'].bool()
    b['terminated'] = b['terminated'].bool()
    b['r'] = b['r'].float()
    return b


def _trainer(tmp, steps_per_round, dist=False, **over):
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.train import Trainer
    env = types.SimpleNamespace(device=torch.device('cpu'), n_envs=6, seed=0, env_id0=0, obs_len=245, max_step=40,
                                width=10, length=10)
    over.setdefault('train_time', 2)
    args = make_args(cuda=False, device='cpu', dist=dist, n_actions=5, n_agents=4, obs_shape=(3, 9, 9, 2, 245), episode_limit=40,
                     n_envs=6, batch_size=4, buffer_size=24, model_dir=os.path.join(tmp, 'model'),
                     result_dir=os.path.join(tmp, 'TrainResult'), evaluate_task=6, **over)
    torch.manual_seed(0)
    tr = Trainer(env, args)
    ep = _episode_batch()
    per_chip = torch.full((6,), steps_per_round // 6, dtype=torch.int64)
    per_chip[0] += steps_per_round - int(per_chip.sum())
    tr.rolloutWorker.generate_episode = lambda: (None, per_chip.clone(), None, None, {k: v.clone() for k, v in ep.items()})
    tr.rolloutWorker.evaluate = lambda task_num: (1.5, 33.0, 2.0, 0.25)
    return tr


def _reference_cadence(n_steps, cycle, per_round):
    """train.py:39-58 restated: (time_steps, k) of every numbered checkpoint, then the rounds played."""
    t, k, saves, rounds = 0, -1, [], 0
    while t < n_steps:
        if t // cycle > k:
            k += 1
            saves.append((t, k))
        t += per_round
        rounds += 1
    return saves, rounds, t


def test_run_cadence_and_files(tmp_path):
    tmp = str(tmp_path)
    tr = _trainer(tmp, 130, n_steps=1000, evaluate_cycle=300)
    tr.run(online_evaluate=True)
    saves, rounds, t_end = _reference_cadence(1000, 300, 130)
    assert saves == [(0, 0), (390, 1), (650, 2), (910, 3)]           # hand-checked against train.py:39-58
    assert tr.saves == saves + [(t_end, None)]
    assert tr.time_steps == t_end == 1040 and tr.trained_times == rounds * 2
    mdir = os.path.join(tmp, 'model', 'vdn', 'fov9')
    names = sorted(os.listdir(mdir))
    want = sorted(['0_%d_%s_net_params.pkl' % (k, net) for k in range(4) for net in ('rnn', 'vdn')] +
                  ['0_rnn_net_params.pkl', '0_vdn_net_params.pkl'])       # policy/vdn.py:205-218
    assert names == want
    assert torch.load(os.path.join(mdir, '0_vdn_net_params.pkl'), weights_only=True) == {}    # the mixer has no parameters
    sd = torch.load(os.path.join(mdir, '0_rnn_net_params.pkl'), weights_only=True)
    for k, v in tr.agents.policy.eval_rnn.state_dict().items():
        assert torch.equal(sd[k], v)
    rdir = os.path.join(tmp, 'TrainResult', 'vdn', 'fov9', '10by10-4d0b')
    pre = 'vdn_env(10,10,4,0,9,True)'                                    # train.py:145-158
    assert sorted(os.listdir(rdir)) == sorted(pre + n + '_0.npy' for n in ('Rewards', 'steps', 'constraints', 'success_rate', 'runtime'))
    # one evaluation per numbered checkpoint + the final one (train.py:50-57, 83-90)
    assert np.load(os.path.join(rdir, pre + 'Rewards_0.npy')).tolist() == [1.5] * 5
    assert np.load(os.path.join(rdir, pre + 'success_rate_0.npy')).tolist() == [0.25] * 5
    assert len(np.load(os.path.join(rdir, pre + 'runtime_0.npy'))) == 5


def test_run_without_online_eval_evaluates_saved_checkpoints(tmp_path, monkeypatch):
    tmp = str(tmp_path)
    tr = _trainer(tmp, 130, n_steps=500, evaluate_cycle=300)
    seen = []
    from marl_dmfb_amd.common import rollout
    monkeypatch.setattr(rollout.Evaluator, 'evaluate', lambda self, task_num: (seen.append(self.agents.args.load_model_name) or (0.5, 40.0, 1.0, 0.0)))
    tr.run(online_evaluate=False)
    assert seen == ['0_0_', '0_1_', '0_']                                 # train.py:96-118
    rdir = os.path.join(tmp, 'TrainResult', 'vdn', 'fov9', '10by10-4d0b')
    assert np.load(os.path.join(rdir, 'vdn_env(10,10,4,0,9,True)steps_0.npy')).tolist() == [40.0] * 3


def _rank(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    # uneven shards: rank 0 collects 100 env steps per round, rank 1 only 37 -- a rank-local counter would let
    # rank 0 leave the loop first and strand rank 1 in the gradient all-reduce
    tr = _trainer(tmp, 100 if rank == 0 else 37, dist=True, n_steps=600, evaluate_cycle=250)
    assert tr.dist
    det_init(tr.agents.policy.target_rnn, salt=0.5)
    tr.run(online_evaluate=True)
    sd = {k: v.clone() for k, v in tr.agents.policy.eval_rnn.state_dict().items()}
    torch.save({'sd': sd, 'time_steps': tr.time_steps, 'trained': tr.trained_times, 'saves': tr.saves}, os.path.join(tmp, 'rank%d.pt' % rank))
    torch.distributed.destroy_process_group()


def test_two_ranks_run_to_completion_with_uneven_shards(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path)
    mp.spawn(_rank, args=(2, port, tmp), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp, 'rank1.pt'))
    saves, rounds, t_end = _reference_cadence(600, 250, 137)       # the GLOBAL count drives the loop on both ranks
    assert r0['time_steps'] == r1['time_steps'] == t_end
    assert r0['trained'] == r1['trained'] == rounds * 2
    assert r0['saves'] == r1['saves'] == saves + [(t_end, None)]
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), 'ranks diverged on %s' % k
    # only rank 0 writes: one file per name, none torn by a concurrent writer
    mdir = os.path.join(tmp, 'model', 'vdn', 'fov9')
    assert len(glob.glob(os.path.join(mdir, '*rnn_net_params.pkl'))) == len(saves) + 1
    sd = torch.load(os.path.join(mdir, '0_rnn_net_params.pkl'), weights_only=True)
    for k, v in r0['sd'].items():
        assert torch.equal(sd[k], v)


def test_cli_defaults_perform_a_sane_number_of_learns():
    """python -m marl_dmfb_amd.train dmfb --drop_num=4 --fov=9: the schedule lengths are stretched so that the
    reference's horizons hold when measured in learns (common/arguments.py:vectorise_schedule)."""
    from marl_dmfb_amd.common.arguments import TRAIN_PARAS, get_train_args
    a = get_train_args(['dmfb', '--drop_num=4', '--fov=9'])
    ref = TRAIN_PARAS[('dmfb', 4)]
    mean_len = 20.0                                    # nominal episode length: half the 40-step limit
    ref_learns = 20 * 100000 / (ref['n_episodes'] * mean_len) * ref['train_time']
    rounds = a.n_steps / (a.n_envs * mean_len)
    learns = rounds * a.train_time
    assert 0.9 * ref_learns <= learns <= 1.1 * ref_learns                # ~50 000 learns, as the reference
    anneal_learns = a.anneal_steps / (a.n_envs * mean_len) * a.train_time
    ref_anneal_learns = ref['anneal_steps'] / (ref['n_episodes'] * mean_len) * ref['train_time']
    assert 0.9 * ref_anneal_learns <= anneal_learns <= 1.1 * ref_anneal_learns
    assert anneal_learns > 10 * a.target_update_cycle                    # the target net is synced many times while exploring
    assert a.batch_size >= ref['batch_size'] and a.train_time >= 1
    assert 10 <= a.n_steps // a.evaluate_cycle <= 40                     # 20 checkpoints, as n_steps=20 x evaluate_cycle=1e5
    b = get_train_args(['dmfb', '--step_scale', '1', '--batch_size', '128', '--train_time', '1'])
    assert (b.n_steps, b.anneal_steps, b.evaluate_cycle, b.batch_size, b.train_time) == (2000000, 150000, 100000, 128, 1)


def _solo_rank(_index, port, tmp, prefetch):
    """world size 1 over gloo with force_dist: the collective path (flat all-reduce, next sample drawn while it is in flight)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.distributed.init_process_group('gloo', rank=0, world_size=1)
    tr = _trainer(os.path.join(tmp, 'p%d' % prefetch), 120, dist=True, force_dist=True, n_steps=600, evaluate_cycle=10 ** 9, train_time=3,
                  prefetch_sample=bool(prefetch))
    assert tr.dist
    calls, state = [], {'in': False}
    orig, pol = tr.buffer.sample, tr.agents.policy
    orig_ar = pol.all_reduce_sum

    def ar(flat, overlap=None):
        state['in'] = True
        try:
            return orig_ar(flat, overlap)
        finally:
            state['in'] = False
    pol.all_reduce_sum = ar
    tr.buffer.sample = lambda k: (calls.append(state['in']) or orig(k))   # was the sample drawn while a collective was in flight?
    tr.agents.policy.allreduce_events = []     # CPU tensors: stays empty, must not break anything
    tr.run(online_evaluate=True)
    torch.save({'sd': tr.agents.policy.eval_rnn.state_dict(), 'calls': calls, 'trained': tr.trained_times}, os.path.join(tmp, 'solo%d.pt' % prefetch))
    torch.distributed.destroy_process_group()


def test_prefetched_sample_changes_nothing(tmp_path):
    """Data-parallel rounds draw the NEXT learn's replay sample while the gradient all-reduce is in flight (train.py:
    collect_and_learn + VDN.overlap_hook).  Same generator, same order: the weights after 5 rounds x 3 learns are bit-identical
    to those of the same loop with the prefetch switched off."""
    tmp = str(tmp_path)
    got = []
    for prefetch in (1, 0):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_solo_rank, args=(port, tmp, prefetch), nprocs=1, join=True)
        got.append(torch.load(os.path.join(tmp, 'solo%d.pt' % prefetch)))
    assert got[0]['trained'] == got[1]['trained'] == 15
    # per round with the prefetch: sample 0 is drawn by the loop (hook not armed yet), samples 1 and 2 from inside the all-reduce
    # (the one-shot hook already consumed); without: the hook is never armed
    assert got[0]['calls'] == [False, True, True] * 5 and got[1]['calls'] == [False] * 15
    for k, v in got[1]['sd'].items():
        assert torch.equal(got[0]['sd'][k], v), k


def test_host_side_length_bound_trains_the_same_weights(tmp_path):
    """Trainer hands the learns the longest episode length ever stored (kept on the host) instead of reading each sampled
    batch's own `_get_max_episode_len` back from the device (agent/agent.py:51-61 computes it on the host, from numpy).  The
    extra steps are padded in every sampled episode, their TD errors masked to zero: same weights up to the order of the
    sums (the padded terms are exact zeros)."""
    from marl_dmfb_amd.agent.agent import Agents
    sds, lens = [], []
    for bound in (True, False):
        tr = _trainer(str(tmp_path), 60, host_len_bound=bound, train_time=3)
        ep = _episode_batch()
        # make the stored episodes differ in length: cut episodes 1.. short (terminate + pad earlier), keep episode 0 long
        T_full = Agents._get_max_episode_len(None, {'terminated': ep['terminated']})
        for e in range(1, ep['terminated'].shape[0]):
            cut = max(2, T_full // 2 - e)
            ep['terminated'][e, cut - 1:] = True
            ep['padded'][e, cut:] = True
        tr.rolloutWorker.generate_episode = lambda ep=ep: (None, torch.full((6,), 10, dtype=torch.int64), None, None,
                                                             {k: v.clone() for k, v in ep.items()})
        seen = []
        learn = tr.agents.policy.learn
        tr.agents.policy.learn = lambda batch, T, *a, **k: (seen.append(T), learn(batch, T, *a, **k))[1]
        det_init(tr.agents.policy.eval_rnn, 3)
        tr.agents.policy.target_rnn.load_state_dict(tr.agents.policy.eval_rnn.state_dict())
        for _ in range(4):
            tr.collect_and_learn()
        sds.append({k: v.clone() for k, v in tr.agents.policy.eval_rnn.state_dict().items()})
        lens.append(seen)
    assert len(lens[0]) == len(lens[1]) == 12
    assert all(a >= b for a, b in zip(lens[0], lens[1])) and len(set(lens[0])) == 1   # the bound covers every sampled batch
    assert any(a > b for a, b in zip(lens[0], lens[1]))                              # ... and was really longer for some
    for k, v in sds[1].items():
        assert torch.allclose(sds[0][k], v, rtol=1e-5, atol=1e-7), k
