"""Training driver with the loop shape of the reference's Trainer.run (train.py:32-94; the stale
runner.py:29-77 has the same shape): collect episodes -> store -> train_time x (sample, learn),
checkpoint (and optionally evaluate) every `evaluate_cycle` env steps, final save + evaluation.

One `generate_episode()` here plays one episode on every chip of the batch, so a round collects
n_envs episodes instead of the reference's n_episodes (2); the learn cadence is therefore stated
explicitly: `train_time` learns of `batch_size` episodes per round (args.train_time/batch_size).

Data parallel (`args.dist`, one rank per GPU): every rank owns its shard of chips and its replay shard.
`time_steps` is the GLOBAL count of collected env steps -- each rank's count rides in the flat gradient
all-reduce of the round's first learn -- so every rank takes the same stop/checkpoint decisions and
runs the same number of learns; only rank 0 writes checkpoints and result files."""
import copy
import os
import time

import numpy as np
import torch

from .agent.agent import Agents
from .common.replay_buffer import ReplayBuffer
from .common.rollout import Evaluator, RolloutWorker


class Trainer:
    def __init__(self, env, args):
        self.env = env
        self.args = args
        self.agents = Agents(args)
        self.rolloutWorker = RolloutWorker(env, self.agents, args)
        # the lock-step episode is replayed as ONE captured HIP graph by default on the GPU (bit-identical to the eager
        # rollout: tests/test_gpu_rollout_graph.py); --no_graph / use_graph=False plays it eagerly
        use_graph = getattr(args, 'use_graph', None)
        self.rolloutWorker.use_graph = (torch.device(env.device).type == 'cuda') if use_graph is None else bool(use_graph)
        self.buffer = ReplayBuffer(args, device=env.device)
        self.episode_rewards, self.episode_steps = [], []
        self.episode_constraints, self.success_rate, self.time_cost = [], [], []
        self.save_path = args.result_dir + '/' + args.alg + '/fov{}/{}by{}-{}d{}b'.format(
            args.fov, args.width, args.length, args.drop_num, args.block_num)
        self._time_steps = 0
        self._pending_steps = None  # data parallel: the reduced step count of the last round, still on the device
        self.trained_times = 0
        self.len_bound = 0  # longest episode the buffer currently holds (first terminated step + 1), kept on the host
        self._round_lens = []  # episode-per-round mode: longest episode of each round still in the ring buffer
        # Continuous rollout (common/rollout.py: generate_steps): a round is `round_steps` lock-steps (default episode_limit) in
        # which every chip plays all the time; episodes go into the ring on the device as they end.  Default on the GPU when the
        # fused lock-step tail applies; args.stream=False keeps one episode per chip per round (the reference's generate_episode
        # batched, with the finished chips idle until the slowest one is done).
        stream = getattr(args, 'stream', None)
        self.stream = bool(self.rolloutWorker.use_graph and self.rolloutWorker.stream_ok()) if stream is None else bool(stream)
        self.last_round = {}
        self._packed = None  # continuous mode: VDN.learn_packed applies (decided at the first learn)
        self.dist = bool(self.agents.policy.dist)
        self.rank = torch.distributed.get_rank() if self.dist else 0
        self.saves = []  # (time_steps, evaluate index or None) of every checkpoint written, for tests/logs

    def _collect_and_learn_stream(self):
        """One round in continuous mode: `round_steps` lock-steps of every chip (episodes close into the ring on the way), ONE
        device read (ring cursor / fill level / episode lengths + the round's counters), then `train_time` learns whose episodes
        are drawn on the host -- so each learn is handed its exact length (agent/agent.py:51-61) without reading anything back."""
        w, pol = self.rolloutWorker, self.agents.policy
        T = self.args.episode_limit
        K = int(getattr(self.args, 'round_steps', None) or T)
        acc = w.generate_steps(self.buffer, K)
        closed, inflated, succ, played = self.buffer.sync_host(acc)
        self.last_round = {'episodes': closed, 'success': succ, 'steps_inflated': inflated, 'played': played}
        if self.dist:
            if K < T:
                raise RuntimeError('data parallel: round_steps must be >= episode_limit (every rank must hold an episode to learn from)')
            local = torch.tensor([inflated // 4096, inflated % 4096], dtype=torch.float32, device=w.device)
            pol.ride_along, pol.ride_along_sum = local, None
        learns = self.args.train_time if self.buffer.current_size > 0 else 0
        k_batch = min(self.buffer.current_size, self.args.batch_size)
        draws = [self.buffer.draw(k_batch) for _ in range(learns)]   # host RNG: the order of the draws is fixed here
        prefetched = []
        if self._packed is None:
            want = getattr(self.args, 'packed_learn', None)
            want = (os.environ.get('MARL_DMFB_PACKED_LEARN', '1') != '0') if want is None else bool(want)   # (env var: A/B timing)
            self._packed = bool(want and hasattr(pol, 'packed_ok') and pol.packed_ok(self.buffer.buffers))
        for k in range(learns):
            idx, lens = draws[k]
            if self._packed:   # the padded steps of the drawn episodes are never computed; the ring is read in place
                # (its unit list is packed and uploaded inside, one small pageable copy per learn: measured faster than one upload of the
                # round's lists up front, which leaves the GPU idle for the packing of all of them, and much faster than a persistent
                # pinned staging buffer, which made later HIP calls of the round stall for tens of milliseconds -- tools/dbg_round.py)
                pol.learn_packed(self.buffer.buffers, idx, lens, self.trained_times)
                self.trained_times += 1
                continue
            mini_batch = prefetched.pop() if prefetched else self.buffer.gather(idx)
            if self.dist and k + 1 < learns and getattr(self.args, 'prefetch_sample', True):
                nxt = draws[k + 1][0]
                pol.overlap_hook = lambda nxt=nxt: prefetched.append(self.buffer.gather(nxt))
            self.agents.train(mini_batch, self.trained_times, max_len=int(lens[0]))
            pol.overlap_hook = None
            self.trained_times += 1
        if self.dist:
            self._finish_dist_round(pol)
        else:
            self._time_steps += inflated
        return played

    def _finish_dist_round(self, pol):
        if pol.ride_along_sum is None:  # no learn ran this round (train_time == 0): reduce the count by itself
            pol.ride_along_sum = pol.all_reduce_sum(pol.ride_along)
            pol.ride_along = None
        self._flush_steps()
        # copied to the host behind this round's learns without blocking; read when somebody asks for time_steps
        # (run() does, every round), by when the copy of a loop that does not ask (bench.py) has long finished
        src = pol.ride_along_sum
        if src.is_cuda:
            host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
            host.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending_steps = (host, ev)
        else:
            self._pending_steps = (src, None)

    def collect_and_learn(self):
        """One round of the outer loop (train.py:59-78).  Returns env steps played this round (this rank)."""
        if self.stream:
            return self._collect_and_learn_stream()
        self.rolloutWorker.stream_restart()
        _, steps, _, success, episode = self.rolloutWorker.generate_episode()
        local = steps.sum()  # failure-inflated count, as train.py:65
        # ONE device read per round: the steps played, the episode length the learns have to cover, and the step count.  Nothing
        # else in the round waits for the device (single rank), so the next round's rollout is queued while the learns still run.
        played, round_len, local_host = (int(v) for v in torch.stack([(~episode['padded']).sum(),
                                                                     Agents.first_terminated_bound(episode['terminated']), local]).tolist())
        self.rolloutWorker.note_played(played)  # decides whether the next rollout keeps finished chips out of the Q-network
        self.buffer.store_episode(episode)
        # the bound covers what the ring buffer holds NOW: a round's longest episode counts until the round has been overwritten
        self._round_lens.append(round_len)
        del self._round_lens[:-max(1, -(-self.buffer.size // max(1, self.env.n_envs)))]
        self.len_bound = max(self._round_lens)  # >= _get_max_episode_len of any batch sampled from the buffer
        max_len = self.len_bound if getattr(self.args, 'host_len_bound', True) else None
        pol = self.agents.policy
        if self.dist:  # this rank's count travels with the gradients of the first learn (two exactly representable floats)
            pol.ride_along = torch.stack([local // 4096, local % 4096]).to(torch.float32)
            pol.ride_along_sum = None
        prefetched = []
        k_batch = min(self.buffer.current_size, self.args.batch_size)
        for k in range(self.args.train_time):
            mini_batch = prefetched.pop() if prefetched else self.buffer.sample(k_batch)
            if self.dist and k + 1 < self.args.train_time and getattr(self.args, 'prefetch_sample', True):
                # the next learn's sample does not depend on this learn: it is drawn (same generator, same order) while this
                # learn's gradient all-reduce is in flight instead of after the optimizer step
                pol.overlap_hook = lambda: prefetched.append(self.buffer.sample(k_batch))
            self.agents.train(mini_batch, self.trained_times, max_len=max_len)
            pol.overlap_hook = None
            self.trained_times += 1
        if self.dist:
            self._finish_dist_round(pol)
        else:
            self._time_steps += local_host
        return played

    def _flush_steps(self):
        if self._pending_steps is not None:
            host, ev = self._pending_steps
            if ev is not None:
                ev.synchronize()
            hi, lo = (int(round(v)) for v in host.tolist())
            self._time_steps += hi * 4096 + lo
            self._pending_steps = None

    @property
    def time_steps(self):
        """Global count of collected env steps (train.py:65).  Under data parallelism the last round's reduced count is read
        from the device only here: a loop that does not look at it (bench.py) never waits for the last learn of a round."""
        self._flush_steps()
        return self._time_steps

    @time_steps.setter
    def time_steps(self, v):
        self._flush_steps()
        self._time_steps = int(v)

    def _evaluate_and_record(self, evaluator=None):
        ev = evaluator or self.rolloutWorker
        r, s, c, ok = ev.evaluate(max(1, self.args.evaluate_task // self.env.n_envs))
        self.rolloutWorker.stream_restart()  # the evaluation reset every chip: the training episodes in flight are gone
        self.episode_rewards.append(r); self.episode_steps.append(s)
        self.episode_constraints.append(c); self.success_rate.append(ok)

    def _save_model(self, k=None):
        self.agents.policy.check_td_inputs()  # on EVERY rank: a rank that raised alone would leave the others in a collective
        if self.rank == 0:
            self.agents.policy.save_model(k)
        self.saves.append((self.time_steps, k))

    def run(self, online_evaluate=False):
        """train.py:32-94: checkpoint `k` (and evaluation k) when time_steps first reaches k * evaluate_cycle; after the
        loop the final checkpoint `{i}_rnn_net_params.pkl`, one more evaluation and the result files (or, with
        online_evaluate off, evaluate_total over the saved checkpoints)."""
        evaluate_steps = -1
        start = time.time()
        while self.time_steps < self.args.n_steps:
            if self.time_steps // self.args.evaluate_cycle > evaluate_steps:
                evaluate_steps += 1
                self.time_cost.append(time.time() - start)
                self._save_model(evaluate_steps)
                if online_evaluate:
                    self._evaluate_and_record()
                    self.train_data_save()
            self.collect_and_learn()
        self._save_model()
        self.time_cost.append(time.time() - start)
        if online_evaluate:
            self._evaluate_and_record()
            self.train_data_save()
        else:
            self.evaluate_total(evaluate_steps + 1)

    def evaluate_total(self, n_saved=None):
        """train.py:96-118: greedy evaluation of every saved checkpoint, then of the final one.  The reference counts
        n_steps // evaluate_cycle checkpoints; the failure-inflated step count can end the loop before the last of
        them was written, so the count actually saved is used when known."""
        args = copy.copy(self.args)
        args.load_model, args.dist = True, False
        n = args.n_steps // args.evaluate_cycle if n_saved is None else n_saved
        names = ['{}_{}_'.format(args.ith_run, k) for k in range(n)] + ['{}_'.format(args.ith_run)]
        if self.dist:
            torch.distributed.barrier()  # rank 0 has written the files
        for name in names:
            args.load_model_name = name
            self._evaluate_and_record(Evaluator(self.env, Agents(args), args.episode_limit))
        self.train_data_save()

    def train_data_save(self):
        """File names of train.py:145-158 (prefix '{alg}_env(W,L,n,blocks,fov,stall)' inside save_path)."""
        if self.rank != 0:
            return
        os.makedirs(self.save_path, exist_ok=True)
        a = self.args
        prefix = self.save_path + '/{}'.format(a.alg) + '_env({},{},{},{},{},{})'.format(
            a.width, a.length, a.drop_num, a.block_num, a.fov, a.stall)
        i = a.ith_run
        np.save(prefix + 'Rewards_{}'.format(i), self.episode_rewards)
        np.save(prefix + 'steps_{}'.format(i), self.episode_steps)
        np.save(prefix + 'constraints_{}'.format(i), self.episode_constraints)
        np.save(prefix + 'success_rate_{}'.format(i), self.success_rate)
        np.save(prefix + 'runtime_{}'.format(i), self.time_cost)


def main(argv=None):
    """`python -m marl_dmfb_amd.train dmfb --drop_num=4 --fov=9 [--n_envs 4096] [--dist]` -- the reference's
    `python train.py dmfb --drop_num=4 --fov=9` (train.py:161-169) on the vectorised HIP env.  The schedule lengths
    are rescaled for the vectorised cadence by common/arguments.py:vectorise_schedule (see there)."""
    import torch.distributed as dist
    from .common.arguments import get_train_args
    args = get_train_args(argv)
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    use_dist = bool(args.dist and world > 1)
    if use_dist:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    try:
        if args.name == 'dmfb':
            from .env.dmfb import VecDMFB
            env = VecDMFB(args.width, args.length, args.drop_num, args.block_num, fov=args.fov, stall=args.stall,
                          n_envs=args.n_envs, seed=args.seed, env_id0=rank * args.n_envs)
        else:
            from .env.meda import VecMEDA
            env = VecMEDA(args.width, args.length, args.drop_num, fov=args.fov, n_envs=args.n_envs, seed=args.seed,
                          env_id0=rank * args.n_envs, version=2 if args.version == '0.2' else 0)
        args.__dict__.update(env.get_env_info())
        args.device = str(env.device)
        args.buffer_size = max(args.buffer_size, 4 * args.n_envs)
        Trainer(env, args).run(online_evaluate=args.online_eval)
    finally:
        if use_dist:
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
