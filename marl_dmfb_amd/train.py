"""Training driver with the loop shape of the reference's Trainer.run (train.py:32-94; the stale
runner.py:29-77 has the same shape): collect episodes -> store -> train_time x (sample, learn),
checkpoint every `evaluate_cycle` env steps.

One `generate_episode()` here plays one episode on every chip of the batch, so a round collects
n_envs episodes instead of the reference's n_episodes (2); the learn cadence is therefore stated
explicitly: `train_time` learns of `batch_size` episodes per round (args.train_time/batch_size)."""
import os
import time

import numpy as np
import torch

from .agent.agent import Agents
from .common.replay_buffer import ReplayBuffer
from .common.rollout import RolloutWorker


class Trainer:
    def __init__(self, env, args):
        self.env = env
        self.args = args
        self.agents = Agents(args)
        self.rolloutWorker = RolloutWorker(env, self.agents, args)
        self.rolloutWorker.use_graph = bool(getattr(args, 'use_graph', False))
        self.buffer = ReplayBuffer(args, device=env.device)
        self.episode_rewards, self.episode_steps = [], []
        self.episode_constraints, self.success_rate, self.time_cost = [], [], []
        self.save_path = args.result_dir + '/' + args.alg + '/fov{}/{}by{}-{}d{}b'.format(
            args.fov, args.width, args.length, args.drop_num, args.block_num)
        self.time_steps = 0
        self.trained_times = 0

    def collect_and_learn(self):
        """One round of the outer loop (train.py:59-78).  Returns env steps played this round."""
        _, steps, _, success, episode = self.rolloutWorker.generate_episode()
        played = int((~episode['padded']).sum().item())
        self.buffer.store_episode(episode)
        for _ in range(self.args.train_time):
            mini_batch = self.buffer.sample(min(self.buffer.current_size, self.args.batch_size))
            self.agents.train(mini_batch, self.trained_times)
            self.trained_times += 1
        self.time_steps += int(steps.sum().item())  # failure-inflated count, as train.py:65
        return played

    def run(self, online_evaluate=False):
        evaluate_steps = -1
        start = time.time()
        while self.time_steps < self.args.n_steps:
            if self.time_steps // self.args.evaluate_cycle > evaluate_steps:
                evaluate_steps += 1
                self.time_cost.append(time.time() - start)
                self.agents.policy.save_model(evaluate_steps)
                if online_evaluate:
                    r, s, c, ok = self.rolloutWorker.evaluate(max(1, self.args.evaluate_task // self.env.n_envs))
                    self.episode_rewards.append(r); self.episode_steps.append(s)
                    self.episode_constraints.append(c); self.success_rate.append(ok)
                    self.train_data_save()
            self.collect_and_learn()
        self.agents.policy.save_model()
        self.time_cost.append(time.time() - start)

    def train_data_save(self):
        """File names of train.py:145-158."""
        os.makedirs(self.save_path, exist_ok=True)
        i = self.args.ith_run
        np.save(self.save_path + '/Rewards_{}'.format(i), self.episode_rewards)
        np.save(self.save_path + '/steps_{}'.format(i), self.episode_steps)
        np.save(self.save_path + '/constraints_{}'.format(i), self.episode_constraints)
        np.save(self.save_path + '/success_rate_{}'.format(i), self.success_rate)
        np.save(self.save_path + '/runtime_{}'.format(i), self.time_cost)


def main(argv=None):
    """`python -m marl_dmfb_amd.train dmfb --drop_num=4 --fov=9 [--n_envs 4096] [--dist]` -- the reference's
    `python train.py dmfb --drop_num=4 --fov=9` (train.py:161-169) on the vectorised HIP env."""
    import torch.distributed as dist
    from .common.arguments import get_train_args
    args = get_train_args(argv)
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if args.dist and world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
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


if __name__ == '__main__':
    main()
