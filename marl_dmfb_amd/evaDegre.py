"""Electrode-degradation sweep with the loop of the reference's evaDegre.py (:8-56), vectorised.

The reference ages 5 chips one after another: per chip `evaluate_epoch` epochs, per epoch it
snapshots `routing_manager.m_health` and plays `evaluate_task` greedy episodes on the SAME chip, so
usage accumulates and `updateHealth` (dmfb.py:465-471) degrades electrodes between episodes.  Here
every chip of a `VecDMFB(b_degrade=True, per_degrade=1.0)` batch ages in parallel; the outputs have
the reference's layout with the chip axis first:
    rewards/steps/success: (chips, epochs)      health: (chips, epochs, W, L)
and are saved under the reference's file names (rewards.npy, steps.npy, success.npy, health.npy in
DegreData/{W}by{W}-{n}d{b}b/)."""
import os

import numpy as np
import torch

from .common.rollout import Evaluator


class Degre_evaluator(Evaluator):
    def __init__(self, env, agents, args):
        super().__init__(env, agents, args.episode_limit)
        self.evaluate_epoch = int(args.evaluate_epoch)
        self.evaluate_task = int(args.evaluate_task)

    def evaluate_process(self):
        E = self.n_envs
        W, L = self.env.width, self.env.length
        rewards = torch.zeros((E, self.evaluate_epoch), dtype=torch.float64, device=self.device)
        steps = torch.zeros_like(rewards)
        success = torch.zeros_like(rewards)
        health = torch.zeros((E, self.evaluate_epoch, W, L), dtype=torch.float64, device=self.device)
        for epoch in range(self.evaluate_epoch):
            health[:, epoch] = self.env.get_map('health')            # evaDegre.py:21
            for _ in range(self.evaluate_task):                       # Evaluator.evaluate (rollout.py:69-85)
                r, s, _, ok = self._generate_episode()
                rewards[:, epoch] += r
                steps[:, epoch] += s.double()
                success[:, epoch] += ok.double()
            rewards[:, epoch] /= self.evaluate_task
            steps[:, epoch] /= self.evaluate_task
            success[:, epoch] /= self.evaluate_task
        return rewards.cpu().numpy(), steps.cpu().numpy(), success.cpu().numpy(), health.cpu().numpy()


def save_results(args, rewards, steps, success, health, root='DegreData'):
    path = os.path.join(root, '{}by{}-{}d{}b'.format(args.width, args.width, args.drop_num, args.block_num))
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, 'rewards.npy'), rewards)
    np.save(os.path.join(path, 'steps.npy'), steps)
    np.save(os.path.join(path, 'success.npy'), success)
    np.save(os.path.join(path, 'health.npy'), health)
    return path


def main(argv=None):
    from .agent.agent import Agents
    from .common.arguments import get_evaluate_args
    from .env.dmfb import VecDMFB
    args = get_evaluate_args(argv)
    n_chips = getattr(args, 'n_envs', 5)
    env = VecDMFB(args.width, args.length, args.drop_num, args.block_num, fov=args.fov, stall=args.stall,
                  b_degrade=True, per_degrade=1.0, n_envs=n_chips, seed=1)
    args.__dict__.update(env.get_env_info())
    args.device = str(env.device)
    agents = Agents(args)
    ev = Degre_evaluator(env, agents, args)
    out = ev.evaluate_process()
    print('saved to', save_results(args, *out))


if __name__ == '__main__':
    main()
