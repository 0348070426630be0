"""SURVEY 8 f1 / BASELINE config 5: train a VDN policy on DMFB 20x20, 10 droplets (od 32, the reference's 10d.yaml values,
vectorised cadence), then run the degradation sweep of the reference's evaDegre.py (:8-56) with it -- `chips` ageing chips x
`evaluate_epoch` epochs x `evaluate_task` greedy episodes -- and write rewards/steps/success/health.npy in the layout of the
reference-held /root/reference/DegreData/20by20-10d0b/*.npy.  `tools/compare_degre.py` prints the two side by side.

    python tools/train_degre.py --rounds 3125 --out gpurun_out/degre

The training env has no degradation (train.py:165 of the reference builds the plain env); the sweep env is
ENV(..., b_degrade=True, per_degrade=1.0) as evaDegre.py:35-36.  Every line of progress goes to <out>/train_log.jsonl."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from marl_dmfb_amd.common.arguments import TRAIN_PARAS, make_args
from marl_dmfb_amd.env.dmfb import VecDMFB
from marl_dmfb_amd.evaDegre import Degre_evaluator, save_results
from marl_dmfb_amd.train import Trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--width', type=int, default=20)
    ap.add_argument('--drop_num', type=int, default=10)
    ap.add_argument('--n_envs', type=int, default=4096)
    ap.add_argument('--block_num', type=int, default=0, help='obstacle blocks per chip (SURVEY 8 f4: GenRandomBlocks, dmfb.py:228-251)')
    ap.add_argument('--no_stall', action='store_true', help='stall=False (dmfb.py:331,345-346)')
    ap.add_argument('--no_sweep', action='store_true', help='training and evaluations only')
    ap.add_argument('--rounds', type=int, default=3125, help='rounds of one episode per chip (3125 x 4 learns = the reference\'s 12 500 learns)')
    ap.add_argument('--seconds', type=float, default=900.0, help='wall-clock cap of the training loop')
    ap.add_argument('--train_time', type=int, default=4)
    ap.add_argument('--batch_size', type=int, default=512)
    ap.add_argument('--buffer_mult', type=int, default=4, help='replay buffer = this many rounds of episodes')
    ap.add_argument('--anneal_rounds', type=float, default=78.0, help='epsilon reaches min_epsilon after this many rounds')
    ap.add_argument('--eval_every', type=int, default=250)
    ap.add_argument('--chips', type=int, default=5, help='ageing chips of the sweep (evaDegre.py: 5)')
    ap.add_argument('--evaluate_epoch', type=int, default=20)
    ap.add_argument('--evaluate_task', type=int, default=100)
    ap.add_argument('--seed', type=int, default=7)
    ap.add_argument('--out', default='gpurun_out/degre')
    ap.add_argument('--load', default='', help='skip training: model directory written by an earlier run')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    log = open(os.path.join(a.out, 'train_log.jsonl'), 'a')

    def emit(**kw):
        log.write(json.dumps(kw) + '\n')
        log.flush()
        print(json.dumps(kw), flush=True)

    W, n, E = a.width, a.drop_num, a.n_envs
    torch.manual_seed(a.seed)
    env = VecDMFB(W, W, n, a.block_num, fov=9, stall=not a.no_stall, n_envs=E, seed=a.seed, device='cuda:0')
    info = env.get_env_info()
    T = info['episode_limit']
    args = make_args(drop_num=n, width=W, length=W, fov=9, device='cuda:0', n_envs=E, batch_size=a.batch_size,
                     train_time=a.train_time, buffer_size=a.buffer_mult * E, anneal_steps=E * T * a.anneal_rounds,
                     model_dir=os.path.join(a.out, 'model'), load_model=bool(a.load), load_model_name='0_', **info)
    if a.load:
        args.model_dir = a.load
        args.load_model_name = '0_'
    tr = Trainer(env, args)
    emit(what='config', width=W, drop_num=n, block_num=a.block_num, stall=not a.no_stall, n_envs=E, od=args.hyper_hidden_dim, train_time=a.train_time, batch_size=a.batch_size,
         buffer=args.buffer_size, anneal_steps=args.anneal_steps, lr=args.lr, target_update_cycle=args.target_update_cycle,
         ref_yaml=TRAIN_PARAS[('dmfb', n)] if ('dmfb', n) in TRAIN_PARAS else None)
    env_steps, t0 = 0, time.time()
    if not a.load:
        r, s, c, ok = tr.rolloutWorker.evaluate(1)
        emit(what='eval', round=0, learns=0, env_steps=0, reward=r, steps=s, constraints=c, success=ok, wall=0.0)
        for k in range(a.rounds):
            env_steps += tr.collect_and_learn()
            if (k + 1) % a.eval_every == 0 or k + 1 == a.rounds or time.time() - t0 > a.seconds:
                r, s, c, ok = tr.rolloutWorker.evaluate(1)
                emit(what='eval', round=k + 1, learns=tr.trained_times, env_steps=env_steps, eps=float(tr.rolloutWorker.epsilon),
                     loss=float(tr.agents.policy.last_loss), reward=r, steps=s, constraints=c, success=ok, wall=time.time() - t0)
                tr.agents.policy.save_model()
            if time.time() - t0 > a.seconds:
                break
        tr.agents.policy.save_model()
    if a.no_sweep:
        return
    # ---- the sweep of evaDegre.py:29-56
    t1 = time.time()
    env2 = VecDMFB(W, W, n, fov=9, stall=True, b_degrade=True, per_degrade=1.0, n_envs=a.chips, seed=1, device='cuda:0')
    args.evaluate_epoch, args.evaluate_task = a.evaluate_epoch, a.evaluate_task
    ev = Degre_evaluator(env2, tr.agents, args)
    ev.use_graph = True
    rewards, steps, success, health = ev.evaluate_process()
    path = save_results(args, rewards, steps, success, health, root=os.path.join(a.out, 'DegreData_%dchips' % a.chips))
    emit(what='sweep', chips=a.chips, epochs=a.evaluate_epoch, tasks=a.evaluate_task, wall=time.time() - t1, path=path,
         success=np.round(success.mean(0), 3).tolist(), steps=np.round(steps.mean(0), 2).tolist(),
         rewards=np.round(rewards.mean(0), 3).tolist(), mean_health=np.round(health.mean(axis=(0, 2, 3)), 4).tolist(),
         train_env_steps=env_steps, train_learns=tr.trained_times)


if __name__ == '__main__':
    main()
