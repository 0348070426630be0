"""does MEDA train with the CRNN (fov-19 tied-conv stack) on the MEDAEnv_v0_2 observation?  The reference's MEDA
training path is broken (env/MEDA/meda.py:676-681 against common/replay_buffer.py:10), so there is no curve
of its own to set this beside; this run is the evidence that the path the reference evidently intended -- `python train.py meda`
with version 0.2 (common/arguments.py:67-68) -- learns to route on this build.

    python tools/train_meda.py --rounds 1500 --out gpurun_out/train_meda

MEDA W x L, `drop_num` droplets, the reference's meda yaml values (TRAIN_PARAS), vectorised cadence: `train_time` learns x
`batch_size` episodes per round of one lock-step pass of `n_envs` chips.  Every line of progress goes to <out>/train_log.jsonl."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from marl_dmfb_amd.common.arguments import TRAIN_PARAS, make_args
from marl_dmfb_amd.env.meda import VecMEDA
from marl_dmfb_amd.train import Trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--width', type=int, default=30)
    ap.add_argument('--length', type=int, default=30)
    ap.add_argument('--drop_num', type=int, default=4)
    ap.add_argument('--n_envs', type=int, default=4096)
    ap.add_argument('--rounds', type=int, default=1500)
    ap.add_argument('--seconds', type=float, default=600.0, help='wall-clock cap of the training loop')
    ap.add_argument('--train_time', type=int, default=4)
    ap.add_argument('--batch_size', type=int, default=512)
    ap.add_argument('--buffer_mult', type=int, default=4, help='replay buffer = this many rounds of episodes')
    ap.add_argument('--anneal_rounds', type=float, default=100.0, help='epsilon reaches min_epsilon after this many rounds')
    ap.add_argument('--eval_every', type=int, default=100)
    ap.add_argument('--seed', type=int, default=7)
    ap.add_argument('--out', default='gpurun_out/train_meda')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    log = open(os.path.join(a.out, 'train_log.jsonl'), 'a')

    def emit(**kw):
        log.write(json.dumps(kw) + '\n')
        log.flush()
        print(json.dumps(kw), flush=True)

    n, E = a.drop_num, a.n_envs
    torch.manual_seed(a.seed)
    env = VecMEDA(n_envs=E, seed=a.seed, device='cuda:0', version=2, width=a.width, length=a.length, n_agents=n, fov=19)
    info = env.get_env_info()
    T = info['episode_limit']
    args = make_args(name='meda', drop_num=n, width=a.width, length=a.length, fov=19, device='cuda:0', n_envs=E, batch_size=a.batch_size,
                     train_time=a.train_time, buffer_size=a.buffer_mult * E, anneal_steps=E * T * a.anneal_rounds,
                     model_dir=os.path.join(a.out, 'model'), **info)
    tr = Trainer(env, args)
    emit(what='config', env='meda v0_2', width=a.width, length=a.length, drop_num=n, n_envs=E, episode_limit=T, od=args.hyper_hidden_dim,
         train_time=a.train_time, batch_size=a.batch_size, buffer=args.buffer_size, anneal_steps=args.anneal_steps, lr=args.lr,
         target_update_cycle=args.target_update_cycle, stream=bool(tr.stream), ref_yaml=TRAIN_PARAS.get(('meda', n)))
    env_steps, t0 = 0, time.time()
    r, s, c, ok = tr.rolloutWorker.evaluate(1)
    emit(what='eval', round=0, learns=0, env_steps=0, reward=r, steps=s, constraints=c, success=ok, wall=0.0)
    for k in range(a.rounds):
        env_steps += tr.collect_and_learn()
        if (k + 1) % a.eval_every == 0 or k + 1 == a.rounds or time.time() - t0 > a.seconds:
            r, s, c, ok = tr.rolloutWorker.evaluate(1)
            emit(what='eval', round=k + 1, learns=tr.trained_times, env_steps=env_steps, eps=float(tr.rolloutWorker.epsilon),
                 loss=float(tr.agents.policy.last_loss), reward=r, steps=s, constraints=c, success=ok, wall=time.time() - t0)
        if time.time() - t0 > a.seconds:
            break
    tr.agents.policy.save_model()
    emit(what='done', rounds=k + 1, learns=tr.trained_times, env_steps=env_steps, wall=time.time() - t0,
         env_steps_per_s=env_steps / (time.time() - t0))


if __name__ == '__main__':
    main()
