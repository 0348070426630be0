"""Flags and training parameters of the reference (common/arguments.py:11-134 and
data-{dmfb,meda}/TrainParas/{2,3,4,5,10}d.yaml), as one function returning the `args` namespace
every other component reads.  Flag names and default values are the reference's; the YAML values
are inlined below (SURVEY.md section 5 lists them) so nothing depends on the working directory
(the reference chdir()s into data-dmfb/, common/config.py:5)."""
import argparse
from types import SimpleNamespace

# yaml document 1 (network) + document 2 (training), per (env name, drop_num)
_COMMON = dict(rnn_hidden_dim=128, qmix_hidden_dim=32, two_hyper_layers=True, lr=5.0e-4, epsilon=1.0,
               epsilon_anneal_scale='step', target_update_cycle=200)
TRAIN_PARAS = {
    ('dmfb', 2): dict(hyper_hidden_dim=32, n_episodes=5, anneal_steps=50000, min_epsilon=0.05, train_time=1, batch_size=128, buffer_size=5000, grad_norm_clip=10),
    ('dmfb', 3): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=100000, min_epsilon=0.05, train_time=1, batch_size=128, buffer_size=5000, grad_norm_clip=9),
    ('dmfb', 4): dict(hyper_hidden_dim=24, n_episodes=2, anneal_steps=150000, min_epsilon=0.05, train_time=1, batch_size=128, buffer_size=5000, grad_norm_clip=9),
    ('dmfb', 5): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=150000, min_epsilon=0.05, train_time=1, batch_size=128, buffer_size=5000, grad_norm_clip=9),
    ('dmfb', 10): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=50000, min_epsilon=0.05, train_time=1, batch_size=256, buffer_size=10000, grad_norm_clip=9),
    ('meda', 2): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=100000, min_epsilon=0.05, train_time=1, batch_size=64, buffer_size=10000, grad_norm_clip=10),
    ('meda', 3): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=300000, min_epsilon=0.05, train_time=2, batch_size=64, buffer_size=10000, grad_norm_clip=10),
    ('meda', 4): dict(hyper_hidden_dim=32, n_episodes=10, anneal_steps=300000, min_epsilon=0.05, train_time=2, batch_size=64, buffer_size=10000, grad_norm_clip=10),
    ('meda', 10): dict(hyper_hidden_dim=32, n_episodes=2, anneal_steps=300000, min_epsilon=0.01, train_time=2, batch_size=128, buffer_size=10000, grad_norm_clip=8),
}


def common_parser():
    p = argparse.ArgumentParser()
    p.add_argument('name', nargs='?', default='dmfb', choices=['dmfb', 'meda'])
    p.add_argument('--seed', type=int, default=12)
    p.add_argument('--alg', type=str, default='vdn')
    p.add_argument('--last_action', default=True, action='store_false')
    p.add_argument('--reuse_network', default=True, action='store_false')
    p.add_argument('--gamma', type=float, default=0.99)
    p.add_argument('--cuda', default=True, action='store_false')
    p.add_argument('--optimizer', type=str, default='ADAM')
    p.add_argument('--evaluate_task', type=int, default=100)
    p.add_argument('--model_dir', type=str, default='./model')
    p.add_argument('--result_dir', type=str, default='./TrainResult')
    p.add_argument('--load_model', default=False, action='store_true')
    p.add_argument('--load_model_name', type=str, default='')
    p.add_argument('--stall', default=True, action='store_false')
    p.add_argument('--drop_num', '-d', type=int, default=4)
    p.add_argument('--block_num', type=int, default=0)
    p.add_argument('--net', type=str, default='crnn')
    p.add_argument('--fov', type=int, default=None)
    p.add_argument('--width', '-w', '--chip_size', type=int, default=None)
    p.add_argument('--length', '-l', type=int, default=None)
    p.add_argument('--version', '-v', type=str, default=None)
    # vectorised-loop additions
    p.add_argument('--n_envs', type=int, default=4096, help='chips advanced in lock-step per GPU')
    p.add_argument('--no_graph', dest='use_graph', default=None, action='store_false',
                   help='play the rollout eagerly instead of replaying it as a captured HIP graph')
    p.add_argument('--dist', default=False, action='store_true', help='shard chips over ranks, all-reduce gradients')
    return p


def vectorise_schedule(args, ref, world=1):
    """Map the reference's per-2-episode training schedule onto rounds of n_envs episodes.

    The reference does `train_time` learns of `batch_size` episodes after every `n_episodes` collected episodes
    (train.py:62-78), i.e. 64 sampled episodes per collected one for 4d.yaml.  One vectorised round collects
    n_envs x world episodes; keeping that ratio would mean ~2000 learns per round ("Update-to-data
    ratio"), so the vectorised loop runs `train_time` (default 4) learns of `batch_size` (default max(yaml, n_envs/8))
    episodes per round instead, and every horizon the reference measures in ENV STEPS is stretched so that it spans
    the same number of LEARNS as in the reference:

        scale = (n_envs * ref.train_time) / (ref.n_episodes * train_time)   (--step_scale overrides; 1 = raw yaml numbers)
        anneal_steps            *= scale            (epsilon anneals on the env steps of a rank's OWN chips)
        n_steps, evaluate_cycle *= scale * world    (Trainer.time_steps counts the env steps of ALL ranks)

    `target_update_cycle` is counted in learns and stays (200).  With the defaults (dmfb, 4 droplets, 4096 chips):
    scale 512, epsilon reaches min_epsilon after ~3 750 learns as in the reference, ~50 000 learns in all, one
    checkpoint per ~2 500 learns."""
    if args.train_time is None:
        args.train_time = 4
    if args.batch_size is None:
        args.batch_size = max(ref['batch_size'], args.n_envs // 8)
    if args.anneal_steps is None:
        args.anneal_steps = ref['anneal_steps']
    scale = args.step_scale
    if scale is None:
        scale = (args.n_envs * ref['train_time']) / float(ref['n_episodes'] * max(1, args.train_time))
    args.step_scale = scale
    args.n_steps = int(round(args.n_steps * scale * world))
    args.anneal_steps = int(round(args.anneal_steps * scale))
    args.evaluate_cycle = int(round(args.evaluate_cycle * scale * world))
    return args


def set_default(args):
    """common/arguments.py:57-81."""
    if args.name == 'dmfb':
        if args.fov is None:
            args.fov = 9
        if args.width is None:
            args.width, args.length = 10, 10
        elif args.length is None:
            args.length = args.width
    else:
        if args.version is None:
            args.version = '0.2'
        if args.fov is None:
            args.fov = 19
        if args.width is None:
            args.width, args.length = (80, 80) if args.drop_num == 10 else (30, 60)
        elif args.length is None:
            args.length = args.width
    return args


def get_train_args(argv=None):
    p = common_parser()
    p.add_argument('--n_steps', type=int, default=20, help='total env steps x 100000')
    p.add_argument('--ith_run', '-i', type=int, default=0)
    p.add_argument('--replay_dir', type=str, default='')
    p.add_argument('--evaluate_cycle', type=int, default=100000)
    p.add_argument('--online_eval', default=True, action='store_false')
    # vectorised cadence (None = derived by vectorise_schedule)
    p.add_argument('--train_time', type=int, default=None, help='learns per round of n_envs episodes (default 4)')
    p.add_argument('--batch_size', type=int, default=None, help='episodes per learn (default max(yaml value, n_envs/8))')
    p.add_argument('--anneal_steps', type=int, default=None, help='epsilon anneal horizon in reference env steps (default: yaml)')
    p.add_argument('--step_scale', type=float, default=None,
                   help='factor applied to n_steps, anneal_steps and evaluate_cycle (default: keep the reference\'s horizons '
                        'measured in learns; 1 = raw reference numbers)')
    args = set_default(p.parse_args(argv))
    given = {k: getattr(args, k) for k in ('train_time', 'batch_size', 'anneal_steps')}
    ref = dict(TRAIN_PARAS[(args.name, args.drop_num)])
    args.__dict__.update(_COMMON)
    args.__dict__.update(ref)
    args.__dict__.update(given)
    args.n_steps = args.n_steps * 100000
    import os
    world = int(os.environ.get('WORLD_SIZE', '1')) if args.dist else 1
    return vectorise_schedule(args, ref, world)


def get_evaluate_args(argv=None):
    p = common_parser()
    p.add_argument('--b-degrade', default=True)
    p.add_argument('--per-degrade', type=float, default=0)
    p.add_argument('--evaluate_epoch', type=int, default=20)
    p.set_defaults(load_model=True, n_envs=5)
    args = set_default(p.parse_args(argv))
    args.__dict__.update(_COMMON)
    args.hyper_hidden_dim = TRAIN_PARAS[('dmfb', 4)]['hyper_hidden_dim']  # evaluation reads 4d.yaml doc 1 (:130-133)
    return args


def make_args(name='dmfb', drop_num=4, width=None, length=None, fov=None, **overrides):
    """Programmatic equivalent of get_train_args for tests/bench (no argv)."""
    a = SimpleNamespace(name=name, seed=12, alg='vdn', last_action=True, reuse_network=True, gamma=0.99, cuda=True,
                        optimizer='ADAM', evaluate_task=100, model_dir='./model', result_dir='./TrainResult',
                        load_model=False, load_model_name='', stall=True, drop_num=drop_num, block_num=0, net='crnn',
                        fov=fov, width=width, length=length, version=None, n_envs=4096, dist=False, n_steps=20 * 100000,
                        ith_run=0, replay_dir='', evaluate_cycle=100000, online_eval=True)
    set_default(a)
    a.__dict__.update(_COMMON)
    a.__dict__.update(TRAIN_PARAS[(name, drop_num)])
    a.__dict__.update(overrides)
    return a
