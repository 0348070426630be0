"""bench.py -- env-steps/sec of the vectorised DMFB + VDN training loop on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  Under torch.distributed.run
(WORLD_SIZE set) every process is one rank on one GPU (RCCL).  Started plainly with `--gpus N > 1`
the process launches N such ranks itself (child processes, before it touches the GPU) and relays
rank 0's line.  Rank 0 prints ONE JSON line.

What one "step" is: one ROUND of the training loop of the reference's Trainer.run
(train.py:59-78) over the whole batch of chips = every chip plays one episode in lock-step
(Q-net forward over envs x agents, epsilon-greedy, fused HIP transition kernel; <= episode_limit
lock-steps), the episodes are stored in the HBM-resident replay buffer, then `train_time` VDN
learns of `batch_size` episodes run (forward, backward, gradient all-reduce when N > 1, clip, Adam).
`value` = env-steps actually played (padding excluded) by all ranks / wall time of the K rounds.

Workload (BASELINE.json configs[1]): DMFB 10x10, 4 droplets, fov 9, 4096 chips per GPU, synthetic
tasks from the Philox generator, randomly initialised CRNN (hyper_hidden_dim 24, fp32).

`roofline` describes the FOV-gather kernel `dmfbk::k_observe<n>` (the kernel BASELINE.json's
north_star grades) at `--roofline_envs` chips per launch, timed live with HIP events that carry the
dispatch's own time stamps; nothing is subtracted from the event figures.  The same kernel inside the 4096-chip training
loop is launch-latency bound and is reported under `tiers.in_loop_step_kernel`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def algo_bytes_per_env_step(n, fov, degrade=False, ext_uniforms=False):
    """algorithmic bytes of one lock-step transition of one chip."""
    writes = n * (3 * fov * fov + 2) + 8 * n + n + 5
    reads = n + (8 * n if ext_uniforms else 0) + (8 * n if degrade else 0)
    state = 2 * (2 * n + n + 8) + (4 * n if degrade else 0)
    return writes + reads + state


def fov_kernel_bytes_per_env(n, fov):
    """the FOV-gather kernel alone, n*(3 fov^2 + 2) written + 5n + 8 read per chip."""
    return n * (3 * fov * fov + 2) + 5 * n + 8


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--n_envs', type=int, default=4096, help='chips per GPU')
    ap.add_argument('--width', type=int, default=10)
    ap.add_argument('--length', type=int, default=10)
    ap.add_argument('--drop_num', type=int, default=4)
    ap.add_argument('--fov', type=int, default=None, help='default 9 (dmfb) / 19 (meda)')
    ap.add_argument('--env', choices=['dmfb', 'meda'], default='dmfb', help='meda: MEDAEnv_v0_2 observation (the one that trains)')
    ap.add_argument('--degrade', action='store_true', help='b_degrade=True, per_degrade=1.0 (BASELINE config 5 / evaDegre.py chips)')
    ap.add_argument('--eval_only', action='store_true',
                    help='a step = one GREEDY evaluation episode per chip, no learn (Evaluator.evaluate / evaDegre.py path)')
    ap.add_argument('--batch_size', type=int, default=512, help='episodes per learn')
    ap.add_argument('--train_time', type=int, default=4, help='learns per round')
    ap.add_argument('--buffer_size', type=int, default=16384, help='episodes kept in the HBM replay buffer')
    ap.add_argument('--no_graph', dest='graph', action='store_false',
                    help='play the rollout eagerly instead of replaying it as a captured HIP graph (the default)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_tiers', action='store_true')
    ap.add_argument('--cpu_seconds', type=float, default=20.0, help='wall seconds of the cpu_baseline sample')
    ap.add_argument('--roofline_envs', type=int, default=262144, help='chips per launch of the roofline kernel')
    ap.add_argument('--launch_check', action='store_true',
                    help='rendezvous + collectives only (no GPU work): checks the --gpus N launch plumbing')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the N ranks ourselves, BEFORE this process touches the GPU
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(a, argv):
    """Child processes (one per GPU) under torch.distributed.run; relay rank 0's JSON line.  The parent makes
    no GPU call, so nothing that initialised the GPU is ever re-executed."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith('{') and '"n_gpus"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout[-4000:] + '\n' + proc.stderr[-8000:] + '\n')
        sys.stderr.write('bench.py: the %d-rank launch failed (rc %d)\n' % (a.gpus, proc.returncode))
        return proc.returncode or 1
    got = json.loads(line).get('n_gpus')
    if got != a.gpus:
        sys.stderr.write('bench.py: asked for %d ranks, %r joined\n' % (a.gpus, got))
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------
# tiers measured on the GPU
# ------------------------------------------------------------------------------------------------
def env_only_tier(cfg, E, iters, device, fov_kernel=False):
    """Env-only tier: transition + observation with auto-reset, uniform random actions, `iters` lock-steps between two
    HIP events.  With fov_kernel=True the FOV-gather kernel (k_observe, the second launch of every lock-step at this
    batch size) is timed inside that same loop: each of its launches carries a HIP event pair that receives the
    dispatch's own start/end time stamps (include/dmfb_vec.h: dmfb_vec_observe_timing) -- the per-kernel duration
    rocprofv3 --kernel-trace reports for the same command."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    env = VecDMFB(n_envs=E, seed=1, device=device, **cfg)
    env.reset()
    g = torch.Generator(device=device).manual_seed(0)
    acts = [torch.randint(0, 5, (E, cfg['n_agents']), device=device, generator=g, dtype=torch.int8) for _ in range(8)]
    for i in range(20):
        env.step(acts[i % 8], autoreset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if fov_kernel:
        env.observe_timing(True)
    e0.record()
    for i in range(iters):
        env.step(acts[i % 8], autoreset=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    b = algo_bytes_per_env_step(cfg['n_agents'], cfg['fov'], degrade=bool(cfg.get('b_degrade')))
    out = {'n_envs': E, 'us_per_lockstep': round(us, 2), 'env_steps_per_s': round(E / us * 1e6),
           'algo_bytes_per_env_step': b, 'algo_GBps': round(E * b / us / 1e3, 1),
           'frac': round(E * b / us / 1e3 / HBM_PEAK_GBPS, 4)}
    if fov_kernel:
        tot_us, launches = env.observe_timing_read()
        where = 'inside the env-only lock-step loop (step-only kernel + this kernel per lock-step)'
        if launches == 0:  # small batch: the lock-step is ONE fused launch; time the standalone observation launches instead
            for i in range(iters):
                env.observe()
            tot_us, launches = env.observe_timing_read()
            where = 'standalone launches (at this batch the lock-step is one fused kernel)'
        env.observe_timing(False)
        kus = tot_us / launches
        fb = fov_kernel_bytes_per_env(cfg['n_agents'], cfg['fov'])
        out['fov_kernel'] = {'kernel': 'dmfbk::k_observe<%d>' % cfg['n_agents'], 'launches_timed': launches, 'where': where,
                             'us_per_launch': round(kus, 2), 'algo_bytes_per_env': fb,
                             'algo_GBps': round(E * fb / kus / 1e3, 1), 'frac': round(E * fb / kus / 1e3 / HBM_PEAK_GBPS, 4)}
        # the same kernel launched back to back (nothing in between): every launch then also waits for the previous
        # launch's dirty L2 lines to be written back, which otherwise overlaps the (latency-bound) transition kernel
        for _ in range(5):
            env.observe()
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            env.observe()
        e1.record()
        torch.cuda.synchronize()
        out['fov_kernel']['back_to_back_us_per_launch'] = round(e0.elapsed_time(e1) * 1e3 / iters, 2)
    env.close()
    return out


def loop_breakdown(trainer, rounds):
    """env+policy tier (rollouts only, no learn) and the per-phase times of a round, each phase bracketed by a
    device synchronisation.  Runs after the timed region, on the same trainer."""
    a = trainer.args
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    played = 0
    for _ in range(rounds):
        _, _, _, _, ep = trainer.rolloutWorker.generate_episode()
        played += int((~ep['padded']).sum().item())
    torch.cuda.synchronize()
    t_roll = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(rounds):
        trainer.buffer.store_episode(ep)
    torch.cuda.synchronize()
    t_store = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(rounds):
        for _ in range(a.train_time):
            mb = trainer.buffer.sample(min(trainer.buffer.current_size, a.batch_size))
            trainer.agents.train(mb, trainer.trained_times)
            trainer.trained_times += 1
    torch.cuda.synchronize()
    t_learn = time.perf_counter() - t0
    return {'env_policy': {'what': 'rollouts only: Q-net forward + epsilon-greedy + env transition, no learn',
                           'rounds': rounds, 'env_steps_per_s': round(played / t_roll, 1),
                           'rollout_ms': round(t_roll / rounds * 1e3, 3)},
            'store_ms': round(t_store / rounds * 1e3, 3),
            'learn_ms': round(t_learn / rounds * 1e3, 3),
            'learn_ms_what': '%d learns x %d episodes (sample + forward + backward + clip + Adam)' % (a.train_time, a.batch_size)}


def in_loop_step_kernel(trainer, env, n, fov):
    """The env transition launch as it runs inside the 4096-chip loop: one extra eager episode (outside the timed
    region) with a HIP event pair around every launch.  Raw event-pair figures: a pair around nothing already
    reads a few microseconds, so this over-states the kernel's own duration (rocprofv3 has that, profiles/)."""
    trainer.rolloutWorker.use_graph = False
    env.timing = []
    trainer.rolloutWorker.generate_episode()
    torch.cuda.synchronize()
    us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in env.timing]
    env.timing = None
    pairs = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ev = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs)
    raw = sum(us) / max(1, len(us))
    b = algo_bytes_per_env_step(n, fov, degrade=env.has_maps)
    return {'kernel': 'dmfbk::k_step<%d,%s>' % (n, 'true' if env.has_maps else 'false'), 'n_envs': env.n_envs, 'launches_timed': len(us),
            'event_pair_raw_us': round(raw, 2), 'empty_event_pair_us': round(ev[len(ev) // 2], 2),
            'algo_bytes_per_env_step': b, 'algo_GBps_raw': round(env.n_envs * b / raw / 1e3, 1),
            'frac_raw': round(env.n_envs * b / raw / 1e3 / HBM_PEAK_GBPS, 4),
            'note': 'launch-latency bound at this batch (SURVEY 8(d) caveat); nothing subtracted'}


# ------------------------------------------------------------------------------------------------
# cpu_baseline: the reference's loop shape on the host cores (one single-chip process per core)
# ------------------------------------------------------------------------------------------------
def _cpu_worker(job):
    """One host core: a single chip (C oracle behind the reference-shaped reset/step protocol), the reference's
    rollout (n Q-net forwards of batch size 1 per step through Agents.choose_action, common/rollout.py:19-39) and
    VDN.learn on torch CPU with ONE thread.  Returns counts and times of three phases."""
    import numpy as np
    torch.set_num_threads(1)
    from oracle.dmfb_oracle import DmfbOracle  # cpu_baseline leg: allowed user of oracle/
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    cfg, seconds, seed, collect_per_learn, learn_batch = job
    n, fov = cfg['n_agents'], cfg['fov']
    T = 2 * (cfg['width'] + cfg['length'])
    O = 3 * fov * fov + 2
    a = make_args(drop_num=n, width=cfg['width'], length=cfg['length'], fov=fov, cuda=False, device='cpu',
                  n_actions=5, n_agents=n, obs_shape=(3, fov, fov, 2, O), episode_limit=T)
    torch.manual_seed(seed)
    np.random.seed(seed)
    agents = Agents(a)
    ora = DmfbOracle(n_envs=1, seed=seed, **cfg)
    rng = np.random.default_rng(seed)
    # phase 1: env only -- reset + step + observe, uniform random actions
    t0 = time.perf_counter()
    env_steps = 0
    ora.reset()
    while time.perf_counter() - t0 < seconds * 0.15:
        r, d, c, s = ora.step(rng.integers(0, 5, (1, n)).astype(np.int32))
        if d.all():
            ora.reset()
        ora.observe()
        env_steps += 1
    t_env = time.perf_counter() - t0
    # phase 2+3: the full loop -- episodes through choose_action (B = 1 per agent), learns at the GPU run's
    # learn/collect ratio
    avail = np.ones(5)
    episodes = []
    t0 = time.perf_counter()
    played = learn_s = 0.0
    learns = 0
    eps = 0.5
    while time.perf_counter() - t0 < seconds * 0.85:
        ora.reset()
        obs = ora.observe()[0]
        agents.policy.init_hidden(1)
        last = np.zeros((n, 5))
        ep = {'o': np.zeros((T, n, O), np.int8), 'o_next': np.zeros((T, n, O), np.int8), 'u': np.zeros((T, n, 1), np.int8),
              'r': np.zeros((T, 1), np.float32), 'avail_u': np.zeros((T, n, 5), np.int8), 'avail_u_next': np.zeros((T, n, 5), np.int8),
              'u_onehot': np.zeros((T, n, 5), np.int8), 'padded': np.ones((T, 1), bool), 'terminated': np.ones((T, 1), bool)}
        for t in range(T):
            acts = [int(agents.choose_action(obs[i], last[i], i, avail, eps)) for i in range(n)]
            r, d, c, s = ora.step(np.asarray(acts, np.int32)[None])
            nxt = ora.observe()[0]
            onehot = np.eye(5)[acts]
            ep['o'][t], ep['o_next'][t], ep['u'][t, :, 0], ep['u_onehot'][t] = obs, nxt, acts, onehot
            ep['r'][t, 0] = r.sum() / n
            ep['avail_u'][t] = 1; ep['avail_u_next'][t] = 1
            ep['padded'][t] = False; ep['terminated'][t] = bool(d.all())
            obs, last = nxt, onehot
            played += 1
            if d.all():
                break
        episodes.append(ep)
        if len(episodes) % collect_per_learn == 0:
            tl = time.perf_counter()
            sel = rng.integers(0, len(episodes), learn_batch)
            batch = {k: torch.from_numpy(np.stack([episodes[j][k] for j in sel])) for k in ep}
            agents.train(batch, learns)
            learns += 1
            learn_s += time.perf_counter() - tl
            episodes = episodes[-4 * collect_per_learn:]
    t_full = time.perf_counter() - t0
    return {'env_steps': env_steps, 't_env': t_env, 'played': played, 't_full': t_full, 'learns': learns, 'learn_s': learn_s}


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(cfg, a):
    """The reference's own loop shape timed on the host: one single-threaded process per core, each a single chip +
    B=1 Q-net forwards + VDN.learn.  Env = the C oracle (kind "port"; the reference's Python env cannot travel to the
    GPU box), so the env share is a generous stand-in: the reference's Python DMFBenv steps ~2.6e3/s per core
    (BASELINE.md).  Five processes, not one per host core: torch's autograd engine opens the GPU device nodes in
    every process that calls backward() -- also on CPU tensors, whatever *_VISIBLE_DEVICES says (tools/probe/
    gpu_open_probe.py) -- and the GPU box admits six processes with the GPU open (this one + five)."""
    import multiprocessing as mp
    cores = max(1, min(5, os.cpu_count() or 1))
    sampled_per_collected = a.train_time * a.batch_size / float(a.n_envs)   # the GPU run's learn/collect ratio
    learn_batch = 32
    collect_per_learn = max(1, int(round(learn_batch / max(sampled_per_collected, 1e-9))))
    ctx = mp.get_context('spawn')  # never fork a process that holds a GPU context
    jobs = [(cfg, float(a.cpu_seconds), 100 + k, collect_per_learn, learn_batch) for k in range(cores)]
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    env_only = sum(r['env_steps'] / r['t_env'] for r in res)
    full = sum(r['played'] / r['t_full'] for r in res)
    return {'value': round(full, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'cpu_model': _cpu_model(),
            'per_core': round(full / cores, 1), 'env_only_value': round(env_only, 1), 'env_only_per_core': round(env_only / cores, 1),
            'learns': int(sum(r['learns'] for r in res)),
            'sample': '%d processes x %.0f s, one chip each: 15%% env-only (C oracle reset/step/observe, random actions), 85%% the '
                      'reference loop shape (per step %d Q-net forwards of batch 1 via Agents.choose_action, torch CPU 1 thread; '
                      'one VDN.learn of %d episodes per %d collected = the GPU run\'s %.2f sampled per collected episode)'
                      % (cores, a.cpu_seconds, cfg['n_agents'], learn_batch, collect_per_learn, sampled_per_collected)}


# ------------------------------------------------------------------------------------------------
def launch_check(a, world, rank, backend):
    """Rendezvous, one SUM and one MAX all-reduce, one line from rank 0.  No GPU work: this only proves that
    `--gpus N` produces N cooperating ranks."""
    import torch.distributed as dist
    dist.init_process_group('gloo' if backend != 'nccl' or not torch.cuda.is_available() else backend)
    t = torch.tensor([1.0, float(rank)], dtype=torch.float64)
    dist.all_reduce(t[0:1], op=dist.ReduceOp.SUM)
    dist.all_reduce(t[1:2], op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({'launch_check': True, 'n_gpus': int(t[0].item()), 'max_rank': int(t[1].item()),
                          'asked': a.gpus}), flush=True)
    dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(a, argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (a.gpus, world))
        return 2
    # BENCH_DIST_BACKEND=gloo rehearses the multi-rank control flow with all ranks on one GPU
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    if a.launch_check:
        return launch_check(a, world, rank, backend)
    # BENCH_FORCE_DIST=1 runs the collective path (RCCL init, broadcast, all-reduce) even with one rank
    force_dist = bool(os.environ.get('BENCH_FORCE_DIST'))
    dist = world > 1 or force_dist
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if dist:
        if force_dist and 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.train import Trainer

    meda = a.env == 'meda'
    if a.fov is None:
        a.fov = 19 if meda else 9
    cfg = dict(width=a.width, length=a.length, n_agents=a.drop_num, fov=a.fov)
    if a.degrade:
        cfg.update(b_degrade=True, per_degrade=1.0)
    if meda:
        from marl_dmfb_amd.env.meda import VecMEDA
        env = VecMEDA(n_envs=a.n_envs, seed=1234, env_id0=rank * a.n_envs, device=device, version=2, **cfg)
    else:
        env = VecDMFB(n_envs=a.n_envs, seed=1234, env_id0=rank * a.n_envs, device=device, **cfg)
    args = make_args(name=a.env, drop_num=a.drop_num, width=a.width, length=a.length, fov=a.fov, device=str(device), dist=dist,
                     n_envs=a.n_envs, batch_size=a.batch_size, train_time=a.train_time, buffer_size=a.buffer_size,
                     use_graph=a.graph, force_dist=force_dist,
                     **env.get_env_info())
    torch.manual_seed(1234 + rank)
    trainer = Trainer(env, args)

    def one_step():
        if not a.eval_only:
            return trainer.collect_and_learn()
        trainer.rolloutWorker._generate_episode()  # greedy episode on every chip; the chips keep ageing (evaDegre.py:19-22)
        return int(trainer.rolloutWorker.last_played.item())

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    played = 0
    for _ in range(a.steps):
        played += one_step()
    torch.cuda.synchronize()
    if dist:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0

    tot = torch.tensor([float(played), dt], device=device, dtype=torch.float64)
    if dist:
        p = tot[0:1].clone() if backend == 'nccl' else tot[0:1].cpu()
        torch.distributed.all_reduce(p, op=torch.distributed.ReduceOp.SUM)
        m = tot[1:2].clone() if backend == 'nccl' else tot[1:2].cpu()
        torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
        c = torch.ones(1, dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.SUM)
        played_all, dt_max, joined = float(p.item()), float(m.item()), int(c.item())
    else:
        played_all, dt_max, joined = float(played), dt, 1

    if rank != 0:
        if dist:
            torch.distributed.destroy_process_group()
        return 0

    n, fov = a.drop_num, a.fov
    what = ('MEDA' if meda else 'DMFB') + (' degrade' if a.degrade else '')
    out = {
        'metric': 'env-steps/sec (whole node), %dx%d %s %d-droplet fov%d' % (a.width, a.length, what, n, fov),
        'value': round(played_all / dt_max, 1), 'unit': 'env-steps/s', 'n_gpus': joined, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(dt_max / a.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8 env state/obs + f64 rewards; fp32 Q-net',
        'data': 'synthetic (Philox task generator, random-init CRNN)',
        'config': {'workload': '%s %dx%d, drop_num=%d, fov=%d, %d parallel envs per GPU%s' % (
            what, a.width, a.length, n, fov, a.n_envs,
            ' (BASELINE configs[1])' if (a.env, a.degrade, a.width, a.length, n, fov, a.n_envs) == ('dmfb', False, 10, 10, 4, 9, 4096) else ''),
            'round': ('one GREEDY evaluation episode per chip (<=%d lock-steps), no learn' % env.max_step) if a.eval_only else
                     'one episode per chip (<=%d lock-steps) + %d learns x %d episodes' % (env.max_step, a.train_time, a.batch_size),
            'parallelism': 'dp%d: chips sharded per rank, one flat RCCL all-reduce per learn' % world,
            'env_steps_per_round': round(played_all / a.steps, 1)},
    }
    tiers = {}
    if meda:  # roofline of the MEDA observation kernel, timed like k_observe<n>: dispatch time stamps inside a lock-step loop
        roof_E = min(a.roofline_envs, 65536)
        trainer = None
        env.close()
        torch.cuda.empty_cache()
        from marl_dmfb_amd.env.meda import VecMEDA
        big = VecMEDA(n_envs=roof_E, seed=1, device=device, version=2, **cfg)
        big.reset()
        g = torch.Generator(device=device).manual_seed(0)
        acts = [torch.randint(0, 9, (roof_E, n), device=device, generator=g, dtype=torch.int8) for _ in range(8)]
        for i in range(10):
            big.step(acts[i % 8], autoreset=True)
        torch.cuda.synchronize()
        big.observe_timing(True)
        for i in range(100):
            big.step(acts[i % 8], autoreset=True)
        us_sum, launches = big.observe_timing_read()
        big.observe_timing(False)
        us = us_sum / launches
        fb = n * (3 * fov * fov + 2) + 5 * n + 8
        traffic = None
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')))
            traffic = tj.get('k_meda_observe_v0_2_%dx%d_%dd_E%d' % (cfg['width'], cfg['length'], n, roof_E))
        except (OSError, ValueError):
            pass
        out['roofline'] = {'bound': 'hbm', 'kernel': 'k_meda_observe<%d> (obs_version 2)' % (4 if n <= 4 else 8 if n <= 8 else 16),
                           'achieved': round(roof_E * fb / us / 1e3, 1),
                           'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(roof_E * fb / us / 1e3 / HBM_PEAK_GBPS, 4), 'traffic': traffic,
                           'envs_per_launch': roof_E, 'algo_bytes_per_env': fb, 'avg_launch_us': round(us, 2), 'launches_timed': launches,
                           'timing': 'HIP event pair per launch carrying the dispatch start/end time stamps, %d launches, inside an '
                                     'env-only lock-step loop (transition kernel + this kernel per lock-step); nothing subtracted' % launches}
        big.close()
        print(json.dumps(out), flush=True)
        if dist:
            torch.distributed.destroy_process_group()
        return 0
    if world == 1 and not a.no_tiers and not a.eval_only:
        tiers.update(loop_breakdown(trainer, max(2, min(a.steps, 6))))
        tiers['env_policy_learn'] = {'env_steps_per_s': out['value'], 'what': out['config']['round']}
        tiers['in_loop_step_kernel'] = in_loop_step_kernel(trainer, env, n, fov)
    trainer = None
    env.close()
    torch.cuda.empty_cache()
    if True:  # every N: rank 0 times the roofline kernel on its own GPU after the timed region
        # the roofline kernel: FOV gather at a batch where the launch runs >= 50 us (SURVEY 8(d) launch-latency caveat)
        big = env_only_tier(cfg, a.roofline_envs, 100, device, fov_kernel=True)
        fk = big['fov_kernel']
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('k_observe_%dx%d_%dd_E%d' % (a.width, a.length, n, a.roofline_envs))
            except Exception:
                traffic = None
        out['roofline'] = {'bound': 'hbm', 'kernel': fk['kernel'], 'achieved': fk['algo_GBps'], 'peak': HBM_PEAK_GBPS,
                           'unit': 'GB/s', 'frac': fk['frac'], 'traffic': traffic, 'envs_per_launch': a.roofline_envs,
                           'algo_bytes_per_env': fk['algo_bytes_per_env'], 'avg_launch_us': fk['us_per_launch'],
                           'launches_timed': fk['launches_timed'],
                           'back_to_back_avg_launch_us': fk['back_to_back_us_per_launch'],
                           'timing': 'HIP event pair per launch carrying the dispatch start/end time stamps, %d launches, %s; nothing '
                                     'subtracted' % (fk['launches_timed'], fk['where'])}
        if not a.no_tiers and world == 1:
            tiers['env_only_large_batch'] = big
            tiers['env_only_%d' % a.n_envs] = env_only_tier(cfg, a.n_envs, 300, device)
    if tiers:
        out['tiers'] = tiers
    if not a.no_cpu_baseline and world == 1 and not a.eval_only:
        out['cpu_baseline'] = cpu_baseline(cfg, a)
    print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
