"""bench.py -- env-steps/sec of the vectorised DMFB + VDN training loop on N MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

What one "step" is: one ROUND of the training loop of the reference's Trainer.run
(train.py:59-78) over the whole batch of chips = every chip plays one episode in lock-step
(Q-net forward over envs x agents, epsilon-greedy, fused HIP transition kernel; <= episode_limit
lock-steps), the episodes are stored in the HBM-resident replay buffer, then `train_time` VDN
learns of `batch_size` episodes run (forward, backward, gradient all-reduce when N > 1, clip, Adam).
`value` = env-steps actually played (padding excluded) by all ranks / wall time of the K rounds.

Workload (BASELINE.json configs[1]): DMFB 10x10, 4 droplets, fov 9, 4096 chips per GPU, synthetic
tasks from the Philox generator, randomly initialised CRNN (hyper_hidden_dim 24, fp32).
"""
import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--n_envs', type=int, default=4096, help='chips per GPU')
    ap.add_argument('--width', type=int, default=10)
    ap.add_argument('--length', type=int, default=10)
    ap.add_argument('--drop_num', type=int, default=4)
    ap.add_argument('--fov', type=int, default=9)
    ap.add_argument('--batch_size', type=int, default=512, help='episodes per learn')
    ap.add_argument('--train_time', type=int, default=4, help='learns per round')
    ap.add_argument('--buffer_size', type=int, default=16384, help='episodes kept in the HBM replay buffer')
    ap.add_argument('--graph', action='store_true', help='replay the rollout as a captured HIP graph (the in-loop kernel '
                    'timing of the roofline object is then taken from an eager pass after the timed region)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_tiers', action='store_true')
    ap.add_argument('--roofline_envs', type=int, default=262144, help='batch for the large-batch roofline figure')
    return ap.parse_args()


def env_only_tier(cfg, E, iters, device, fov_kernel=False):
    """Env-only tier: fused transition kernel with auto-reset, uniform random actions."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    env = VecDMFB(n_envs=E, seed=1, device=device, **cfg)
    env.reset()
    g = torch.Generator(device=device).manual_seed(0)
    acts = [torch.randint(0, 5, (E, cfg['n_agents']), device=device, generator=g, dtype=torch.int8) for _ in range(8)]
    for i in range(20):
        env.step(acts[i % 8], autoreset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        env.step(acts[i % 8], autoreset=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    b = algo_bytes_per_env_step(cfg['n_agents'], cfg['fov'])
    out = {'n_envs': E, 'us_per_launch': round(us, 2), 'env_steps_per_s': round(E / us * 1e6),
           'algo_GBps': round(E * b / us / 1e3, 1), 'frac': round(E * b / us / 1e3 / HBM_PEAK_GBPS, 4)}
    if fov_kernel:
        # the FOV-gather kernel alone (k_observe): n*(3 fov^2 + 2) bytes written + 4n+... read per chip
        env.observe()
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            env.observe()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        n, fov = cfg['n_agents'], cfg['fov']
        fb = n * (3 * fov * fov + 2) + 5 * n + 8  # FOV-gather kernel alone
        out['fov_kernel'] = {'kernel': 'dmfbk::k_observe<%d>' % n, 'us_per_launch': round(us, 2), 'algo_bytes_per_env': fb,
                             'algo_GBps': round(E * fb / us / 1e3, 1), 'frac': round(E * fb / us / 1e3 / HBM_PEAK_GBPS, 4)}
    env.close()
    return out


def cpu_baseline(cfg, args_ns, seconds=20.0):
    """The same loop on the host CPU: the C oracle env (kind "port", 1 thread) driven by the same
    host-side Agents/VDN code on torch CPU.  Bounded sample; rank 0, N = 1 only."""
    import numpy as np
    from oracle.dmfb_oracle import DmfbOracle  # cpu_baseline leg: allowed user of oracle/
    from marl_dmfb_amd.agent.agent import Agents
    from marl_dmfb_amd.common.arguments import make_args
    # the box exposes every host core but one GPU's share is 16 (more threads only add contention)
    cores = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    n, fov = cfg['n_agents'], cfg['fov']
    E = 128
    T = 2 * (cfg['width'] + cfg['length'])
    a = make_args(drop_num=n, width=cfg['width'], length=cfg['length'], fov=fov, cuda=False, device='cpu',
                  n_actions=5, n_agents=n, obs_shape=(3, fov, fov, 2, 3 * fov * fov + 2), episode_limit=T)
    agents = Agents(a)
    ora = DmfbOracle(n_envs=E, seed=1, **cfg)
    rng = np.random.default_rng(0)
    # env-only: reset + step + observe, random actions, single thread
    t0 = time.perf_counter()
    env_steps = 0
    ora.reset()
    while time.perf_counter() - t0 < seconds * 0.25:
        r, d, c, s = ora.step(rng.integers(0, 5, (E, n)).astype(np.int32))
        term = d.all(axis=1)
        if term.any():
            ora.reset(mask=term.astype(np.uint8))
        ora.observe()
        env_steps += E
    env_only = env_steps / (time.perf_counter() - t0)
    # full loop: one episode per chip with the CRNN on CPU, then learns at the GPU run's cadence
    t0 = time.perf_counter()
    played = 0
    rounds = 0
    while time.perf_counter() - t0 < seconds * 0.75 or rounds == 0:
        ora.reset()
        obs = torch.from_numpy(ora.observe())
        hidden = torch.zeros((E * n, a.rnn_hidden_dim))
        last = torch.zeros((E, n, 5), dtype=torch.int8)
        alive = np.ones(E, bool)
        ep = {'o': torch.zeros((E, T, n, obs.shape[-1]), dtype=torch.int8), 'o_next': torch.zeros((E, T, n, obs.shape[-1]), dtype=torch.int8),
              'u': torch.zeros((E, T, n, 1), dtype=torch.int8), 'r': torch.zeros((E, T, 1)),
              'avail_u': torch.zeros((E, T, n, 5), dtype=torch.int8), 'avail_u_next': torch.zeros((E, T, n, 5), dtype=torch.int8),
              'u_onehot': torch.zeros((E, T, n, 5), dtype=torch.int8), 'padded': torch.ones((E, T, 1), dtype=torch.bool),
              'terminated': torch.ones((E, T, 1), dtype=torch.bool)}
        for t in range(T):
            acts, hidden = agents.choose_actions(obs, last, hidden, 0.5)
            onehot = torch.nn.functional.one_hot(acts, 5).to(torch.int8)
            # the single-chip reference stops stepping a finished chip; emulate with a per-env mask
            idx = np.nonzero(alive)[0]
            ep['o'][idx, t] = obs[idx]
            r, d, c, s = ora.step(acts.numpy().astype(np.int32))
            obs = torch.from_numpy(ora.observe())
            term = d.all(axis=1)
            ep['o_next'][idx, t] = obs[idx]
            ep['u'][idx, t] = acts[idx].unsqueeze(-1).to(torch.int8)
            ep['u_onehot'][idx, t] = onehot[idx]
            ep['avail_u'][idx, t] = 1
            ep['avail_u_next'][idx, t] = 1
            ep['r'][idx, t, 0] = torch.from_numpy((r.sum(axis=1) / n)[idx]).float()
            ep['padded'][idx, t] = False
            ep['terminated'][idx, t, 0] = torch.from_numpy(term[idx])
            played += int(alive.sum())
            last = onehot
            alive = alive & ~term
            if not alive.any():
                break
        learns = max(1, round(args_ns.train_time * args_ns.batch_size * E / (args_ns.n_envs * 64)))
        for k in range(learns):
            sel = torch.randint(0, E, (64,))
            agents.train({key: v[sel] for key, v in ep.items()}, k)
        rounds += 1
    full = played / (time.perf_counter() - t0)
    return {'value': round(full, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'env_only_value_1core': round(env_only, 1),
            'sample': '%d rounds of %d chips x <=%d lock-steps (CRNN on torch CPU, %d threads) + learns of 64 '
                      'episodes at the GPU run\'s learn/collect ratio; env = C oracle, 1 thread' % (rounds, E, T, cores)}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # BENCH_FORCE_DIST=1 runs the collective path (RCCL init, broadcast, all-reduce) even with one rank
    force_dist = bool(os.environ.get('BENCH_FORCE_DIST'))
    dist = world > 1 or force_dist
    # BENCH_DIST_BACKEND=gloo rehearses the multi-rank control flow with all ranks on one GPU
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if dist:
        if force_dist and 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.train import Trainer

    cfg = dict(width=a.width, length=a.length, n_agents=a.drop_num, fov=a.fov)
    env = VecDMFB(n_envs=a.n_envs, seed=1234, env_id0=rank * a.n_envs, device=device, **cfg)
    args = make_args(drop_num=a.drop_num, width=a.width, length=a.length, fov=a.fov, device=str(device), dist=dist,
                     n_envs=a.n_envs, batch_size=a.batch_size, train_time=a.train_time, buffer_size=a.buffer_size,
                     use_graph=a.graph, force_dist=force_dist,
                     **env.get_env_info())
    torch.manual_seed(1234 + rank)
    trainer = Trainer(env, args)

    for _ in range(a.warmup):
        trainer.collect_and_learn()
    env.timing = []
    torch.cuda.synchronize()
    if dist:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    played = 0
    for _ in range(a.steps):
        played += trainer.collect_and_learn()
    torch.cuda.synchronize()
    if dist:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if not env.timing:  # graph replays do not re-record the events: time one eager episode of the same loop
        trainer.rolloutWorker.use_graph = False
        trainer.rolloutWorker.generate_episode()
        torch.cuda.synchronize()
    kern_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in env.timing]
    env.timing = None
    # an event pair with nothing between it still measures ~2-3 us of marker latency: calibrate and subtract,
    # so that the figure is the kernel's own duration (what rocprofv3 --kernel-trace reports)
    pairs = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ev = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in pairs)
    event_overhead_us = ev[len(ev) // 2]
    tot = torch.tensor([float(played), dt], device=device, dtype=torch.float64)
    if dist:
        p = tot[0:1].clone() if backend == 'nccl' else tot[0:1].cpu()
        torch.distributed.all_reduce(p, op=torch.distributed.ReduceOp.SUM)
        m = tot[1:2].clone() if backend == 'nccl' else tot[1:2].cpu()
        torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
        played_all, dt_max = float(p.item()), float(m.item())
    else:
        played_all, dt_max = float(played), dt

    if rank != 0:
        if dist:
            torch.distributed.destroy_process_group()
        return

    b = algo_bytes_per_env_step(a.drop_num, a.fov)
    raw_us = sum(kern_us) / max(1, len(kern_us))
    avg_us = max(raw_us - event_overhead_us, 0.1)
    achieved = a.n_envs * b / avg_us / 1e3  # GB/s
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get('k_step_%dx%d_%dd_E%d' % (a.width, a.length, a.drop_num, a.n_envs))
        except Exception:
            traffic = None
    out = {
        'metric': 'env-steps/sec (whole node), %dx%d DMFB %d-droplet fov%d' % (a.width, a.length, a.drop_num, a.fov),
        'value': round(played_all / dt_max, 1), 'unit': 'env-steps/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(dt_max / a.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int8 env state/obs + f64 rewards; fp32 Q-net',
        'data': 'synthetic (Philox task generator, random-init CRNN)',
        'config': {'workload': 'DMFB %dx%d, drop_num=%d, fov=%d, %d parallel envs per GPU%s' % (
            a.width, a.length, a.drop_num, a.fov, a.n_envs,
            ' (BASELINE configs[1])' if (a.width, a.length, a.drop_num, a.fov, a.n_envs) == (10, 10, 4, 9, 4096) else ''),
            'round': 'one episode per chip (<=%d lock-steps) + %d learns x %d episodes' % (
                env.max_step, a.train_time, a.batch_size),
            'parallelism': 'dp%d: chips sharded per rank, one flat RCCL all-reduce per learn' % world,
            'env_steps_per_round': round(played_all / a.steps, 1)},
        'roofline': {'bound': 'hbm', 'kernel': 'dmfbk::k_step<%d,false>' % a.drop_num, 'achieved': round(achieved, 1),
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 4),
                     'traffic': traffic, 'launches_timed': len(kern_us), 'avg_launch_us': round(avg_us, 2),
                     'event_pair_raw_us': round(raw_us, 2), 'event_overhead_us': round(event_overhead_us, 2),
                     'algo_bytes_per_env_step': b, 'envs_per_launch': a.n_envs},
    }
    if not a.no_tiers and world == 1:
        trainer = None
        torch.cuda.empty_cache()
        out['tiers'] = {'env_only_4096': env_only_tier(cfg, a.n_envs, 300, device),
                        'env_only_large_batch': env_only_tier(cfg, a.roofline_envs, 100, device, fov_kernel=True)}
    if not a.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(cfg, a)
    print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
