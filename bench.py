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
    ap.add_argument('--no_stream', dest='stream', action='store_false', default=None,
                    help='one episode per chip per round (finished chips idle) instead of the continuous rollout')
    ap.add_argument('--no_graph', dest='graph', action='store_false',
                    help='play the rollout eagerly instead of replaying it as a captured HIP graph (the default)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--dump_weights', default='', help='rank 0 saves the eval network\'s state_dict here after the timed rounds (tests)')
    ap.add_argument('--no_tiers', action='store_true')
    ap.add_argument('--cpu_seconds', type=float, default=24.0, help='CPU seconds per process of the cpu_baseline sample (three phases)')
    ap.add_argument('--roofline_envs', type=int, default=655360,
                    help='chips per launch of the roofline kernel: 655 360 x 980 B = 642 MB of output, 2.5x the 256 MiB Infinity Cache')
    ap.add_argument('--roofline_envs_cached', type=int, default=262144,
                    help='second, cache-assisted batch (257 MB of output) reported as tiers.fov_kernel_cache_resident')
    ap.add_argument('--trained_tier_rounds', type=int, default=500,
                    help='rounds of training before the env+policy tier with a trained policy is measured (0: skip that tier)')
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
    if trainer.stream:
        return loop_breakdown_stream(trainer, rounds)
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


def loop_breakdown_stream(trainer, rounds):
    """The same for the continuous rollout: a round = episode_limit lock-steps of every chip (episodes go into the ring on the
    device as they end, so there is no store phase), then the learns on host-drawn episodes."""
    a, w, buf = trainer.args, trainer.rolloutWorker, trainer.buffer
    K = int(getattr(a, 'round_steps', None) or a.episode_limit)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    played = closed = 0
    for _ in range(rounds):
        c, _, _, p = buf.sync_host(w.generate_steps(buf, K))
        played += p
        closed += c
    torch.cuda.synchronize()
    t_roll = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(rounds):
        for _ in range(a.train_time):
            idx, lens = buf.draw(min(buf.current_size, a.batch_size))
            if trainer._packed:
                trainer.agents.policy.learn_packed(buf.buffers, idx, lens, trainer.trained_times)
            else:
                trainer.agents.train(buf.gather(idx), trainer.trained_times, max_len=int(lens[0]))
            trainer.trained_times += 1
    torch.cuda.synchronize()
    t_learn = time.perf_counter() - t0
    return {'env_policy': {'what': 'rollouts only (continuous: %d lock-steps of every chip per round, finished episodes written into the '
                                   'replay ring on the device): Q-net forward + epsilon-greedy + env transition + reset of ended chips, no learn' % K,
                           'rounds': rounds, 'env_steps_per_s': round(played / t_roll, 1), 'rollout_ms': round(t_roll / rounds * 1e3, 3),
                           'episodes_closed_per_round': round(closed / rounds, 1)},
            'store_ms': 0.0,
            'learn_ms': round(t_learn / rounds * 1e3, 3),
            'learn_ms_what': '%d learns x %d episodes (host draw + gather + forward + backward + clip + Adam), each over its batch\'s '
                             'longest episode' % (a.train_time, a.batch_size)}


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
    return {'kernel': 'dmfbk::k_step<%d,%s,true>' % (n, 'true' if env.has_maps else 'false'), 'n_envs': env.n_envs, 'launches_timed': len(us),
            'event_pair_raw_us': round(raw, 2), 'empty_event_pair_us': round(ev[len(ev) // 2], 2),
            'algo_bytes_per_env_step': b, 'algo_GBps_raw': round(env.n_envs * b / raw / 1e3, 1),
            'frac_raw': round(env.n_envs * b / raw / 1e3 / HBM_PEAK_GBPS, 4),
            'note': 'launch-latency bound at this batch (SURVEY 8(d) caveat); nothing subtracted'}


def trained_policy_tier(cfg, a, device, rounds):
    """env+policy tier with a policy that actually finishes episodes: train a fresh learner for `rounds` rounds (the loop of
    tools/train_sanity.py: 4 learns x 256 episodes per round, epsilon annealed over the first 60 %), then time rollouts only -- with
    the finished chips kept out of the Q-network's conv front end and GRU-head kernel (the default, Evaluator.compact_every) and with
    every row riding along to the slowest chip (compact_every = 0).  The headline runs a random-init policy whose episodes all
    last the full 40 lock-steps, which is the best case for a lock-step batch; this is the other end."""
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.train import Trainer
    E = a.n_envs
    env = VecDMFB(n_envs=E, seed=7, device=device, **cfg)
    info = env.get_env_info()
    args = make_args(device=str(device), n_envs=E, batch_size=256, train_time=4, buffer_size=8 * E,
                     anneal_steps=E * info['episode_limit'] * rounds * 0.6, use_graph=a.graph, stream=a.stream, **info)
    torch.manual_seed(0)
    tr = Trainer(env, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        tr.collect_and_learn()
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    w = tr.rolloutWorker
    out = {'trained_rounds': rounds, 'train_seconds': round(t_train, 2), 'epsilon': round(float(w.epsilon), 4),
           'what': 'rollouts only (epsilon-greedy at the trained epsilon, episodes recorded), policy trained for %d rounds of %d chips' % (rounds, E)}
    # the FULL loop (rollout + learns at the headline's cadence) under this policy: the figure to hold against the random-init
    # headline (whose episodes all last episode_limit steps)
    tr.args.batch_size, tr.args.train_time = a.batch_size, a.train_time
    for _ in range(3):
        tr.collect_and_learn()
    torch.cuda.synchronize()
    reps_l = 12
    t0 = time.perf_counter()
    played_l = closed_l = succ_l = 0
    for _ in range(reps_l):
        played_l += tr.collect_and_learn()
        closed_l += tr.last_round.get('episodes', E)
        succ_l += tr.last_round.get('success', 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out['env_policy_learn'] = {
        'env_steps_per_s': round(played_l / dt, 1), 'ms_per_round': round(dt / reps_l * 1e3, 3), 'env_steps_per_round': round(played_l / reps_l, 1),
        'episodes_per_round': round(closed_l / reps_l, 1), 'mean_steps_per_episode': round(played_l / max(1, closed_l), 2),
        'success_rate': round(succ_l / max(1, closed_l), 3), 'continuous_rollout': bool(tr.stream),
        'what': 'full loop under the trained policy: %s + %d learns x %d episodes per round, %d rounds timed' % (
            ('%d lock-steps of every chip' % info['episode_limit']) if tr.stream else 'one episode per chip', a.train_time, a.batch_size, reps_l)}
    if tr.stream:
        buf = tr.buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p_s = c_s = 0
        for _ in range(reps_l):
            c, _, _, p = buf.sync_host(w.generate_steps(buf, info['episode_limit']))
            p_s += p
            c_s += c
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out['continuous_rollout'] = {'env_steps_per_s': round(p_s / dt, 1), 'rollout_ms': round(dt / reps_l * 1e3, 3),
                                     'episodes_per_round': round(c_s / reps_l, 1), 'what': 'rollouts only, every chip playing all the time'}
    w.live_threshold = 2.0   # this tier measures both forms whatever the live share (the default switches at 0.75)
    forms = (('skipping_finished_chips', w.compact_every or 4), ('all_rows_every_step', 0))
    reps = 6
    for rnd in range(2):     # A B A B: each form keeps its faster pass (clocks and caches drift for seconds after the training burst)
        for name, every in forms:
            w.compact_every = every          # (the captured graphs are keyed by the form in use)
            for _ in range(2):
                w.generate_episode()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            played, succ = 0, 0.0
            for _ in range(reps):
                _, steps, _, success, ep = w.generate_episode()
                played += int((~ep['padded']).sum().item())
                succ += float((success > 0).float().mean().item())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rec = {'env_steps_per_s': round(played / dt, 1), 'rollout_ms': round(dt / reps * 1e3, 3),
                   'mean_steps_per_episode': round(played / (reps * E), 2), 'success_rate': round(succ / reps, 3)}
            if name not in out or rec['rollout_ms'] < out[name]['rollout_ms']:
                out[name] = rec
            out['live_share'] = round(played / float(reps * E * info['episode_limit']), 3)
    env.close()
    return out


TRAFFIC_SOURCE = ('profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel at this grid (separate captures, '
                  'tools/capture_profiles.sh), NOT measured in this run; null = no capture for this shape')
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, f32 operands (the reference's arithmetic type)


def conv_front_roofline(net, n, fov, rows_list, device):
    """`roofline_loop`: the dominant kernel of the timed loop, the Q-network front end (conv1+ReLU+conv2+ReLU + vector MLP + concat,
    network/base_net.py:59-68) as ONE hand-written fp32-MFMA launch, timed alone at the loop's two launch shapes (rollout lock-step
    rows = chips x droplets; learn rows = episodes x T x droplets).  Useful FLOP per row = 2 x [(fov-2)^2 od 27 + (fov-4)^2 od od 9 +
    7 x 10] (MACs of conv1, conv2, mlp1); the kernel pads od 24 to two 16-wide tiles, padding is not counted."""
    od = net.convs[0].out_channels
    flop_row = 2 * ((fov - 2) ** 2 * od * 27 + (fov - 4) ** 2 * od * od * 9 + 7 * 10)
    g = torch.Generator(device=device).manual_seed(7)
    shapes = []
    for R in rows_list:
        obs = torch.randint(0, n + 1, (R, 3 * fov * fov + 2), device=device, generator=g, dtype=torch.int8)
        oh = torch.zeros((R, 5), dtype=torch.int8, device=device)
        oh[:, 1] = 1
        with torch.no_grad():
            if not net._hip_conv_ok(obs):
                return None
            for _ in range(5):
                net._front_features_hip(obs, oh, padded=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                net._front_features_hip(obs, oh, padded=True)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        tf = R * flop_row / us / 1e6
        shapes.append({'rows_per_launch': R, 'avg_launch_us': round(us, 2), 'achieved': round(tf, 2), 'frac': round(tf / FP32_MFMA_PEAK_TFLOPS, 4)})
    top = shapes[0]
    return {'bound': 'mfma', 'kernel': 'crnn_mfma::k_conv9_mfma<%d, 0>' % od, 'achieved': top['achieved'], 'peak': FP32_MFMA_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': top['frac'], 'dtype': 'fp32 operands and accumulation (v_mfma_f32_16x16x4_f32)',
            'rows_per_launch': top['rows_per_launch'], 'avg_launch_us': top['avg_launch_us'], 'useful_flop_per_row': flop_row,
            'other_launch_shapes': shapes[1:],
            'timing': '30 back-to-back launches between two HIP events on the launch stream (output buffer allocation included)'}


# ------------------------------------------------------------------------------------------------
# cpu_baseline: the reference's loop shape on the host cores (one single-chip process per core)
# ------------------------------------------------------------------------------------------------
def _cpu_env_worker(job):
    """Phases that never call backward(): (a) env only -- C oracle reset/step/observe, uniform random actions; (b) the reference's
    rollout -- per step n Q-net forwards of batch 1 through Agents.choose_action (common/rollout.py:19-39), torch CPU, ONE thread,
    no learn.  Such a process opens no GPU device node, so one runs per host core."""
    import numpy as np
    cfg, sec_env, sec_roll, seed = job
    from oracle.dmfb_oracle import DmfbOracle  # cpu_baseline leg: allowed user of oracle/
    n, fov = cfg['n_agents'], cfg['fov']
    ora = DmfbOracle(n_envs=1, seed=seed, **cfg)
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    env_steps = 0
    ora.reset()
    while time.perf_counter() - t0 < sec_env:
        r, d, c, s = ora.step(rng.integers(0, 5, (1, n)).astype(np.int32))
        if d.all():
            ora.reset()
        ora.observe()
        env_steps += 1
    t_env = time.perf_counter() - t0
    played, t_roll = 0, 0.0
    if sec_roll > 0:
        torch.set_num_threads(1)
        from marl_dmfb_amd.agent.agent import Agents
        from marl_dmfb_amd.common.arguments import make_args
        T = 2 * (cfg['width'] + cfg['length'])
        a = make_args(drop_num=n, width=cfg['width'], length=cfg['length'], fov=fov, cuda=False, device='cpu', n_actions=5, n_agents=n,
                      obs_shape=(3, fov, fov, 2, 3 * fov * fov + 2), episode_limit=T)
        torch.manual_seed(seed)
        np.random.seed(seed)
        agents = Agents(a)
        avail = np.ones(5)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < sec_roll:
            ora.reset()
            obs = ora.observe()[0]
            agents.policy.init_hidden(1)
            last = np.zeros((n, 5))
            for t in range(T):
                acts = [int(agents.choose_action(obs[i], last[i], i, avail, 0.5)) for i in range(n)]
                r, d, c, s = ora.step(np.asarray(acts, np.int32)[None])
                obs, last = ora.observe()[0], np.eye(5)[acts]
                played += 1
                if d.all():
                    break
        t_roll = time.perf_counter() - t0
    return {'env_steps': env_steps, 't_env': t_env, 'played': played, 't_roll': t_roll}


def _cpu_worker(job):
    """The full loop on one host core: a single chip (C oracle behind the reference-shaped reset/step protocol), the reference's
    rollout (n Q-net forwards of batch size 1 per step through Agents.choose_action, common/rollout.py:19-39) and
    VDN.learn on torch CPU with ONE thread, at the GPU run's learn/collect ratio."""
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
    avail = np.ones(5)
    episodes = []
    t0 = time.perf_counter()
    played = learn_s = 0.0
    learns = 0
    eps = 0.5
    while time.perf_counter() - t0 < seconds:
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
    return {'played': played, 't_full': t_full, 'learns': learns, 'learn_s': learn_s}


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def host_cores():
    """Cores this process may actually use: the scheduler affinity mask, capped by a cgroup CPU quota when one is set."""
    try:
        c = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        c = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    c = min(c, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    c = min(c, max(1, int(q / int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, c)


def cpu_baseline(cfg, a):
    """The reference's own loop shape timed on the host, one single-threaded single-chip process per core.
    Env = the C oracle (kind "port"; the reference's Python env cannot travel to the GPU box), so the env share is a generous
    stand-in: the reference's Python DMFBenv steps ~2.6e3/s per core.
    Phases: env-only and rollout (no learn) run on EVERY host core -- they never call backward(); the full loop with VDN.learn
    runs on five processes, because torch's autograd engine opens the GPU device nodes in every process that calls backward()
    (on CPU tensors too, whatever *_VISIBLE_DEVICES says: tools/probe/gpu_open_probe.py) and the GPU box admits six processes
    with the GPU open (this one + five).  `value` is the measured full loop on those five cores; `whole_host_loop_estimate`
    combines the whole-host rollout rate with the per-core learn cost measured there."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')  # never fork a process that holds a GPU context
    sec = float(a.cpu_seconds)
    cores_all = min(host_cores(), int(os.environ.get('BENCH_CPU_MAX_PROCS', '64')))
    sampled_per_collected = a.train_time * a.batch_size / float(a.n_envs)   # the GPU run's learn/collect ratio
    learn_batch = 32
    collect_per_learn = max(1, int(round(learn_batch / max(sampled_per_collected, 1e-9))))
    phases, note = {}, None
    try:
        jobs = [(cfg, sec * 0.25, sec * 0.35, 1000 + k) for k in range(cores_all)]
        with ctx.Pool(cores_all) as pool:
            res = pool.map_async(_cpu_env_worker, jobs).get(timeout=sec * 0.6 + 240)
        env_only = sum(r['env_steps'] / r['t_env'] for r in res)
        roll = sum(r['played'] / r['t_roll'] for r in res)
        phases['env_only'] = {'cores': cores_all, 'whole_host': round(env_only, 1), 'per_core': round(env_only / cores_all, 1),
                              'what': 'C oracle reset/step/observe, uniform random actions, %.0f s per process' % (sec * 0.25)}
        phases['rollout'] = {'cores': cores_all, 'whole_host': round(roll, 1), 'per_core': round(roll / cores_all, 1),
                             'what': 'env + %d Q-net forwards of batch 1 per step (Agents.choose_action, torch CPU 1 thread), no learn, '
                                     '%.0f s per process' % (cfg['n_agents'], sec * 0.35)}
    except Exception as e:  # noqa: BLE001  (the baseline must never take the bench line down)
        note = 'whole-host phases failed: %s' % type(e).__name__
    cores = max(1, min(5, cores_all))
    jobs = [(cfg, sec * 0.4, 100 + k, collect_per_learn, learn_batch) for k in range(cores)]
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    full = sum(r['played'] / r['t_full'] for r in res)
    learns = int(sum(r['learns'] for r in res))
    learn_s_per_step = sum(r['learn_s'] for r in res) / max(1.0, sum(r['played'] for r in res))
    phases['loop_with_learn'] = {'cores': cores, 'value': round(full, 1), 'per_core': round(full / cores, 1), 'learns': learns,
                                 'learn_core_seconds_per_env_step': round(learn_s_per_step, 6),
                                 'what': 'rollout + one VDN.learn of %d episodes per %d collected (= the GPU run\'s %.2f sampled per '
                                         'collected episode), %.0f s per process' % (learn_batch, collect_per_learn, sampled_per_collected, sec * 0.4)}
    out = {'value': round(full, 1), 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'cpu_model': _cpu_model(),
           'per_core': round(full / cores, 1), 'host_cores': cores_all, 'phases': phases,
           'sample': '%d processes x %.0f s, one chip each, the reference loop shape: per step %d Q-net forwards of batch 1 via '
                     'Agents.choose_action (torch CPU, 1 thread), env = C oracle, one VDN.learn of %d episodes per %d collected; '
                     'env-only and rollout phases on all %d host cores (phases.*)'
                     % (cores, sec * 0.4, cfg['n_agents'], learn_batch, collect_per_learn, cores_all)}
    if 'rollout' in phases:
        per_core_roll = phases['rollout']['per_core']
        est = cores_all / (1.0 / max(per_core_roll, 1e-9) + learn_s_per_step)
        out['env_only_value'] = phases['env_only']['whole_host']
        out['env_only_per_core'] = phases['env_only']['per_core']
        out['whole_host_loop_estimate'] = {'value': round(est, 1), 'cores': cores_all,
                                           'how': 'cores / (1 / rollout per-core rate + learn core-seconds per env-step)'}
    if note:
        out['note'] = note
    return out


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
                     use_graph=a.graph, force_dist=force_dist, stream=a.stream,
                     **env.get_env_info())
    torch.manual_seed(1234 + rank)
    trainer = Trainer(env, args)
    pol = trainer.agents.policy

    def one_step():
        if not a.eval_only:
            return trainer.collect_and_learn()
        trainer.rolloutWorker._generate_episode()  # greedy episode on every chip; the chips keep ageing (evaDegre.py:19-22)
        return int(trainer.rolloutWorker.last_played.item())

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist:
        pol.allreduce_events = []  # every gradient all-reduce of the timed region is bracketed by a HIP event pair
        torch.distributed.barrier()
    t0 = time.perf_counter()
    played = 0
    for _ in range(a.steps):
        played += one_step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0  # this rank's own time, before it waits for the slowest one
    if dist:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0

    if a.dump_weights and rank == 0:
        torch.save({k: v.detach().cpu() for k, v in pol.eval_rnn.state_dict().items()}, a.dump_weights)
    tot = torch.tensor([float(played), dt], device=device, dtype=torch.float64)
    if dist:
        p = tot[0:1].clone() if backend == 'nccl' else tot[0:1].cpu()
        torch.distributed.all_reduce(p, op=torch.distributed.ReduceOp.SUM)
        m = tot[1:2].clone() if backend == 'nccl' else tot[1:2].cpu()
        torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
        c = torch.ones(1, dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(c, op=torch.distributed.ReduceOp.SUM)
        played_all, dt_max, joined = float(p.item()), float(m.item()), int(c.item())
        # per-rank record (diagnosis of a sub-linear scaling curve): own round time, env steps played, mean all-reduce time
        ev = pol.allreduce_events or []
        ar_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / max(1, len(ev))
        pol.allreduce_events = None
        mine = torch.tensor([dt_own / a.steps * 1e3, float(played), ar_ms, float(len(ev))], dtype=torch.float64,
                            device=device if backend == 'nccl' else 'cpu')
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        per_rank = [[float(v) for v in t.tolist()] for t in every]
    else:
        played_all, dt_max, joined = float(played), dt, 1
        per_rank = None

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
                     ('continuous rollout: %d lock-steps of every chip (a chip whose episode ends starts the next one at once; episodes '
                      'close into the replay ring on the device) + %d learns x %d episodes' % (env.max_step, a.train_time, a.batch_size))
                     if trainer.stream else
                     'one episode per chip (<=%d lock-steps) + %d learns x %d episodes' % (env.max_step, a.train_time, a.batch_size),
            'parallelism': 'dp%d: chips sharded per rank, one flat RCCL all-reduce per learn' % world,
            'env_steps_per_round': round(played_all / a.steps, 1)},
    }
    from marl_dmfb_amd.common import gemm_tuning
    out['gemm_solutions'] = gemm_tuning.mode()
    if per_rank is not None:
        ms = [r[0] for r in per_rank]
        ar = [r[2] for r in per_rank]
        out['ranks'] = {'ms_per_step': [round(v, 3) for v in ms], 'ms_per_step_min': round(min(ms), 3), 'ms_per_step_max': round(max(ms), 3),
                        'played': [int(r[1]) for r in per_rank],
                        'allreduce_ms_per_learn': [round(v, 4) for v in ar], 'allreduce_ms_per_learn_max': round(max(ar), 4),
                        'allreduces_timed_per_rank': int(per_rank[0][3]),
                        'what': 'ms_per_step: each rank\'s own time per round before the closing barrier; allreduce_ms_per_learn: HIP event '
                                'pair on the compute stream around the flat gradient all-reduce (the next learn\'s replay sample is queued '
                                'inside that bracket and overlaps the collective)'}

    tiers = {}
    if meda:  # roofline of the MEDA observation kernel, timed like k_observe<n>: dispatch time stamps inside a lock-step loop
        trainer = None
        env.close()
        torch.cuda.empty_cache()
        from marl_dmfb_amd.env.meda import VecMEDA
        fb = n * (3 * fov * fov + 2) + 5 * n + 8
        try:
            tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')))
        except (OSError, ValueError):
            tj = {}

        def meda_obs_kernel(roof_E):
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
            big.close()
            us = us_sum / launches
            out_bytes = roof_E * n * (3 * fov * fov + 2)
            return {'kernel': 'k_meda_observe<%d> (obs_version 2)' % (4 if n <= 4 else 8 if n <= 8 else 16),
                    'achieved': round(roof_E * fb / us / 1e3, 1), 'frac': round(roof_E * fb / us / 1e3 / HBM_PEAK_GBPS, 4),
                    'traffic': tj.get('k_meda_observe_v0_2_%dx%d_%dd_E%d' % (cfg['width'], cfg['length'], n, roof_E)),
                    'envs_per_launch': roof_E, 'algo_bytes_per_env': fb, 'avg_launch_us': round(us, 2), 'launches_timed': launches,
                    'output_bytes_per_launch': out_bytes, 'output_over_infinity_cache': round(out_bytes / float(256 << 20), 2)}
        # the graded launch writes >= 2.4x the 256 MiB Infinity Cache (as the DMFB figure does); the cache-assisted 65 536-chip
        # launch (1.06x) stays as a tier
        roof_E = max(65536, min(a.roofline_envs, 163840))
        big = meda_obs_kernel(roof_E)
        out['roofline'] = dict({'bound': 'hbm', 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s'}, **big)
        out['roofline']['traffic_source'] = TRAFFIC_SOURCE
        out['roofline']['timing'] = ('HIP event pair per launch carrying the dispatch start/end time stamps, %d launches, inside an env-only '
                                     'lock-step loop (transition kernel + this kernel per lock-step); nothing subtracted' % big['launches_timed'])
        if roof_E > 65536:
            small = meda_obs_kernel(65536)
            small['note'] = 'cache-assisted: %.0f MB of output against the 256 MiB Infinity Cache; not an HBM figure' % (small['output_bytes_per_launch'] / 1e6)
            out['tiers'] = {'fov_kernel_cache_resident': small}
        print(json.dumps(out), flush=True)
        if dist:
            torch.distributed.destroy_process_group()
        return 0
    if world == 1 and not a.no_tiers and not a.eval_only:
        tiers.update(loop_breakdown(trainer, max(2, min(a.steps, 6))))
        tiers['env_policy_learn'] = {'env_steps_per_s': out['value'], 'what': out['config']['round']}
        tiers['in_loop_step_kernel'] = in_loop_step_kernel(trainer, env, n, fov)
    if rank == 0 and not a.eval_only and fov == 9:
        net = trainer.agents.policy.eval_rnn
        if hasattr(net, '_hip_conv_ok'):
            rl = conv_front_roofline(net, n, fov, [a.n_envs * n, a.batch_size * env.max_step * n], device)
            if rl:
                out['roofline_loop'] = rl
    trainer = None
    env.close()
    torch.cuda.empty_cache()
    if True:  # every N: rank 0 times the roofline kernel on its own GPU after the timed region
        # the roofline kernel: FOV gather at a batch whose OUTPUT is 2.5x the 256 MiB Infinity Cache (655 360 chips x 980 B =
        # 642 MB), so that `achieved` is an HBM figure; FETCH/WRITE_SIZE (`traffic`) cannot tell the Infinity Cache from HBM
        # (MI355X_MICROARCH.md, HBM), the batch size can.  The cache-assisted 262 144-chip launch (257 MB) is kept as a tier.
        big = env_only_tier(cfg, a.roofline_envs, 60, device, fov_kernel=True)
        fk = big['fov_kernel']
        tj = {}
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
            except Exception:
                tj = {}
        traffic = tj.get('k_observe_%dx%d_%dd_E%d' % (a.width, a.length, n, a.roofline_envs))
        out_bytes = a.roofline_envs * n * (3 * fov * fov + 2)
        out['roofline'] = {'bound': 'hbm', 'kernel': fk['kernel'], 'achieved': fk['algo_GBps'], 'peak': HBM_PEAK_GBPS,
                           'unit': 'GB/s', 'frac': fk['frac'], 'traffic': traffic, 'traffic_source': TRAFFIC_SOURCE, 'envs_per_launch': a.roofline_envs,
                           'algo_bytes_per_env': fk['algo_bytes_per_env'], 'avg_launch_us': fk['us_per_launch'],
                           'launches_timed': fk['launches_timed'],
                           'back_to_back_avg_launch_us': fk['back_to_back_us_per_launch'],
                           'output_bytes_per_launch': out_bytes, 'output_over_infinity_cache': round(out_bytes / float(256 << 20), 2),
                           'timing': 'HIP event pair per launch carrying the dispatch start/end time stamps, %d launches, %s; nothing '
                                     'subtracted' % (fk['launches_timed'], fk['where'])}
        if a.roofline_envs_cached and a.roofline_envs_cached != a.roofline_envs:
            small = env_only_tier(cfg, a.roofline_envs_cached, 100, device, fov_kernel=True)
            sk = small['fov_kernel']
            tiers['fov_kernel_cache_resident'] = {
                'kernel': sk['kernel'], 'envs_per_launch': a.roofline_envs_cached, 'avg_launch_us': sk['us_per_launch'],
                'algo_GBps': sk['algo_GBps'], 'frac': sk['frac'], 'back_to_back_avg_launch_us': sk['back_to_back_us_per_launch'],
                'traffic': tj.get('k_observe_%dx%d_%dd_E%d' % (a.width, a.length, n, a.roofline_envs_cached)),
                'note': 'cache-assisted: %.0f MB of output against the 256 MiB Infinity Cache; not an HBM figure'
                        % (a.roofline_envs_cached * n * (3 * fov * fov + 2) / 1e6)}
            if not a.no_tiers and world == 1:
                tiers['env_only_cache_resident_batch'] = {k: v for k, v in small.items() if k != 'fov_kernel'}
        if not a.no_tiers and world == 1:
            tiers['env_only_large_batch'] = big
            tiers['env_only_%d' % a.n_envs] = env_only_tier(cfg, a.n_envs, 300, device)
            if a.trained_tier_rounds > 0 and not a.degrade and fov == 9:
                tiers['env_policy_trained'] = trained_policy_tier(cfg, a, device, a.trained_tier_rounds)
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
