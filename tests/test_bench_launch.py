"""bench.py --gpus N must start N cooperating ranks by itself when no launcher set WORLD_SIZE
(BASELINE.json metric "1/2/4/8 MI355X").  The CPU test checks the launch plumbing with
`--launch_check` (rendezvous + all-reduces, no GPU work); the GPU test runs the real workload with two ranks
sharing the one card over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=900)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    p = _run(['--gpus', '2', '--launch_check'], BENCH_DIST_BACKEND='gloo')
    assert p.returncode == 0, p.stderr[-2000:]
    rec = _json_line(p.stdout)
    assert rec['n_gpus'] == 2 and rec['max_rank'] == 1 and rec['asked'] == 2


def test_world_size_mismatch_is_an_error():
    e = dict(os.environ, WORLD_SIZE='3', RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launch_check'], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE' in p.stderr


@pytest.mark.gpu
def test_gpus_2_real_workload_on_one_card():
    """Two ranks share the card (gloo carries the collectives): the line must say n_gpus == 2 and the whole-job value
    must count both ranks' chips."""
    p = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--n_envs', '512', '--batch_size', '64', '--train_time', '1',
              '--buffer_size', '1024', '--roofline_envs', '16384', '--roofline_envs_cached', '0'], BENCH_DIST_BACKEND='gloo')
    assert p.returncode == 0, p.stderr[-3000:]
    rec = _json_line(p.stdout)
    assert rec['n_gpus'] == 2
    assert rec['config']['env_steps_per_round'] > 512 * 2  # both shards counted (>= 2 lock-steps per chip)
    assert rec['roofline']['kernel'] == 'dmfbk::k_observe<4>' and rec['roofline']['frac'] > 0
    assert 'cpu_baseline' not in rec
    # a multi-rank line must be diagnosable from the record alone: per-rank round time, env steps, all-reduce time, GEMM mode
    rk = rec['ranks']
    assert len(rk['ms_per_step']) == len(rk['played']) == len(rk['allreduce_ms_per_learn']) == 2
    assert rk['ms_per_step_min'] <= rk['ms_per_step_max'] <= rec['ms_per_step'] * 1.05
    assert sum(rk['played']) == round(rec['config']['env_steps_per_round'] * rec['steps'])
    assert rk['allreduces_timed_per_rank'] == 2 * 1 and min(rk['allreduce_ms_per_learn']) > 0     # steps x train_time
    assert rec['gemm_solutions'].startswith(('tuned', 'library default', 'tuning to'))


@pytest.mark.gpu
def test_one_rank_over_rccl_reproduces_the_plain_run(tmp_path):
    """The RCCL path with one rank (BENCH_FORCE_DIST=1: process group on the `nccl` backend, parameter broadcast, the flat
    gradient all-reduce of every learn with the next sample queued behind it, the step count riding along): every learn's
    all-reduce is timed, and -- the un-normalised gradients being divided by the (all-reduced) mask count inside the clip + Adam
    kernel on both paths -- the weights after the run equal those of the plain run bit for bit."""
    import torch
    common = ['--steps', '2', '--warmup', '1', '--no_cpu_baseline', '--no_tiers', '--n_envs', '1024', '--batch_size', '128',
              '--train_time', '3', '--buffer_size', '4096', '--roofline_envs', '16384', '--roofline_envs_cached', '0']
    recs, weights = [], []
    for name, env in (('dist', {'BENCH_FORCE_DIST': '1'}), ('plain', {})):
        path = str(tmp_path / (name + '.pt'))
        p = _run(common + ['--dump_weights', path], **env)
        assert p.returncode == 0, p.stderr[-3000:]
        recs.append(_json_line(p.stdout))
        weights.append(torch.load(path))
    dist, plain = recs
    assert dist['n_gpus'] == 1 and 'ranks' in dist and 'ranks' not in plain
    assert dist['ranks']['allreduces_timed_per_rank'] == 2 * 3                 # steps x train_time
    assert dist['ranks']['allreduce_ms_per_learn'][0] > 0
    assert dist['config']['env_steps_per_round'] == plain['config']['env_steps_per_round']
    for k in weights[0]:
        assert torch.equal(weights[0][k], weights[1][k]), 'RCCL one-rank run and plain run differ in %s' % k
