"""GPU parity tests proper: the HIP path (through the C ABI) against
  (1) the committed reference golden episodes, bit for bit;
  (2) the CPU oracle on identical seeds (Philox contract), bit for bit, incl. auto-reset,
      strict reset(mask), degradation maps, ragged batch sizes and all action dtypes;
  (3) size-independent invariants at BASELINE.json's full batch sizes."""
import os

import numpy as np
import pytest
import torch

from dmfb_replay import golden_files, replay, _bits
from oracle.dmfb_oracle import DmfbOracle

pytestmark = pytest.mark.gpu


def _vec(**kw):
    from vec_adapter import make_vec
    return make_vec(**kw)


@pytest.mark.parametrize('path', golden_files(), ids=os.path.basename)
def test_hip_replays_reference_golden(path):
    assert replay(path, _vec) > 0


def _lockstep(cfg, E, steps, seed, autoreset, record=True, act_dtype=np.int32, greedy=0.6, explicit_health=True):
    """Oracle and HIP env on the same Philox seed, same actions: every output must match."""
    O = DmfbOracle(n_envs=E, seed=seed, **cfg)
    V = _vec(n_envs=E, seed=seed, **cfg)
    rng = np.random.default_rng(seed + 7)
    so, eo = O.get_task()
    sv, ev = V.get_task()
    np.testing.assert_array_equal(so, sv)
    np.testing.assert_array_equal(eo, ev)
    if cfg.get('b_degrade'):
        np.testing.assert_array_equal(_bits(O.get_map('degrade')), _bits(V.get_map('degrade')))
        # age the chips so that health actually matters within the test
        h = rng.random((E, cfg['width'], cfg['length'])) * 0.6 + 0.4
        u = rng.integers(40, 52, (E, cfg['width'], cfg['length'])).astype(np.float64)
        for B in (O, V):
            if explicit_health:   # replaces the generator's map: the kernel gathers the float64 health map
                B.set_map('health', h)
            B.set_map('usage', u)  # (with the generator's own maps the kernel rebuilds health from the degrade counts)
    O.reset(); V.reset()
    np.testing.assert_array_equal(O.observe(), V.observe())
    n = cfg['n_agents']
    n_eps = 0
    for t in range(steps):
        st = O.get_state()
        so, eo = O.get_task()
        toward = np.zeros((E, n), np.int64)
        dx = eo[..., 0] - st['pos'][..., 0]
        dy = eo[..., 1] - st['pos'][..., 1]
        toward = np.where(dx > 0, 1, np.where(dx < 0, 2, np.where(dy < 0, 3, np.where(dy > 0, 4, 0))))
        rand = rng.integers(0, 5, (E, n))
        actions = np.where(rng.random((E, n)) < greedy, toward, rand).astype(act_dtype)
        ro, do, co, suo = O.step(actions.astype(np.int32), record=record)
        term = do.all(axis=1)
        rv, dv, cv, suv = V.step(torch.as_tensor(actions).cuda(), record=record, autoreset=autoreset)
        np.testing.assert_array_equal(_bits(ro), _bits(rv), err_msg='rewards t=%d' % t)
        np.testing.assert_array_equal(do, dv, err_msg='dones t=%d' % t)
        np.testing.assert_array_equal(co, cv, err_msg='constraints t=%d' % t)
        np.testing.assert_array_equal(suo, suv, err_msg='success t=%d' % t)
        np.testing.assert_array_equal(term.astype(np.uint8), V.last_info['terminated'])
        team = np.array([np.sum([np.float64(x) for x in ro[e]]) / n for e in range(min(E, 64))])
        np.testing.assert_array_equal(_bits(team), _bits(V.last_info['team_reward'][:len(team)]))
        if not autoreset:
            np.testing.assert_array_equal(O.observe(), V.last_obs, err_msg='terminal obs t=%d' % t)
        if term.any():
            O.reset(mask=term.astype(np.uint8))
            if not autoreset:
                V.reset(mask=term.astype(np.uint8))
                V.last_obs = V.v.obs.cpu().numpy()
            n_eps += int(term.sum())
        np.testing.assert_array_equal(O.observe(), V.last_obs, err_msg='obs t=%d' % t)
        if cfg.get('n_blocks') and (t % 16 == 0 or term.any()):
            np.testing.assert_array_equal(O.get_blocks(), V.get_blocks(), err_msg='blocks t=%d' % t)
        sv_ = V.get_state()
        so_ = O.get_state()
        for k in ('pos', 'dist', 'step_count', 'constraints'):
            np.testing.assert_array_equal(so_[k], sv_[k], err_msg='%s t=%d' % (k, t))
    if cfg.get('b_degrade'):
        for m in ('health', 'usage', 'degrade'):
            np.testing.assert_array_equal(_bits(O.get_map(m)), _bits(V.get_map(m)), err_msg=m)
    return n_eps


A = dict(width=10, length=10, n_agents=4, fov=9)
D = dict(width=50, length=50, n_agents=10, fov=9)
Ecfg = dict(width=20, length=20, n_agents=10, fov=9, b_degrade=True, per_degrade=1.0)


def test_lockstep_A_autoreset():
    assert _lockstep(A, E=1024, steps=120, seed=3, autoreset=True) > 1000


def test_lockstep_A_strict_reset_ragged_batch():
    assert _lockstep(A, E=37, steps=90, seed=5, autoreset=False) > 10


def test_lockstep_A_int8_and_int64_actions():
    _lockstep(A, E=200, steps=30, seed=11, autoreset=True, act_dtype=np.int8)
    _lockstep(A, E=200, steps=30, seed=12, autoreset=True, act_dtype=np.int64)


def test_lockstep_D_50x50_10d():
    assert _lockstep(D, E=256, steps=260, seed=21, autoreset=True, greedy=0.9) > 50


def test_lockstep_E_degrade_chain():
    assert _lockstep(Ecfg, E=192, steps=200, seed=31, autoreset=True, greedy=0.8) > 100


def test_lockstep_E_degrade_strict():
    _lockstep(Ecfg, E=50, steps=100, seed=32, autoreset=False, greedy=0.8)


def test_lockstep_odd_shapes():
    _lockstep(dict(width=12, length=9, n_agents=3, fov=7, with_maps=True), E=100, steps=60, seed=41, autoreset=True)
    _lockstep(dict(width=10, length=10, n_agents=4, fov=6, stall=False), E=100, steps=60, seed=42, autoreset=True)
    _lockstep(dict(width=16, length=16, n_agents=7, fov=5), E=70, steps=80, seed=43, autoreset=True)


@pytest.mark.parametrize('cfg,E', [(A, 4096), (D, 1024), (Ecfg, 4096)], ids=['B_4096', 'D_1024_per_gpu', 'E_4096'])
def test_full_size_invariants(cfg, E):
    """BASELINE.json batch sizes: properties that need no oracle."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    v = VecDMFB(n_envs=E, seed=9, **cfg)
    n, fov = cfg['n_agents'], cfg['fov']
    ff = fov * fov
    g = torch.Generator(device='cuda').manual_seed(1)
    v.reset()
    s0, e0 = v.get_task()
    pts = torch.cat([s0, e0], dim=1).long()
    d2 = ((pts[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1) + torch.eye(2 * n, device='cuda', dtype=torch.long) * 99
    assert int(d2.min()) > 2                       # _Generate_Start_End acceptance rule (dmfb.py:220)
    prev = v.get_state()
    for t in range(60):
        a = torch.randint(0, 5, (E, n), device='cuda', generator=g, dtype=torch.int64)
        obs, r, d, info = v.step(a, autoreset=True)
        st = v.get_state()
        pos = st['pos'].long()
        key = pos[..., 0] * 256 + pos[..., 1]
        srt = key.sort(dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())             # no two droplets share a cell (dmfb.py:341-343)
        assert bool((pos[..., 0] >= 0).all() and (pos[..., 0] < cfg['width']).all())
        assert bool((pos[..., 1] >= 0).all() and (pos[..., 1] < cfg['length']).all())
        o = obs.long()
        idx = torch.arange(1, n + 1, device='cuda')[None, :].expand(E, n)
        assert bool((o[:, :, (fov // 2) * fov + fov // 2] == idx).all())   # self at the window centre
        moved = (pos - prev['pos'].long()).abs().sum(-1)
        keep = info['terminated'] == 0
        assert bool((moved[keep] <= 1).all())                       # one cell per step at most
        was_done = (prev['dist'] == 0) & keep[:, None]
        assert bool((moved[was_done] == 0).all())                   # finished droplets stay (stall, dmfb.py:331)
        assert bool((o[:, :, 2 * ff:3 * ff] <= 1).all() and (o[:, :, :2 * ff] <= n).all() and (o >= -10).all())
        prev = st


def test_lockstep_split_launch_path(monkeypatch):
    """Large batches use a step-only launch (all four waves stepping) + the observation kernel;
    DMFB_VEC_SPLIT_MIN_ENVS=1 forces that path at test sizes.  Must be bit-identical too."""
    monkeypatch.setenv('DMFB_VEC_SPLIT_MIN_ENVS', '1')
    assert _lockstep(A, E=1500, steps=100, seed=51, autoreset=True) > 1000
    _lockstep(A, E=300, steps=60, seed=52, autoreset=False)
    _lockstep(D, E=300, steps=230, seed=53, autoreset=True, greedy=0.9)
    _lockstep(Ecfg, E=333, steps=150, seed=54, autoreset=True, greedy=0.8)
    _lockstep(dict(width=12, length=9, n_agents=3, fov=7, with_maps=True), E=100, steps=60, seed=55, autoreset=True)


def test_lockstep_lane_per_droplet_kernel(monkeypatch):
    """The opt-in lane-per-droplet transition (csrc/dmfb_step_lanes.h, DMFB_VEC_LANES=1: 16 lanes per chip, DPP row broadcasts,
    wave ballots for the clash test; n >= 8, step-only launches) is bit-identical to the oracle too.  It is NOT the default: it
    measured slower than the lane-per-chip kernel."""
    monkeypatch.setenv('DMFB_VEC_SPLIT_MIN_ENVS', '1')
    monkeypatch.setenv('DMFB_VEC_LANES', '1')
    _lockstep(D, E=300, steps=230, seed=53, autoreset=True, greedy=0.9)
    _lockstep(Ecfg, E=333, steps=150, seed=54, autoreset=True, greedy=0.8)
    _lockstep(Ecfg, E=50, steps=100, seed=32, autoreset=False, greedy=0.8)
    _lockstep(dict(Ecfg, n_blocks=8), E=96, steps=120, seed=64, autoreset=True, greedy=0.8)
    _lockstep(dict(width=40, length=40, n_agents=16, fov=9), E=70, steps=180, seed=56, autoreset=True, greedy=0.9)
    _lockstep(dict(width=20, length=20, n_agents=8, fov=7, stall=False), E=90, steps=90, seed=57, autoreset=True)


def test_lockstep_with_obstacle_blocks():
    """GenRandomBlocks (dmfb.py:228-251) through the Philox contract, _isTouchingBlocks and the block
    layer of the observation, fused and split launch shapes."""
    assert _lockstep(dict(A, n_blocks=4), E=600, steps=120, seed=61, autoreset=True) > 300
    _lockstep(dict(A, n_blocks=5), E=77, steps=90, seed=62, autoreset=False)
    _lockstep(dict(width=20, length=20, n_agents=6, fov=9, n_blocks=14), E=150, steps=170, seed=63, autoreset=True, greedy=0.85)
    _lockstep(dict(Ecfg, n_blocks=8), E=96, steps=120, seed=64, autoreset=True, greedy=0.8)


def test_lockstep_blocks_split_launch(monkeypatch):
    monkeypatch.setenv('DMFB_VEC_SPLIT_MIN_ENVS', '1')
    _lockstep(dict(A, n_blocks=4), E=300, steps=90, seed=65, autoreset=True)
    _lockstep(dict(width=20, length=20, n_agents=6, fov=9, n_blocks=14), E=100, steps=100, seed=66, autoreset=False)


def test_block_density_rule_matches_reference():
    # GenRandomBlocks returns without blocks above 20 % coverage (dmfb.py:232-234): 6 blocks on 10x10 = 24 %
    v = _vec(width=10, length=10, n_agents=2, fov=5, n_blocks=6, n_envs=4)
    o = DmfbOracle(10, 10, 2, n_blocks=6, fov=5, n_envs=4)
    assert v.get_blocks().shape[1] == 0 and o.get_blocks().shape[1] == 0


def test_unaligned_obs_buffer_and_tiny_tiles(monkeypatch):
    """The obs tile is phase-aligned in LDS to its HBM destination, so any tile size and any (even odd)
    destination address must give the same bytes."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    ref = None
    for tile in ('1', '2', '4', '16', '64'):
        monkeypatch.setenv('DMFB_VEC_MIN_TILE', tile)
        v = VecDMFB(n_envs=333, seed=77, **D)
        v.reset()
        n, L = D['n_agents'], v.obs_len
        big = torch.zeros(333 * n * L + 64, dtype=torch.int8, device='cuda')
        for off in (0, 3, 8, 13):
            view = big[off:off + 333 * n * L].view(333, n, L)
            v.observe(obs=view)
            got = view.cpu().numpy().copy()
            if ref is None:
                ref = got
            np.testing.assert_array_equal(got, ref, err_msg='tile %s offset %d' % (tile, off))
        a = torch.randint(0, 5, (333, n), device='cuda')
        v.step(a, autoreset=True)          # fused launch with the same tile
        torch.cuda.synchronize()
    o = DmfbOracle(n_envs=333, seed=77, **D)
    o.reset()
    np.testing.assert_array_equal(ref, o.observe())


def test_reset_new_and_masked_restart():
    """reset(new=True) re-initialises the maps and draws a NEW degradation map (dmfb.py:178-181);
    restart(mask) puts the masked chips back on their starts (dmfb.py:599-605)."""
    rng = np.random.default_rng(9)
    O = DmfbOracle(n_envs=24, seed=5, **Ecfg)
    V = _vec(n_envs=24, seed=5, **Ecfg)
    d0 = V.get_map('degrade')
    np.testing.assert_array_equal(_bits(O.get_map('degrade')), _bits(d0))
    h = rng.random((24, 20, 20)) * 0.5 + 0.5
    for B in (O, V):
        B.set_map('health', h)
        B.set_map('usage', np.full((24, 20, 20), 60.0))
        B.reset(new=True)
    for m in ('health', 'usage', 'degrade'):
        np.testing.assert_array_equal(_bits(O.get_map(m)), _bits(V.get_map(m)), err_msg=m)
    assert np.all(V.get_map('health') == 1.0) and np.all(V.get_map('usage') == 0.0)
    assert not np.array_equal(V.get_map('degrade'), d0)
    for a, b in zip(O.get_task(), V.get_task()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(O.observe(), V.observe())
    for t in range(12):
        a = rng.integers(0, 5, (24, 10)).astype(np.int32)
        O.step(a); V.step(a)
    mask = (rng.random(24) < 0.5).astype(np.uint8)
    O.restart(mask=mask); V.restart(mask=mask)
    so, sv = O.get_state(), V.get_state()
    for k in ('pos', 'dist', 'step_count', 'constraints'):
        np.testing.assert_array_equal(so[k], sv[k], err_msg=k)
    np.testing.assert_array_equal(O.observe(), V.observe())
    starts, _ = V.get_task()
    np.testing.assert_array_equal(sv['pos'][mask == 1], starts[mask == 1])


def _usage_log_walk(cfg, E, steps, seed, restart_every, check_every, set_usage_at=None, reset_every=None):
    """addUsage goes through a per-chip log on the GPU (include/dmfb_vec.h): the maps must agree with the
    oracle's at every point where they are read, also when the log fills up (restart() loops never reset it), when the
    usage map is read or replaced in the middle of an episode, and across reset(new=False) (updateHealth)."""
    O = DmfbOracle(n_envs=E, seed=seed, **cfg)
    V = _vec(n_envs=E, seed=seed, **cfg)
    rng = np.random.default_rng(seed)
    n = cfg['n_agents']
    u0 = rng.integers(30, 50, (E, cfg['width'], cfg['length'])).astype(np.float64)
    for B in (O, V):
        B.set_map('usage', u0)
    for t in range(steps):
        a = rng.integers(0, 5, (E, n)).astype(np.int32)
        ro, do, co, so = O.step(a)
        rv, dv, cv, sv = V.step(a)
        np.testing.assert_array_equal(_bits(ro), _bits(rv), err_msg='rewards t=%d' % t)
        if restart_every and (t + 1) % restart_every == 0:   # restart(): counters and droplets back, maps (and the log) untouched
            m = (rng.random(E) < 0.7).astype(np.uint8)
            O.restart(mask=m); V.restart(mask=m)
        if reset_every and (t + 1) % reset_every == 0:       # reset(new=False): log folded in, then updateHealth
            m = (rng.random(E) < 0.5).astype(np.uint8)
            O.reset(mask=m); V.reset(mask=m)
            np.testing.assert_array_equal(_bits(O.get_map('health')), _bits(V.get_map('health')), err_msg='health t=%d' % t)
        if set_usage_at is not None and t == set_usage_at:
            u1 = rng.integers(0, 60, (E, cfg['width'], cfg['length'])).astype(np.float64)
            O.set_map('usage', u1); V.set_map('usage', u1)
        if (t + 1) % check_every == 0:
            np.testing.assert_array_equal(O.get_map('usage'), V.get_map('usage'), err_msg='usage t=%d' % t)
    np.testing.assert_array_equal(O.get_map('usage'), V.get_map('usage'))
    np.testing.assert_array_equal(_bits(O.get_map('health')), _bits(V.get_map('health')))


def test_usage_log_fills_up_without_reset():
    # max_step = 2*(12+9) = 42 log slots; 150 steps with restart() only: the log is folded in every time it is full
    _usage_log_walk(dict(width=12, length=9, n_agents=3, fov=7, with_maps=True), E=70, steps=150, seed=5, restart_every=11,
                    check_every=150)


def test_usage_map_read_and_replaced_mid_episode():
    _usage_log_walk(dict(width=10, length=10, n_agents=4, fov=9, b_degrade=True, per_degrade=1.0), E=33, steps=120, seed=6,
                    restart_every=0, check_every=7, set_usage_at=31, reset_every=25)


def test_usage_log_large_chip_global_atomic_path():
    # 80 x 70 = 5600 cells: above the LDS-histogram limit, the log is folded in with 32-bit global atomics on the u16 pairs
    _usage_log_walk(dict(width=80, length=70, n_agents=5, fov=5, b_degrade=True, per_degrade=0.5), E=9, steps=330, seed=7,
                    restart_every=0, check_every=100, reset_every=60)


def test_lockstep_E_generator_maps_compact_health_path():
    """Maps never replaced from outside: the transition rebuilds a cell's health from its degrade count (DevPtrs::kmap) and
    the Philox degrade factor; transitions and all three maps must still match the oracle, which keeps plain float64 maps."""
    assert _lockstep(Ecfg, E=160, steps=400, seed=33, autoreset=True, greedy=0.8, explicit_health=False) > 100
    _lockstep(dict(width=12, length=9, n_agents=3, fov=7, b_degrade=True, per_degrade=0.6), E=90, steps=200, seed=34, autoreset=False,
              greedy=0.7, explicit_health=False)


def test_degrade_count_saturation():
    """The 4-bit degrade counts saturate at 15: up to 14 degradations of a cell the transition rebuilds its health from
    the count, from 15 on it falls back to the float64 map.  Parity is checked on both sides of the boundary and far
    beyond it (262 degradations), with odd cell counts too (two cells share a byte, eight a word)."""
    for cfg, E, seed in ((dict(width=10, length=10, n_agents=4, fov=9, b_degrade=True, per_degrade=1.0), 12, 35),
                         (dict(width=9, length=7, n_agents=3, fov=5, b_degrade=True, per_degrade=0.8), 7, 36)):
        O = DmfbOracle(n_envs=E, seed=seed, **cfg)
        V = _vec(n_envs=E, seed=seed, **cfg)
        n = cfg['n_agents']
        rng = np.random.default_rng(seed)
        over = np.full((E, cfg['width'], cfg['length']), 60.0)
        over[:, ::2, 1::3] = 10.0        # a third of the cells is not degraded: neighbouring nibbles differ
        done = 0
        for target in (13, 14, 15, 16, 17, 262):
            while done < target:         # every reset degrades the marked cells once (usage 60 > 50)
                for B in (O, V):
                    B.set_map('usage', over)
                    B.reset()
                done += 1
            np.testing.assert_array_equal(_bits(O.get_map('health')), _bits(V.get_map('health')), err_msg='health after %d' % done)
            for t in range(25):
                a = rng.integers(0, 5, (E, n)).astype(np.int32)
                ro, do, co, so = O.step(a)
                rv, dv, cv, sv = V.step(a)
                np.testing.assert_array_equal(_bits(ro), _bits(rv), err_msg='rewards after %d t=%d' % (done, t))
                np.testing.assert_array_equal(O.get_state()['pos'], V.get_state()['pos'], err_msg='pos after %d t=%d' % (done, t))
            np.testing.assert_array_equal(O.observe(), V.observe())
        assert O.get_map('health').max() == 1.0 and O.get_map('health').min() < 0.7 ** 15


@pytest.mark.parametrize('cfg,E', [(A, 150001), (D, 40003)], ids=['A', 'D'])
def test_observe_persistent_multi_tile_and_masks(cfg, E):
    """Batches where every workgroup of the persistent observation kernel walks many tiles (records prefetched two
    tiles ahead; positions and refresh flags double-buffered), whole and under partial masks: rows of masked-off
    chips keep what the buffer held."""
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from oracle.dmfb_oracle import DmfbOracle
    V = VecDMFB(n_envs=E, seed=17, **cfg)
    O = DmfbOracle(n_envs=E, seed=17, **cfg)
    V.reset(); O.reset()
    sh = V.launch_shape()
    assert (E + sh['observe_tile'] - 1) // sh['observe_tile'] > 2 * sh['observe_workgroups']
    rng = np.random.default_rng(3)
    n = cfg['n_agents']
    for t in range(3):
        a = rng.integers(0, 5, (E, n)).astype(np.int32)
        obs, r, d, info = V.step(torch.as_tensor(a, device='cuda'), autoreset=False)
        O.step(a)
        assert np.array_equal(obs.cpu().numpy(), O.observe()), 'obs differ at step %d' % t
    want_full = O.observe()
    for frac in (0.5, 0.03, 0.97):
        mask = (rng.random(E) < frac).astype(np.uint8)
        mask[:200] = 0; mask[-300:-100] = 1
        buf = torch.full_like(V.obs, 77)
        V.observe(mask=torch.as_tensor(mask, device='cuda'), obs=buf)
        want = want_full.copy()
        want[mask == 0] = 77
        assert np.array_equal(buf.cpu().numpy(), want), 'masked observe differs at frac %.2f' % frac


def test_lockstep_maximum_sizes():
    """The limits of include/dmfb_vec.h: 255 x 255 cells (positions packed in bytes, direction zoom over +-254, max_step
    1020 = the usage-log length field nearly full), 16 droplets, maps on the global-atomic fold path; and a 1-droplet chip."""
    _lockstep(dict(width=255, length=255, n_agents=16, fov=9, b_degrade=True, per_degrade=0.9), E=5, steps=40, seed=51,
              autoreset=True, greedy=0.9)
    _lockstep(dict(width=255, length=17, n_agents=16, fov=13, with_maps=True), E=3, steps=600, seed=52, autoreset=True, greedy=0.95)
    _lockstep(dict(width=5, length=5, n_agents=1, fov=5), E=130, steps=40, seed=53, autoreset=True)
