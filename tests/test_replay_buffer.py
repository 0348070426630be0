"""ReplayBuffer ring insertion / sampling (reference common/replay_buffer.py:32-75) on the CPU device: the slice fast path
and the indexed path must both reproduce the reference's three-branch wrap rule."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from marl_dmfb_amd.common.replay_buffer import ReplayBuffer

REF = '/root/reference'


def _args(size=10):
    return types.SimpleNamespace(n_actions=5, n_agents=2, obs_shape=(3, 1, 1, 2, 7), buffer_size=size, episode_limit=3,
                                 device='cpu', cuda=False, alg='vdn')


def _batch(first_id, count, a):
    T, n, O, A = a.episode_limit, a.n_agents, a.obs_shape[-1], a.n_actions
    ids = np.arange(first_id, first_id + count)
    f = lambda shape, dt: (np.broadcast_to(ids.reshape((-1,) + (1,) * (len(shape) - 1)) % 100, shape).astype(dt))
    return {'o': f((count, T, n, O), np.int8), 'u': f((count, T, n, 1), np.int8), 'r': f((count, T, 1), np.float32),
            'o_next': f((count, T, n, O), np.int8), 'avail_u': f((count, T, n, A), np.int8),
            'avail_u_next': f((count, T, n, A), np.int8), 'u_onehot': f((count, T, n, A), np.int8),
            'padded': f((count, T, 1), np.int8) % 2 == 1, 'terminated': f((count, T, 1), np.int8) % 2 == 0}


def _rule(cur, size, inc):
    """common/replay_buffer.py:58-75 restated: (slot indices, new current_idx)."""
    if cur + inc <= size:
        return list(range(cur, cur + inc)), cur + inc
    if cur < size:
        overflow = inc - (size - cur)
        return list(range(cur, size)) + list(range(overflow)), overflow
    return list(range(inc)), inc


INCS = [3, 4, 3, 2, 9, 1, 10, 5, 5, 7, 6]   # exact fill, restart-at-0 branch, wraps, a full-size batch


def test_ring_rule_and_contents():
    a = _args()
    buf = ReplayBuffer(a, device='cpu')
    for v in buf.buffers.values():
        v.zero_()                       # the buffers are torch.empty like the reference's np.empty
    model = {k: np.zeros(tuple(v.shape), dtype=v.numpy().dtype) for k, v in buf.buffers.items()}
    cur, filled, eid = 0, 0, 0
    for inc in INCS:
        batch = _batch(eid, inc, a)
        eid += inc
        idx, cur = _rule(cur, a.buffer_size, inc)
        filled = min(a.buffer_size, filled + inc)
        for k in model:
            model[k][idx] = batch[k]
        buf.store_episode({k: torch.as_tensor(v) for k, v in batch.items()})
        assert (buf.current_idx, buf.current_size) == (cur, filled)
        for k in model:
            assert np.array_equal(buf.buffers[k].numpy(), model[k]), (k, inc)
    with pytest.raises(ValueError):
        buf.store_episode({k: torch.as_tensor(v) for k, v in _batch(0, a.buffer_size + 1, a).items()})


def test_sample_shapes_dtypes_and_range():
    a = _args()
    buf = ReplayBuffer(a, device='cpu')
    buf.store_episode({k: torch.as_tensor(v) for k, v in _batch(0, 4, a).items()})
    buf.generator = torch.Generator().manual_seed(0)
    s = buf.sample(64)
    assert s['o'].shape == (64, 3, 2, 7) and s['o'].dtype == torch.int8 and s['r'].dtype == torch.float32
    assert s['padded'].dtype == torch.bool and s['u'].shape == (64, 3, 2, 1)
    ids = s['u'][:, 0, 0, 0].numpy()
    assert set(ids.tolist()) <= {0, 1, 2, 3} and len(set(ids.tolist())) == 4   # only filled slots, with replacement


@pytest.mark.skipif(not os.path.isdir(REF), reason='live reference not present (GPU box)')
def test_against_live_reference_class():
    sys.path.insert(0, REF)
    try:
        from common.replay_buffer import ReplayBuffer as RefBuffer
    finally:
        sys.path.remove(REF)
    a = _args()
    mine, ref = ReplayBuffer(a, device='cpu'), RefBuffer(a)
    for k in ref.buffers:
        ref.buffers[k][...] = 0
    for k, v in mine.buffers.items():
        v.zero_()
    eid = 0
    for inc in INCS:
        batch = _batch(eid, inc, a)
        eid += inc
        ref.store_episode(batch)
        mine.store_episode({k: torch.as_tensor(v) for k, v in batch.items()})
        assert (mine.current_idx, mine.current_size) == (ref.current_idx, ref.current_size)
        for k in ref.buffers:
            assert np.array_equal(mine.buffers[k].numpy().astype(ref.buffers[k].dtype), ref.buffers[k]), (k, inc)


def test_ring_bookkeeping_host_mirror_and_length_sorted_draw():
    """The device-side ring bookkeeping the continuous rollout writes (include/rollout_ops.h: rollout_ring) is kept in step by
    store_episode too; sync_host mirrors it in ONE read; draw picks filled slots uniformly with replacement
    (common/replay_buffer.py:53) and hands them over sorted by episode length, longest first, with their lengths."""
    a = _args(size=12)
    a.seed = 3
    buf = ReplayBuffer(a)
    T = a.episode_limit
    ep = _batch(0, 7, a)
    lens = np.array([3, 1, 2, 3, 1, 1, 2])
    ep['padded'] = np.arange(T)[None, :, None] >= lens[:, None, None]
    ep['terminated'] = np.arange(T)[None, :, None] >= lens[:, None, None] - 1
    buf.store_episode(ep)
    extra = buf.sync_host(torch.tensor([5, 6, 7]))
    assert extra == [5, 6, 7]
    assert (buf.current_idx, buf.current_size) == (7, 7)
    assert buf.host_len[:7].tolist() == lens.tolist() and not buf.host_len[7:].any()
    assert buf.ring_state[:2].tolist() == [7, 7]
    seen = set()
    for _ in range(40):
        idx, ln = buf.draw(5)
        assert idx.shape == ln.shape == (5,) and (idx < 7).all() and (idx >= 0).all()
        assert ln.tolist() == lens[idx].tolist() and (np.diff(ln) <= 0).all()      # longest first
        seen.update(idx.tolist())
        got = buf.gather(idx)
        for k, v in got.items():
            assert torch.equal(v, buf.buffers[k][torch.as_tensor(idx)])
    assert seen == set(range(7))                                                     # every filled slot is reachable, no empty one
    # the ring wraps: lengths of the overwritten slots follow
    ep2 = _batch(7, 8, a)
    ep2['padded'] = np.zeros((8, T, 1), bool)
    buf.store_episode(ep2)
    buf.sync_host()
    assert buf.current_size == 12 and buf.host_len.tolist() == [3, 3, 3] + lens[3:].tolist() + [3] * 5


def test_pack_units_lists_the_valid_steps_step_after_step():
    """VDN.pack_units (host side of learn_packed): counts[t] = episodes longer than t; units = slot * T + t of those episodes,
    step 0 first -- the layout of gru_seq_forward_packed / vdn_td_forward_packed (include/crnn_ops.h, include/vdn_ops.h)."""
    from marl_dmfb_amd.policy.vdn import VDN
    idx = np.array([9, 2, 5, 7])          # sorted by length, longest first
    lens = np.array([4, 2, 2, 1])
    counts, units = VDN.pack_units(idx, lens, 10)
    assert counts.tolist() == [4, 3, 1, 1]
    assert units.dtype == np.int32
    assert units.tolist() == [90, 20, 50, 70, 91, 21, 51, 92, 93]
    assert len(units) == lens.sum()
