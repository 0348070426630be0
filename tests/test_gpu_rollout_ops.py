"""rollout_select_actions / rollout_post_step (include/rollout_ops.h) against the op-by-op tensor formulation of
the reference's per-step book-keeping (common/rollout.py:101-150, agent/agent.py:41-45)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _philox(k0, k1, c):
    M0, M1 = 0xD2511F53, 0xCD9E8D57
    c = [int(x) for x in c]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


@pytest.mark.parametrize('evaluate', [True, False])
def test_select_actions(evaluate):
    from marl_dmfb_amd import _lib
    lib = _lib.rollout_ops()
    E, n, A, T, t = 37, 3, 5, 6, 4
    g = torch.Generator(device='cuda').manual_seed(5)
    q = torch.randn(E * n, A, device='cuda', generator=g)
    q[3] = q[3, 0]  # ties -> first maximum
    eps = torch.tensor([0.4], device='cuda')
    draw = torch.tensor([7], dtype=torch.int32, device='cuda')
    actions = torch.full((E, n), -1, dtype=torch.int32, device='cuda')
    last = torch.full((E, n, A), 9, dtype=torch.int8, device='cuda')
    ep_u = torch.zeros((E, T, n, 1), dtype=torch.int8, device='cuda')
    ep_oh = torch.zeros((E, T, n, A), dtype=torch.int8, device='cuda')
    seed = 0x1234567811223344
    vp = C.c_void_p
    rc = lib.rollout_select_actions(vp(q.data_ptr()), E, n, A, vp(eps.data_ptr()), int(evaluate), seed, vp(draw.data_ptr()),
                                    vp(actions.data_ptr()), vp(last.data_ptr()), vp(ep_u.data_ptr()), vp(ep_oh.data_ptr()), T, t, None)
    assert rc == 0
    torch.cuda.synchronize()
    qn = q.cpu().numpy()
    want = np.zeros(E * n, np.int64)
    n_rand = 0
    for r in range(E * n):
        a = int(np.argmax(qn[r]))
        if not evaluate:
            w = _philox(seed & 0xFFFFFFFF, seed >> 32, (r, 7, 0, 0x600))
            if np.float32(w[0] >> 8) * np.float32(2.0 ** -24) < np.float32(0.4):
                a = (w[1] * A) >> 32
                n_rand += 1
        want[r] = a
    if not evaluate:
        assert 0.25 * E * n < n_rand < 0.55 * E * n
    assert np.array_equal(actions.cpu().numpy().reshape(-1), want)
    oh = np.eye(A, dtype=np.int8)[want].reshape(E, n, A)
    assert np.array_equal(last.cpu().numpy(), oh)
    assert np.array_equal(ep_u[:, t, :, 0].cpu().numpy(), want.reshape(E, n))
    assert np.array_equal(ep_oh[:, t].cpu().numpy(), oh)
    assert int(ep_u.abs().sum()) == int(np.abs(want).sum()) and int(ep_oh.sum()) == E * n   # other slots untouched


@pytest.mark.parametrize('f64', [False, True])
def test_post_step(f64):
    from marl_dmfb_amd import _lib
    lib = _lib.rollout_ops()
    E, T, t = 2500, 7, 3
    g = torch.Generator(device='cuda').manual_seed(1)
    alive = (torch.rand(E, device='cuda', generator=g) < 0.7).to(torch.uint8)
    term = (torch.rand(E, device='cuda', generator=g) < 0.3).to(torch.uint8)
    tr = torch.randn(E, device='cuda', generator=g, dtype=torch.float64)
    cons = torch.randint(0, 4, (E,), device='cuda', generator=g, dtype=torch.int32)
    cons_in = cons.double() * 0.6 if f64 else cons
    succ = (torch.rand(E, device='cuda', generator=g) < 0.2).to(torch.uint8)
    ep_r = torch.zeros((E, T, 1), device='cuda')
    ep_pad = torch.ones((E, T, 1), dtype=torch.bool, device='cuda')
    ep_term = torch.ones((E, T, 1), dtype=torch.bool, device='cuda')
    s_r = torch.randn(E, device='cuda', generator=g, dtype=torch.float64)
    s_c = torch.zeros(E, dtype=torch.float64, device='cuda')
    s_s = torch.zeros(E, dtype=torch.int64, device='cuda')
    steps = torch.arange(E, device='cuda')
    eps = torch.tensor([0.9], device='cuda')
    n_alive = torch.zeros(4, dtype=torch.int32, device='cuda')
    draw = torch.tensor([41], dtype=torch.int32, device='cuda')
    row = 4 * 245 if not f64 else 3 * 245   # dword path / byte path
    obs = torch.randint(-3, 9, (E, row), dtype=torch.int8, device='cuda', generator=g)
    ep_o = torch.zeros((E, T, row), dtype=torch.int8, device='cuda')
    ep_on = torch.zeros((E, T, row), dtype=torch.int8, device='cuda')
    a0, r0, st0 = alive.clone(), s_r.clone(), steps.clone()
    vp = C.c_void_p
    rc = lib.rollout_post_step(E, T, t, vp(alive.data_ptr()), vp(term.data_ptr()), vp(tr.data_ptr()), vp(cons_in.data_ptr()), int(f64),
                               vp(succ.data_ptr()), vp(ep_r.data_ptr()), vp(ep_pad.data_ptr()), vp(ep_term.data_ptr()),
                               vp(s_r.data_ptr()), vp(s_c.data_ptr()), vp(s_s.data_ptr()), vp(steps.data_ptr()), vp(eps.data_ptr()),
                               1e-4, 0.05, vp(n_alive.data_ptr()), vp(draw.data_ptr()), vp(obs.data_ptr()), row, vp(ep_o.data_ptr()),
                               vp(ep_on.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(ep_r[:, t, 0], tr.float())
    assert float(ep_r[:, :t].abs().sum()) == 0.0 and float(ep_r[:, t + 1:].abs().sum()) == 0.0
    assert torch.equal(ep_pad[:, t, 0], a0 == 0) and torch.equal(ep_term[:, t, 0], term != 0)
    assert bool(ep_pad[:, :t].all()) and bool(ep_term[:, t + 1:].all())
    assert torch.equal(s_r, r0 + tr) and torch.equal(s_c, cons_in.double()) and torch.equal(s_s, succ.long())
    assert torch.equal(steps, st0 + a0.long())
    assert torch.equal(alive, a0 & (1 - term))
    assert torch.equal(ep_on[:, t], obs * a0.view(E, 1).to(torch.int8))
    assert torch.equal(ep_o[:, t + 1], obs * (a0 & (1 - term)).view(E, 1).to(torch.int8))
    assert int(ep_on[:, :t].abs().sum()) == 0 and int(ep_on[:, t + 1:].abs().sum()) == 0
    assert int(ep_o[:, :t + 1].abs().sum()) == 0 and int(ep_o[:, t + 2:].abs().sum()) == 0
    assert n_alive.tolist() == [int(alive.sum()), 0, 0, 0] and int(draw) == 42
    want_eps = max(np.float32(0.9) - np.float32(1e-4) * np.float32(int(a0.sum())), np.float32(0.05))
    assert abs(float(eps) - float(want_eps)) < 1e-6


@pytest.mark.parametrize('A,H', [(5, 128), (9, 128)])
def test_gru_head_select_matches_grucell_and_linear(A, H):
    """rollout_gru_head_select against nn.GRUCell + nn.Linear + argmax on the same projections (network/base_net.py:69-70,
    agent/agent.py:45).  fp32 tolerances: h 1e-6 absolute, q 1e-5; the action must be the argmax of the kernel's own q."""
    from marl_dmfb_amd import _lib
    lib = _lib.rollout_ops()
    E, n, F = 33, 4, 70
    R = E * n
    torch.manual_seed(A)
    cell = torch.nn.GRUCell(F, H).cuda()
    fc = torch.nn.Linear(H, A).cuda()
    x = torch.randn(R, F, device='cuda')
    h0 = torch.rand(R, H, device='cuda') * 2 - 1
    with torch.no_grad():
        h_ref = cell(x, h0)
        q_ref = fc(h_ref)
        ig = x @ cell.weight_ih.t()
        hg = h0 @ cell.weight_hh.t()
    h = h0.clone()
    actions = torch.full((R,), -1, dtype=torch.int32, device='cuda')
    last = torch.zeros((R, A), dtype=torch.int8, device='cuda')
    q = torch.zeros((R, A), device='cuda')
    vp = C.c_void_p
    rc = lib.rollout_gru_head_select(vp(ig.data_ptr()), vp(hg.data_ptr()), vp(cell.bias_ih.data_ptr()), vp(cell.bias_hh.data_ptr()),
                                     vp(h.data_ptr()), vp(fc.weight.data_ptr()), vp(fc.bias.data_ptr()), E, n, H, A, None, 1, 0, None,
                                     vp(actions.data_ptr()), vp(last.data_ptr()), None, None, 1, 0, vp(q.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(q.cpu().numpy(), q_ref.cpu().numpy(), rtol=0, atol=1e-5)
    assert torch.equal(actions.long(), q.argmax(dim=1))
    assert torch.equal(last, torch.nn.functional.one_hot(actions.long(), A).to(torch.int8))


def test_compact_alive_lists_live_chips_in_order():
    from marl_dmfb_amd import _lib
    lib = _lib.rollout_ops()
    vp = C.c_void_p
    for E, p in ((1, 1.0), (37, 0.5), (1024, 0.3), (4096, 0.65), (5000, 0.0), (9001, 0.9)):
        g = torch.Generator(device='cuda').manual_seed(E)
        alive = (torch.rand(E, device='cuda', generator=g) < p).to(torch.uint8)
        lst = torch.full((E,), -7, dtype=torch.int32, device='cuda')
        cnt = torch.full((1,), -1, dtype=torch.int32, device='cuda')
        assert lib.rollout_compact_alive(E, vp(alive.data_ptr()), vp(lst.data_ptr()), vp(cnt.data_ptr()), None) == 0
        want = torch.nonzero(alive).reshape(-1).to(torch.int32)
        k = int(cnt.item())
        assert k == want.numel()
        assert torch.equal(lst[:k], want) and bool((lst[k:] == -7).all())


@pytest.mark.parametrize('n', [1, 2, 3, 4, 10])
def test_live_row_kernels_match_the_full_ones(n):
    """crnn_front9_forward_live / rollout_gru_head_select_live (the rollout's kernels for the chips still playing) against the
    full-batch kernels: same values, compact x rows, finished chips untouched.  n = 1: one row per chip, where the kernel's
    multiply-high row / rows_per_chip must degenerate to the identity."""
    import types
    from marl_dmfb_amd import _lib
    from marl_dmfb_amd.network.base_net import CRNN
    lib = _lib.rollout_ops()
    vp = C.c_void_p
    E, A, H, T, t = 203, 5, 128, 7, 3
    torch.manual_seed(4)
    net = CRNN(types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=24, rnn_hidden_dim=H, n_actions=A, fov=9)).cuda()
    g = torch.Generator(device='cuda').manual_seed(1)
    obs = torch.randint(0, 5, (E * n, 245), device='cuda', generator=g, dtype=torch.int8)
    la = torch.zeros((E * n, A), dtype=torch.int8, device='cuda')
    la[torch.arange(E * n), torch.randint(0, A, (E * n,), device='cuda', generator=g)] = 1
    alive = (torch.rand(E, device='cuda', generator=g) < 0.6).to(torch.uint8)
    lst = torch.empty(E, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert lib.rollout_compact_alive(E, vp(alive.data_ptr()), vp(lst.data_ptr()), vp(cnt.data_ptr()), None) == 0
    k = int(cnt.item())
    rows = (lst[:k].long()[:, None] * n + torch.arange(n, device='cuda')[None]).reshape(-1)
    with torch.no_grad():
        x_full = net._front_features_hip(obs, la, padded=True)
        x_live = torch.full_like(x_full, 123.0)
        net.front_features_live(obs, la, lst, cnt, n, x_live)
    assert torch.equal(x_live[:k * n], x_full[rows])
    assert bool((x_live[k * n:] == 123.0).all())                      # rows beyond the live ones are not written
    # head + pick: ig compact, everything else in chip order
    ig_full = torch.randn(E * n, 3 * H, device='cuda', generator=g)
    hg = torch.randn(E * n, 3 * H, device='cuda', generator=g)
    h0 = torch.randn(E * n, H, device='cuda', generator=g)
    ig_live = torch.zeros_like(ig_full)
    ig_live[:k * n] = ig_full[rows]
    eps = torch.tensor([0.3], device='cuda')
    draw = torch.tensor([5], dtype=torch.int32, device='cuda')
    outs = []
    for live in (False, True):
        h = h0.clone()
        actions = torch.full((E * n,), -1, dtype=torch.int32, device='cuda')
        last = torch.full((E * n, A), 9, dtype=torch.int8, device='cuda')
        ep_u = torch.zeros((E, T, n, 1), dtype=torch.int8, device='cuda')
        ep_oh = torch.zeros((E, T, n, A), dtype=torch.int8, device='cuda')
        q = torch.zeros((E * n, A), device='cuda')
        args = [vp(net.rnn.bias_ih.data_ptr()), vp(net.rnn.bias_hh.data_ptr()), vp(h.data_ptr()), vp(net.fc1.weight.data_ptr()),
                vp(net.fc1.bias.data_ptr()), E, n, H, A, vp(eps.data_ptr()), 0, 77, vp(draw.data_ptr()), vp(actions.data_ptr()),
                vp(last.data_ptr()), vp(ep_u.data_ptr()), vp(ep_oh.data_ptr()), T, t, vp(q.data_ptr())]
        if live:
            rc = lib.rollout_gru_head_select_live(vp(ig_live.data_ptr()), vp(hg.data_ptr()), *args, vp(lst.data_ptr()), vp(cnt.data_ptr()), None)
        else:
            rc = lib.rollout_gru_head_select(vp(ig_full.data_ptr()), vp(hg.data_ptr()), *args, None)
        assert rc == 0
        outs.append((h, actions, last, ep_u, ep_oh, q))
    full, lv = outs
    dead = torch.ones(E * n, dtype=torch.bool, device='cuda')
    dead[rows] = False
    for a, b in zip(full, lv):
        if a.dim() == 4:   # episode tensors (E, T, n, .): slot t of the live rows; finished chips stay zero
            a2, b2 = a[:, t].reshape(E * n, -1), b[:, t].reshape(E * n, -1)
            assert int(b2[dead].abs().sum()) == 0 and int(b.abs().sum()) == int(b[:, t].abs().sum())
        else:
            a2, b2 = a.reshape(E * n, -1), b.reshape(E * n, -1)
        assert torch.equal(a2[rows], b2[rows])
    assert torch.equal(lv[0][dead], h0[dead])                          # hidden state of finished chips untouched
    assert bool((lv[1][dead] == -1).all()) and bool((lv[2][dead] == 9).all())
