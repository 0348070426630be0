"""The hand-written HIP conv front end (include/crnn_ops.h) against a plain PyTorch fp32 reference of
the same op (conv2d+ReLU twice, network/base_net.py:63-65).  Floating point: rtol 1e-5, atol 1e-5
(same products, different summation order)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _safe_rows(net, obs, margin=2e-5):
    """Rows none of whose conv pre-activations lies within `margin` of zero (float64 forward of the same weights).
    A pre-activation at rounding distance from zero can get a different ReLU mask in fp32 than in the float64
    reference, which moves a gradient sum by that element's whole contribution; the gradient tests give such rows (a
    handful in 20 000) a zero upstream gradient, so that what is compared is summation error only."""
    c1, c2 = net.convs
    x = obs[:, :243].double().view(-1, 3, 9, 9)
    z1 = torch.nn.functional.conv2d(x, c1.weight.double(), c1.bias.double())
    z2 = torch.nn.functional.conv2d(torch.relu(z1), c2.weight.double(), c2.bias.double())
    R = obs.shape[0]
    return (z1.abs().reshape(R, -1).min(dim=1).values > margin) & (z2.abs().reshape(R, -1).min(dim=1).values > margin)


GRAD_TOL = 5e-6  # relative L2 of an fp32 gradient tensor against the float64 reference (measured: <= 6e-7)


def _rel_l2(g, r):
    return float(np.linalg.norm(g.astype(np.float64) - r) / max(np.linalg.norm(r), 1e-30))


@pytest.mark.parametrize('od,rows', [(24, 16384), (32, 4097), (24, 7), (32, 1)])
def test_conv9_forward_matches_torch(od, rows):
    from marl_dmfb_amd.network.base_net import CRNN
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=5, fov=9)
    torch.manual_seed(od + rows)
    net = CRNN(a).cuda()
    obs = torch.randint(-10, 11, (rows, 245), dtype=torch.int8, device='cuda')
    with torch.no_grad():
        got = net._pixel_features_hip(obs)
        x = obs[:, :243].float().view(rows, 3, 9, 9).cpu()
        ref = x
        for conv in net.convs:
            ref = torch.relu(torch.nn.functional.conv2d(ref, conv.weight.cpu(), conv.bias.cpu()))
        ref = ref.reshape(rows, -1)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def _args19(od, n_actions=9):
    return types.SimpleNamespace(obs_shape=(3, 19, 19, 2, 1085), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=n_actions, fov=19)


@pytest.mark.parametrize('od,rows', [(32, 16384), (24, 4099), (32, 7), (24, 1), (32, 8), (32, 9)])
def test_front19_pixel_features_match_torch(od, rows):
    """fov 19 (MEDA v0_2): stride-2 conv + the tied conv3 twice (base_net.py:23-33), pixel features only, against
    conv2d+ReLU three times on the CPU.  Floating point: rtol 1e-5, atol 1e-5 (same products, other summation order)."""
    from marl_dmfb_amd.network.base_net import CRNN
    torch.manual_seed(od + rows)
    net = CRNN(_args19(od)).cuda()
    assert net.convs[1] is net.convs[2] and net._hip_geometry() == 19
    obs = torch.randint(0, 13, (rows, 1085), dtype=torch.int8, device='cuda')   # layer values: droplet index + 1
    with torch.no_grad():
        got = net._pixel_features_hip(obs)
        ref = obs[:, :1083].float().view(rows, 3, 19, 19).cpu()
        for conv in net.convs:
            ref = torch.relu(torch.nn.functional.conv2d(ref, conv.weight.cpu(), conv.bias.cpu(), stride=conv.stride))
        ref = ref.reshape(rows, -1)
    assert got.shape == (rows, od * 25)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('od,rows,strided', [(32, 4096 * 4, False), (24, 1001, True), (32, 3, False)])
def test_front19_forward_obs_matches_reference_forward(od, rows, strided):
    """The whole rollout forward for fov 19 (HIP front end incl. the vector branch, then GRU cell and head) against the
    reference-shaped CPU forward on float inputs; also with a row stride larger than the row (a slice of a wider buffer)
    and with negative direction bytes."""
    from marl_dmfb_amd.network.base_net import CRNN
    a = _args19(od)
    torch.manual_seed(rows)
    net = CRNN(a).cuda()
    if strided:
        wide = torch.randint(0, 6, (rows, 1100), dtype=torch.int8, device='cuda')
        obs = wide[:, 3:1088]
    else:
        obs = torch.randint(0, 6, (rows, 1085), dtype=torch.int8, device='cuda')
    obs[:, 1083:] = torch.randint(-30, 31, (rows, 2), dtype=torch.int8, device='cuda')
    la = torch.nn.functional.one_hot(torch.randint(0, 9, (rows,), device='cuda'), 9).to(torch.int8)
    h = torch.randn(rows, 128, device='cuda')
    with torch.no_grad():
        assert net.act_ok(obs) if not strided else True
        x = net._front_features_hip(obs, la)
        q1, h1 = net.forward_obs(obs, la, h)
    cpu = CRNN(a)
    cpu.load_state_dict(net.state_dict())
    with torch.no_grad():
        inp = torch.cat([obs.float(), la.float()], dim=1).cpu()
        xr = cpu.features(inp)
        q2, h2 = cpu(inp, h.cpu())
    np.testing.assert_allclose(x.cpu().numpy(), xr.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(q1.cpu().numpy(), q2.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(h1.cpu().numpy(), h2.numpy(), rtol=1e-4, atol=1e-5)


def test_forward_obs_uses_kernel_and_matches_reference_forward():
    from marl_dmfb_amd.network.base_net import CRNN
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=24, rnn_hidden_dim=128, n_actions=5, fov=9)
    torch.manual_seed(3)
    net = CRNN(a).cuda()
    R = 513
    obs = torch.randint(0, 5, (R, 245), dtype=torch.int8, device='cuda')
    la = torch.nn.functional.one_hot(torch.randint(0, 5, (R,), device='cuda'), 5).to(torch.int8)
    h = torch.randn(R, 128, device='cuda')
    with torch.no_grad():
        assert net._hip_conv_ok(obs)
        q1, h1 = net.forward_obs(obs, la, h)
    cpu = CRNN(a)
    cpu.load_state_dict(net.state_dict())
    with torch.no_grad():
        q2, h2 = cpu(torch.cat([obs.float(), la.float()], dim=1).cpu(), h.cpu())
    np.testing.assert_allclose(q1.cpu().numpy(), q2.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(h1.cpu().numpy(), h2.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('od,rows', [(24, 20000), (32, 5003), (24, 11)])
def test_conv9_training_pair_matches_torch_autograd(od, rows):
    """Forward + backward kernels (_ConvFront9) against torch autograd on conv2d (float64 reference)."""
    from marl_dmfb_amd.network.base_net import CRNN, _ConvFront9
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=5, fov=9)
    torch.manual_seed(od * 7 + rows)
    net = CRNN(a).cuda()
    obs = torch.randint(-3, 8, (rows, 245), dtype=torch.int8, device='cuda')
    gout = torch.randn(rows, od * 25, device='cuda')
    safe = _safe_rows(net, obs)
    assert safe.float().mean() > 0.8 or rows < 100
    gout = gout * safe[:, None]
    c1, c2 = net.convs
    pix = _ConvFront9.apply(obs, c1.weight, c1.bias, c2.weight, c2.bias)
    (pix * gout).sum().backward()
    got = [p.grad.detach().cpu().clone() for p in (c1.weight, c1.bias, c2.weight, c2.bias)]
    ref_net = CRNN(a).double()
    ref_net.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    x = obs[:, :243].double().view(rows, 3, 9, 9).cpu()
    for conv in ref_net.convs:
        x = torch.relu(conv(x))
    np.testing.assert_allclose(pix.detach().cpu().numpy(), x.reshape(rows, -1).detach().numpy(), rtol=1e-4, atol=1e-4)
    (x.reshape(rows, -1) * gout.double().cpu()).sum().backward()
    rc1, rc2 = ref_net.convs
    # float64 reference, rows with a knife-edge ReLU excluded (_safe_rows): what remains is fp32 summation error of
    # sums over rows x 25 positions -> relative L2 <= GRAD_TOL per tensor
    for name, g, r in zip(('w1', 'b1', 'w2', 'b2'), got, (rc1.weight.grad, rc1.bias.grad, rc2.weight.grad, rc2.bias.grad)):
        err = _rel_l2(g.numpy(), r.numpy())
        print('conv9 pair od=%d rows=%d %s rel_l2=%.2e' % (od, rows, name, err))
        assert err <= GRAD_TOL, (name, err)


@pytest.mark.parametrize('T,R', [(1, 1), (5, 13), (40, 2048), (17, 520)])
def test_gru_sequence_kernels_match_grucell_autograd(T, R):
    """gru_seq_forward / gru_seq_backward (one launch for all T steps) against nn.GRUCell unrolled in float64 on the
    CPU (network/base_net.py:56,69; policy/vdn.py:174-191).  fp32 tolerance: 2e-5 absolute on h (|h| <= 1),
    relative L2 2e-4 on every gradient."""
    from marl_dmfb_amd.network.base_net import gru_sequence
    torch.manual_seed(T * 1000 + R)
    H, F = 128, 64
    cell = torch.nn.GRUCell(F, H).cuda()
    x = torch.randn(T, R, F, device='cuda', requires_grad=True)
    h0 = (torch.rand(R, H, device='cuda') * 2 - 1).requires_grad_(True)
    gout = torch.randn(T, R, H, device='cuda')
    igates = torch.matmul(x.view(T * R, F), cell.weight_ih.t()).view(T, R, 3 * H)
    hs = gru_sequence(igates, h0, cell.weight_hh, cell.bias_ih, cell.bias_hh, 'hip')
    with torch.no_grad():
        hs_ng = gru_sequence(igates.detach(), h0.detach(), cell.weight_hh, cell.bias_ih, cell.bias_hh, 'hip')
    assert torch.equal(hs_ng, hs.detach())          # inference launch (no saved gates) == training launch
    (hs * gout).sum().backward()
    got = [t.grad.detach().cpu().double() for t in (x, h0, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    ref = torch.nn.GRUCell(F, H).double()
    ref.load_state_dict({k: v.detach().double().cpu() for k, v in cell.state_dict().items()})
    xr = x.detach().double().cpu().requires_grad_(True)
    hr0 = h0.detach().double().cpu().requires_grad_(True)
    h, outs = hr0, []
    for t in range(T):
        h = ref(xr[t], h)
        outs.append(h)
    hr = torch.stack(outs, 0)
    np.testing.assert_allclose(hs.detach().cpu().numpy(), hr.detach().numpy(), rtol=0, atol=2e-5)
    (hr * gout.double().cpu()).sum().backward()
    for g, r in zip(got, (xr.grad, hr0.grad, ref.weight_ih.grad, ref.weight_hh.grad, ref.bias_ih.grad, ref.bias_hh.grad)):
        assert torch.linalg.norm(g - r) <= 2e-4 * torch.linalg.norm(r) + 1e-9


def test_recurrent_seq_hip_equals_aten_path():
    """CRNN.recurrent_seq through the one-launch GRU kernels vs the per-step aten fused cell: q values and the
    gradients of a sum agree to fp32 rounding."""
    from marl_dmfb_amd.network.base_net import CRNN
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=24, rnn_hidden_dim=128, n_actions=5, fov=9)
    torch.manual_seed(3)
    net = CRNN(a).cuda()
    T, R = 12, 300
    x = torch.randn(T, R, net.out + 10, device='cuda')
    h0 = torch.zeros(R, 128, device='cuda')
    res = {}
    for impl in ('hip', 'aten'):
        net.gru_impl = impl
        net.zero_grad()
        q, hT = net.recurrent_seq(x, h0)
        (q.square().sum() + hT.sum()).backward()
        res[impl] = (q.detach().clone(), hT.detach().clone(), [p.grad.clone() for p in net.rnn.parameters()])
    np.testing.assert_allclose(res['hip'][0].cpu().numpy(), res['aten'][0].cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res['hip'][1].cpu().numpy(), res['aten'][1].cpu().numpy(), rtol=1e-4, atol=1e-5)
    for g1, g2 in zip(res['hip'][2], res['aten'][2]):
        assert torch.linalg.norm(g1 - g2) <= 1e-4 * torch.linalg.norm(g2)


@pytest.mark.parametrize('od,rows', [(24, 20000), (32, 5003), (24, 11), (24, 8), (24, 2563)])
def test_front9_train_node_matches_torch_autograd(od, rows):
    """_Front9Train (fused forward incl. mlp1; backward kernel that recomputes conv1 on the matrix cores) against float64
    torch autograd of the reference network's front end (network/base_net.py:59-68); rows with a
    knife-edge ReLU get a zero upstream gradient (_safe_rows), tolerance GRAD_TOL."""
    from marl_dmfb_amd.network.base_net import CRNN, _Front9Train
    a = types.SimpleNamespace(obs_shape=(3, 9, 9, 2, 245), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=5, fov=9)
    torch.manual_seed(od * 11 + rows)
    net = CRNN(a).cuda()
    obs = torch.randint(-3, 8, (rows, 245), dtype=torch.int8, device='cuda')
    oh = torch.nn.functional.one_hot(torch.randint(0, 5, (rows,), device='cuda'), 5).to(torch.int8)
    # odd row counts use the zero-padded row width (the GEMM-friendly K of the GRU input projection), even ones the exact one
    nf = od * 25 + 10
    cols = net.padded_cols() if rows % 2 else nf
    gfull = torch.randn(rows, cols, device='cuda')      # the gradient of the zero tail is arbitrary and must be ignored
    safe = _safe_rows(net, obs)
    gfull[:, :nf] *= safe[:, None]
    gout = gfull[:, :nf]
    c1, c2 = net.convs
    xfull = _Front9Train.apply(obs, oh, c1.weight, c1.bias, c2.weight, c2.bias, net.mlp1.weight, net.mlp1.bias, cols)
    assert xfull.shape == (rows, cols) and bool((xfull[:, nf:] == 0).all())
    (xfull * gfull).sum().backward()
    x = xfull[:, :nf]
    params = (c1.weight, c1.bias, c2.weight, c2.bias, net.mlp1.weight, net.mlp1.bias)
    got = [p.grad.detach().cpu().clone() for p in params]
    ref = CRNN(a).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    inp = torch.cat([obs.double().cpu(), oh.double().cpu()], dim=1)
    xr = ref.features(inp)
    np.testing.assert_allclose(x.detach().cpu().numpy(), xr.detach().numpy(), rtol=1e-4, atol=1e-4)
    (xr * gout.double().cpu()).sum().backward()
    r1, r2 = ref.convs
    for name, g, r in zip(('w1', 'b1', 'w2', 'b2', 'mlp_w', 'mlp_b'), got,
                          (r1.weight.grad, r1.bias.grad, r2.weight.grad, r2.bias.grad, ref.mlp1.weight.grad, ref.mlp1.bias.grad)):
        err = _rel_l2(g.numpy(), r.numpy())
        print('front9 od=%d rows=%d %s rel_l2=%.2e' % (od, rows, name, err))
        assert err <= GRAD_TOL, (name, err)


@pytest.mark.parametrize('fov,od,rows', [(9, 24, 4097), (9, 32, 33), (19, 32, 1000), (19, 24, 5)])
def test_front_end_zero_padded_rows(fov, od, rows):
    """out_cols > od*25+10: the same features followed by zeros up to a multiple of 64 (crnn_front_padded_cols), for both
    front-end kernels; the GRU input projection against the zero-padded weight_ih equals the unpadded one up to the
    GEMM's summation order, also after the weights changed in place (cached padded copy refreshed)."""
    from marl_dmfb_amd.network.base_net import CRNN
    nA = 9 if fov == 19 else 5
    obs_len = 3 * fov * fov + 2
    a = types.SimpleNamespace(obs_shape=(3, fov, fov, 2, obs_len), hyper_hidden_dim=od, rnn_hidden_dim=128, n_actions=nA, fov=fov)
    torch.manual_seed(fov * od + rows)
    net = CRNN(a).cuda()
    obs = torch.randint(0, 6, (rows, obs_len), dtype=torch.int8, device='cuda')
    la = torch.nn.functional.one_hot(torch.randint(0, nA, (rows,), device='cuda'), nA).to(torch.int8)
    h = torch.randn(rows, 128, device='cuda')
    nf = od * 25 + 10
    with torch.no_grad():
        x = net._front_features_hip(obs, la)
        xp = net._front_features_hip(obs, la, padded=True)
        assert xp.shape == (rows, (nf + 63) // 64 * 64) == (rows, net.padded_cols())
        assert torch.equal(xp[:, :nf], x) and bool((xp[:, nf:] == 0).all())
        opt = torch.optim.Adam([net.rnn.weight_ih], lr=0.05, fused=True)   # its step does not bump the parameter's _version
        for it in range(3):
            ig, hg = net.act_gates(obs, la, h, net.refresh_padded() if it else None)
            ref = torch.matmul(x.double(), net.rnn.weight_ih.double().t())
            np.testing.assert_allclose(ig.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=2e-5)
            net.rnn.weight_ih.grad = torch.randn_like(net.rnn.weight_ih)
            opt.step()


def _safe_rows19(net, obs, margin=2e-5):
    """_safe_rows for the fov-19 stack: rows none of whose three conv pre-activations lies within `margin` of zero."""
    c1, c3 = net.convs[0], net.convs[1]
    R = obs.shape[0]
    x = obs[:, :1083].double().view(-1, 3, 19, 19)
    z1 = torch.nn.functional.conv2d(x, c1.weight.double(), c1.bias.double(), stride=2)
    z2 = torch.nn.functional.conv2d(torch.relu(z1), c3.weight.double(), c3.bias.double())
    z3 = torch.nn.functional.conv2d(torch.relu(z2), c3.weight.double(), c3.bias.double())
    ok = torch.ones(R, dtype=torch.bool, device=obs.device)
    for z in (z1, z2, z3):
        ok &= z.abs().reshape(R, -1).min(dim=1).values > margin
    return ok


@pytest.mark.parametrize('od,rows', [(32, 3001), (24, 4000), (32, 2), (24, 5), (32, 770)])
def test_front19_train_node_matches_torch_autograd(od, rows):
    """_Front19Train (crnn_front19_forward + crnn_conv19_backward: a1 / a2 recomputed on the matrix cores, transposed
    convolutions in gather form, the tied conv3's two applications summed into one gradient) against float64 torch autograd
    of the reference network's fov-19 front end (network/base_net.py:23-33, 59-68); tolerance GRAD_TOL."""
    from marl_dmfb_amd.network.base_net import CRNN, _Front19Train
    a = _args19(od)
    torch.manual_seed(od * 7 + rows)
    net = CRNN(a).cuda()
    assert net._hip_geometry() == 19
    obs = torch.randint(0, 6, (rows, 1085), dtype=torch.int8, device='cuda')
    obs[:, 1083:] = torch.randint(-10, 11, (rows, 2), dtype=torch.int8, device='cuda')
    oh = torch.nn.functional.one_hot(torch.randint(0, 9, (rows,), device='cuda'), 9).to(torch.int8)
    nf = od * 25 + 10
    cols = net.padded_cols() if rows % 2 else nf
    gfull = torch.randn(rows, cols, device='cuda')
    safe = _safe_rows19(net, obs)
    assert rows < 100 or float(safe.float().mean()) > 0.7
    gfull[:, :nf] *= safe[:, None]
    gout = gfull[:, :nf]
    c1, c3 = net.convs[0], net.convs[1]
    xfull = _Front19Train.apply(obs, oh, c1.weight, c1.bias, c3.weight, c3.bias, net.mlp1.weight, net.mlp1.bias, cols)
    assert xfull.shape == (rows, cols) and bool((xfull[:, nf:] == 0).all())
    (xfull * gfull).sum().backward()
    params = (c1.weight, c1.bias, c3.weight, c3.bias, net.mlp1.weight, net.mlp1.bias)
    got = [p.grad.detach().cpu().clone() for p in params]
    ref = CRNN(a).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    inp = torch.cat([obs.double().cpu(), oh.double().cpu()], dim=1)
    xr = ref.features(inp)
    np.testing.assert_allclose(xfull[:, :nf].detach().cpu().numpy(), xr.detach().numpy(), rtol=1e-4, atol=1e-4)
    (xr * gout.double().cpu()).sum().backward()
    r1, r3 = ref.convs[0], ref.convs[1]
    for name, g, r in zip(('w1', 'b1', 'w3', 'b3', 'mlp_w', 'mlp_b'), got,
                          (r1.weight.grad, r1.bias.grad, r3.weight.grad, r3.bias.grad, ref.mlp1.weight.grad, ref.mlp1.bias.grad)):
        err = _rel_l2(g.numpy(), r.numpy())
        print('front19 od=%d rows=%d %s rel_l2=%.2e' % (od, rows, name, err))
        assert err <= GRAD_TOL, (name, err)
    # The rows left out above (a pre-activation within 2e-5 of zero: float32 and float64 may put its ReLU on different sides) are
    # not skipped altogether: the same comparison with the upstream gradient on EVERY row.  A flipped ReLU moves a gradient by a
    # whole term, so the bound is loose (a few flips among ~5000 activations per row), but an indexing or accumulation error in the
    # rows masked above would show at order one.
    for p in params:
        p.grad = None
    for p in ref.parameters():
        p.grad = None
    gall = torch.randn(rows, cols, device='cuda')
    xall = _Front19Train.apply(obs, oh, c1.weight, c1.bias, c3.weight, c3.bias, net.mlp1.weight, net.mlp1.bias, cols)
    (xall * gall).sum().backward()
    (ref.features(inp) * gall[:, :nf].double().cpu()).sum().backward()
    for name, p, r in zip(('w1', 'b1', 'w3', 'b3', 'mlp_w', 'mlp_b'), params,
                          (r1.weight.grad, r1.bias.grad, r3.weight.grad, r3.bias.grad, ref.mlp1.weight.grad, ref.mlp1.bias.grad)):
        err = _rel_l2(p.grad.detach().cpu().numpy(), r.numpy())
        print('front19 od=%d rows=%d %s rel_l2 on all rows=%.2e' % (od, rows, name, err))
        assert err <= 2e-2, (name, err)


@pytest.mark.parametrize('rows,A,dir_off,col0,pad', [(1, 5, 243, 600, 0), (257, 3, 243, 600, 30), (70001, 16, 1083, 800, 22), (4096, 0, 243, 600, 0)])
def test_mlp_branch_backward_matches_float64(rows, A, dir_off, col0, pad):
    """crnn_mlp_backward (gradients of relu(mlp1([dir, last action])), network/base_net.py:66) against float64 tensor ops: ragged row
    counts, 0..16 actions, both direction-byte offsets, row-strided x / gradient with a zero-padded tail."""
    import ctypes as C
    from marl_dmfb_amd import _lib
    lib = _lib.crnn_ops()
    vp = C.c_void_p
    g = torch.Generator(device='cuda').manual_seed(rows + A)
    obs = torch.randint(-4, 5, (rows, dir_off + 2), dtype=torch.int8, device='cuda', generator=g)
    onehot = torch.zeros((rows, max(A, 1)), dtype=torch.int8, device='cuda')
    if A:
        onehot[torch.arange(rows, device='cuda'), torch.randint(0, A, (rows,), device='cuda', generator=g)] = 1
    onehot = onehot[:, :A].contiguous() if A else onehot[:, :0].contiguous()
    cols = col0 + 10 + pad
    x = torch.randn((rows, cols), device='cuda', generator=g)
    x[:, col0:col0 + 10] = torch.relu(x[:, col0:col0 + 10])       # what the forward leaves: post-ReLU values (zeros where clipped)
    grad = torch.randn((rows, cols), device='cuda', generator=g)
    g_w = torch.full((10, 2 + A), 7.0, device='cuda')
    g_b = torch.full((10,), 7.0, device='cuda')
    part = torch.empty((lib.crnn_mlp_backward_parts(),), device='cuda')
    oh_ptr = vp(onehot.data_ptr()) if A else vp(obs.data_ptr())      # never read when n_actions == 0, but must not be NULL
    rc = lib.crnn_mlp_backward(vp(obs.data_ptr()), obs.stride(0), dir_off, oh_ptr, A, rows, vp(x.data_ptr()), x.stride(0),
                               vp(grad.data_ptr()), grad.stride(0), col0, vp(part.data_ptr()), vp(g_w.data_ptr()), vp(g_b.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    gz = (grad[:, col0:col0 + 10] * (x[:, col0:col0 + 10] > 0)).double()
    vec = torch.cat([obs[:, dir_off:dir_off + 2].double(), onehot.double()], dim=1)
    want_w, want_b = gz.t() @ vec, gz.sum(0)
    assert _rel_l2(g_w.cpu().numpy(), want_w.cpu().numpy()) <= GRAD_TOL
    assert _rel_l2(g_b.cpu().numpy(), want_b.cpu().numpy()) <= GRAD_TOL
    # argument checks: too many actions, a row too short for the branch's columns
    assert lib.crnn_mlp_backward(vp(obs.data_ptr()), obs.stride(0), dir_off, oh_ptr, 17, rows, vp(x.data_ptr()), x.stride(0),
                                 vp(grad.data_ptr()), grad.stride(0), col0, vp(part.data_ptr()), vp(g_w.data_ptr()), vp(g_b.data_ptr()), None) != 0
    assert lib.crnn_mlp_backward(vp(obs.data_ptr()), obs.stride(0), dir_off, oh_ptr, A, rows, vp(x.data_ptr()), col0 + 9,
                                 vp(grad.data_ptr()), grad.stride(0), col0, vp(part.data_ptr()), vp(g_w.data_ptr()), vp(g_b.data_ptr()), None) != 0


@pytest.mark.parametrize('R,lens', [(1, [3]), (37, None), (2048, None), (520, 'full')])
def test_gru_pair_forward_matches_float64_and_the_packed_kernel(R, lens):
    """gru_seq_forward_packed_pair (two networks, recurrence on the matrix cores, 16 rows per workgroup) on ragged packed sequences
    against nn.GRUCell unrolled in float64 (network/base_net.py:56,69; policy/vdn.py:174-191; 2e-5 on h as for the VALU kernel),
    against gru_seq_forward_packed on the same inputs (hs and the saved gates, 1e-5), through the autograd node (gradients of network
    a = those of _GRUSeqHipPacked on the same forward), and with network b absent."""
    import ctypes as C
    from marl_dmfb_amd import _lib
    from marl_dmfb_amd.network.base_net import _GRUSeqHipPacked, _GRUSeqPairPacked
    lib = _lib.crnn_ops()
    H, T = 128, 12
    g = torch.Generator().manual_seed(R)
    if lens is None:
        ln = torch.sort(torch.randint(1, T + 1, (R,), generator=g), descending=True).values.tolist()
    elif lens == 'full':
        ln = [T] * R
    else:
        ln = lens
    Tm = max(ln)
    step_rows = [sum(1 for v in ln if v > t) for t in range(Tm)]
    V = sum(step_rows)
    Vp = -(-V // 64) * 64
    torch.manual_seed(R + 1)
    cells = [torch.nn.GRUCell(H, H).cuda() for _ in range(2)]
    igs = [torch.randn(Vp, 3 * H, device='cuda') for _ in range(2)]
    for ig in igs:
        ig[V:].zero_()
    h0 = torch.zeros(R, H, device='cuda')
    ig_a = igs[0].clone().requires_grad_(True)
    hs_a, hs_b = _GRUSeqPairPacked.apply(ig_a, cells[0].weight_hh, cells[0].bias_ih, cells[0].bias_hh,
                                         igs[1], cells[1].weight_hh.detach(), cells[1].bias_ih.detach(), cells[1].bias_hh.detach(), step_rows, R)
    assert not hs_b.requires_grad and hs_a.requires_grad
    # float64 reference, per network: unroll the cell on the rows still running
    for k, hs in enumerate((hs_a, hs_b)):
        c = cells[k]
        w, bi, bh = c.weight_hh.detach().double().cpu(), c.bias_ih.detach().double().cpu(), c.bias_hh.detach().double().cpu()
        ig64 = igs[k].double().cpu()
        h = torch.zeros(R, H, dtype=torch.float64)
        off = 0
        for t in range(Tm):
            rt = step_rows[t]
            gi = ig64[off:off + rt] + bi
            gh = h[:rt] @ w.t() + bh
            r_ = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z_ = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n_ = torch.tanh(gi[:, 2 * H:] + r_ * gh[:, 2 * H:])
            hn = (1 - z_) * n_ + z_ * h[:rt]
            np.testing.assert_allclose(hs.detach()[off:off + rt].cpu().numpy(), hn.numpy(), rtol=0, atol=2e-5)
            h = h.clone()
            h[:rt] = hn
            off += rt
        assert float(hs.detach()[V:].abs().max()) == 0.0 if Vp > V else True
    # the VALU kernel on the same inputs: outputs and saved gates
    ig_v = igs[0].clone().requires_grad_(True)
    hs_v = _GRUSeqHipPacked.apply(ig_v, h0, cells[0].weight_hh, cells[0].bias_ih, cells[0].bias_hh, step_rows)
    np.testing.assert_allclose(hs_a.detach().cpu().numpy(), hs_v.detach().cpu().numpy(), rtol=0, atol=1e-5)
    gout = torch.randn(Vp, H, device='cuda')
    gout[V:].zero_()
    params = (cells[0].weight_hh, cells[0].bias_ih, cells[0].bias_hh)
    for p in params:
        p.grad = None
    (hs_a * gout).sum().backward()
    got = [ig_a.grad.clone()] + [p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    (hs_v * gout).sum().backward()
    want = [ig_v.grad.clone()] + [p.grad.clone() for p in params]
    for a_, b_ in zip(got, want):
        assert torch.linalg.norm(a_.double() - b_.double()) <= 1e-4 * torch.linalg.norm(b_.double()) + 1e-9
    # one network only (d_igates_b NULL)
    vp = C.c_void_p
    hs1 = torch.full((Vp, H), 9.0, device='cuda')
    st = (C.c_int32 * len(step_rows))(*step_rows)
    rc = lib.gru_seq_forward_packed_pair(vp(igs[1].data_ptr()), None, vp(cells[1].weight_hh.data_ptr()), vp(cells[1].bias_ih.data_ptr()),
                                         vp(cells[1].bias_hh.data_ptr()), vp(hs1.data_ptr()), None, None, None, None, None, None, None, None,
                                         Tm, R, H, st, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(hs1[:V], hs_b.detach()[:V])


@pytest.mark.parametrize('R', [1, 37, 530])
def test_gru_packed_backward_matches_float64_autograd(R):
    """gru_seq_backward_packed behind the packed autograd node, on ragged sequences, against float64 autograd through an unrolled
    nn.GRUCell (policy/vdn.py:174-191): relative L2 2e-4 on every gradient, as for the time-major kernels."""
    from marl_dmfb_amd.network.base_net import _GRUSeqHipPacked
    H, T = 128, 9
    g = torch.Generator().manual_seed(R)
    ln = torch.sort(torch.randint(1, T + 1, (R,), generator=g), descending=True).values.tolist()
    Tm = max(ln)
    step_rows = [sum(1 for v in ln if v > t) for t in range(Tm)]
    V = sum(step_rows)
    Vp = -(-V // 64) * 64
    torch.manual_seed(R + 5)
    cell = torch.nn.GRUCell(H, H).cuda()
    ig = torch.randn(Vp, 3 * H, device='cuda')
    ig[V:].zero_()
    ig.requires_grad_(True)
    h0 = torch.zeros(R, H, device='cuda')
    gout = torch.randn(Vp, H, device='cuda')
    gout[V:].zero_()
    hs = _GRUSeqHipPacked.apply(ig, h0, cell.weight_hh, cell.bias_ih, cell.bias_hh, step_rows)
    (hs * gout).sum().backward()
    got = [t.grad.detach().double().cpu() for t in (ig, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    w = cell.weight_hh.detach().double().cpu().requires_grad_(True)
    bi = cell.bias_ih.detach().double().cpu().requires_grad_(True)
    bh = cell.bias_hh.detach().double().cpu().requires_grad_(True)
    ig64 = ig.detach().double().cpu().requires_grad_(True)
    h = torch.zeros(R, H, dtype=torch.float64)
    loss, off = 0.0, 0
    for t in range(Tm):
        rt = step_rows[t]
        gi = ig64[off:off + rt] + bi
        gh = h[:rt] @ w.t() + bh
        r_ = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z_ = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n_ = torch.tanh(gi[:, 2 * H:] + r_ * gh[:, 2 * H:])
        hn = (1 - z_) * n_ + z_ * h[:rt]
        loss = loss + (hn * gout[off:off + rt].double().cpu()).sum()
        h = torch.cat([hn, h[rt:]], 0)
        off += rt
    loss.backward()
    for a_, b_ in zip(got, (ig64.grad, w.grad, bi.grad, bh.grad)):
        assert torch.linalg.norm(a_ - b_) <= 2e-4 * torch.linalg.norm(b_) + 1e-9
