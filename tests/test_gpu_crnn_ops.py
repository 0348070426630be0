"""The hand-written HIP conv front end (include/crnn_ops.h) against a plain PyTorch fp32 reference of
the same op (conv2d+ReLU twice, network/base_net.py:63-65).  Floating point: rtol 1e-5, atol 1e-5
(same products, different summation order)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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
