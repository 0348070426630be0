"""End-to-end sanity on the GPU: the vectorised rollout + HBM replay + VDN.learn loop actually learns the
routing task (greedy team reward rises, constraint violations vanish) within a few seconds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_short_training_run_improves_greedy_policy():
    from marl_dmfb_amd.common.arguments import make_args
    from marl_dmfb_amd.env.dmfb import VecDMFB
    from marl_dmfb_amd.train import Trainer
    E, rounds = 512, 60
    torch.manual_seed(0)
    env = VecDMFB(10, 10, 4, fov=9, n_envs=E, seed=7, device='cuda:0')
    args = make_args(device='cuda:0', n_envs=E, batch_size=256, train_time=4, buffer_size=8 * E,
                     anneal_steps=E * 40 * rounds * 0.6, **env.get_env_info())
    tr = Trainer(env, args)
    r0, _, c0, _ = tr.rolloutWorker.evaluate(2)
    for _ in range(rounds):
        tr.collect_and_learn()
    r1, _, c1, _ = tr.rolloutWorker.evaluate(2)
    assert torch.isfinite(tr.agents.policy.last_loss)
    assert r1 > r0 + 40.0, (r0, r1)          # untrained greedy policy collides constantly (reward around -100)
    assert c1 < 0.2 * c0 + 1.0, (c0, c1)
