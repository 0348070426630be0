"""Host logic of Agents (reference agent/agent.py:49-70) that runs without a GPU."""
import numpy as np
import torch

from marl_dmfb_amd.agent.agent import Agents


def _ref_max_len(terminated, episode_limit):
    """agent/agent.py:49-59 restated (first terminated step per episode, max over episodes, +1)."""
    m = 0
    for e in range(terminated.shape[0]):
        for t in range(episode_limit):
            if terminated[e, t, 0] == 1:
                if t + 1 >= m:
                    m = t + 1
                break
    return m


def test_max_episode_len_matches_reference_loop():
    rng = np.random.default_rng(0)
    for trial in range(50):
        B, T = int(rng.integers(1, 9)), int(rng.integers(1, 12))
        term = (rng.random((B, T, 1)) < 0.25)
        if trial % 7 == 0:
            term[:] = False                      # no episode terminated: the reference returns 0
        if trial % 5 == 0:
            term[rng.integers(0, B)] = False     # one never-terminating episode is ignored by the reference
        want = _ref_max_len(term, T)
        assert Agents._get_max_episode_len(None, {'terminated': torch.as_tensor(term)}) == want
        assert Agents._get_max_episode_len(None, {'terminated': term}) == want   # numpy batches (reference type) too
