"""Agents (reference agent/agent.py:7-70): picks the policy class, epsilon-greedy action
selection, trims a sampled batch to its longest episode and calls policy.learn.

`choose_action` keeps the reference's one-agent signature; `choose_actions` is the batched form
the vectorised rollout uses: one forward over (envs x agents) rows and epsilon-greedy on device."""
import numpy as np
import torch


class Agents:
    def __init__(self, args):
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        if args.alg == 'vdn':
            from ..policy.vdn import VDN
            self.policy = VDN(args)
        else:
            raise Exception('No such algorithm')
        self.args = args
        self.device = self.policy.device

    # ---- reference signature, one agent of one env (agent/agent.py:22-48)
    def choose_action(self, obs, last_action, agent_num, avail_actions, epsilon, evaluate=False):
        inputs = np.asarray(obs).copy()
        avail_actions_ind = np.nonzero(avail_actions)[0]
        if self.args.last_action:
            inputs = np.hstack((inputs, last_action))
        hidden_state = self.policy.eval_hidden[:, agent_num, :].to(self.device)
        inputs = torch.tensor(inputs, dtype=torch.float32, device=self.device).unsqueeze(0)
        avail = torch.tensor(avail_actions, dtype=torch.float32, device=self.device).unsqueeze(0)
        with torch.no_grad():
            q_value, h = self.policy.eval_rnn(inputs, hidden_state)
        self.policy.eval_hidden[:, agent_num, :] = h.to(self.policy.eval_hidden.device)
        q_value[avail == 0.0] = -float('inf')
        if np.random.uniform() < epsilon and not evaluate:
            action = np.random.choice(avail_actions_ind)
        else:
            action = torch.argmax(q_value)
        return action

    # ---- batched: obs (E, n, obs) int8, last_action (E, n, A), hidden (E*n, H)
    @torch.no_grad()
    def choose_actions(self, obs, last_action, hidden, epsilon, evaluate=False, generator=None):
        E, n = obs.shape[0], obs.shape[1]
        q, h = self.policy.eval_rnn.forward_obs(obs.reshape(E * n, -1), last_action.reshape(E * n, -1), hidden)
        greedy = q.argmax(dim=1)
        if evaluate:
            return greedy.view(E, n), h
        explore = torch.rand(E * n, device=q.device, generator=generator) < epsilon
        rnd = torch.randint(0, self.n_actions, (E * n,), device=q.device, generator=generator)
        return torch.where(explore, rnd, greedy).view(E, n), h

    def _get_max_episode_len(self, batch):
        """1 + the largest index of an episode's first terminated step (agent/agent.py:51-61)."""
        terminated = batch['terminated']
        if not isinstance(terminated, torch.Tensor):
            terminated = torch.as_tensor(terminated)
        t = (terminated[:, :, 0] == 1)
        has = t.any(dim=1)
        first = torch.where(has, t.int().argmax(dim=1), torch.full_like(has, -1, dtype=torch.int64))
        return int(first.max().item()) + 1

    @staticmethod
    def first_terminated_bound(terminated):
        """Device scalar: the `_get_max_episode_len` of a batch, not yet read by the host (train.py reads it together with
        the round's step count, one transfer)."""
        t = (terminated[:, :, 0] == 1)
        has = t.any(dim=1)
        first = torch.where(has, t.int().argmax(dim=1), torch.full_like(has, -1, dtype=torch.int64))
        return first.max() + 1

    def train(self, batch, train_step, epsilon=None, max_len=None):
        """agent/agent.py:63-70.  max_len: an upper bound of `_get_max_episode_len(batch)` the caller already holds on the host
        (Trainer: the longest episode ever stored in the replay buffer).  The steps between the batch's own length and the bound
        are padded in every episode of the batch: their TD errors are masked to exact zeros, so loss and gradients are those of
        the exact length -- and the host does not have to read the length back from the device before it can queue the learn."""
        max_episode_len = self._get_max_episode_len(batch) if max_len is None else int(max_len)
        for key in batch.keys():
            if key != 'z':
                batch[key] = batch[key][:, :max_episode_len]
        return self.policy.learn(batch, max_episode_len, train_step, epsilon)
