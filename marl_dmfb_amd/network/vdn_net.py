"""VDN mixer (reference network/vdn_net.py:5-10): the team value is the plain sum of the
per-agent values over the agent axis.  It has no parameters, so its state_dict is empty, which
is what the reference saves as `*_vdn_net_params.pkl`."""
import torch.nn as nn


class VDNNet(nn.Module):
    AGENT_AXIS = 2  # q_values: (episodes, T, n_agents)

    def forward(self, q_values):
        return q_values.sum(dim=self.AGENT_AXIS, keepdim=True)
