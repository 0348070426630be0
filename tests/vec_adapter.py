"""numpy-in / numpy-out adapter over marl_dmfb_amd.env.dmfb.VecDMFB with the method set of
oracle.dmfb_oracle.DmfbOracle, so the same replay/parity helpers drive both."""
import numpy as np
import torch

from marl_dmfb_amd.env.dmfb import VecDMFB


class VecAdapter:
    def __init__(self, **kw):
        self.v = VecDMFB(**kw)
        self.E, self.n = self.v.n_envs, self.v.n_agents

    def reset(self, mask=None, new=False):
        self.v.reset(mask=mask, new=new)

    def restart(self, mask=None):
        self.v.restart(mask=mask)

    def set_task(self, starts, ends):
        self.v.set_task(np.asarray(starts, np.int32), np.asarray(ends, np.int32))

    def set_blocks(self, blocks):
        self.v.set_blocks(np.asarray(blocks, np.int32))

    def get_blocks(self):
        return self.v.get_blocks().cpu().numpy()

    def get_task(self):
        s, e = self.v.get_task()
        return s.cpu().numpy(), e.cpu().numpy()

    def get_state(self):
        return {k: t.cpu().numpy() for k, t in self.v.get_state().items()}

    def get_map(self, which):
        return self.v.get_map(which).cpu().numpy()

    def set_map(self, which, arr):
        self.v.set_map(which, np.asarray(arr, np.float64))

    def step(self, actions, uniforms=None, record=True, autoreset=False):
        a = actions if isinstance(actions, torch.Tensor) else np.asarray(actions)
        obs, r, d, info = self.v.step(a, uniforms, record=record, autoreset=autoreset)
        torch.cuda.synchronize()
        self.last_obs = obs.cpu().numpy()
        self.last_info = {k: t.cpu().numpy() for k, t in info.items()}
        return r.cpu().numpy(), d.cpu().numpy(), info['constraints'].cpu().numpy(), info['success'].cpu().numpy()

    def observe(self):
        return self.v.observe().cpu().numpy()


def make_vec(**kw):
    return VecAdapter(**kw)


class MedaAdapter:
    """Same idea for marl_dmfb_amd.env.meda.VecMEDA vs oracle.meda_oracle.MedaOracle."""

    def __init__(self, **kw):
        from marl_dmfb_amd.env.meda import VecMEDA
        self.v = VecMEDA(**kw)
        self.E, self.n = self.v.n_envs, self.v.n_agents

    def reset(self, mask=None):
        self.v.reset(mask=mask)

    def restart(self, mask=None):
        self.v.restart(mask=mask)

    def set_task(self, starts, ends):
        self.v.set_task(np.asarray(starts, np.int32), np.asarray(ends, np.int32))

    def get_task(self):
        s, e = self.v.get_task()
        return s.cpu().numpy(), e.cpu().numpy()

    def get_state(self):
        return {k: t.cpu().numpy() for k, t in self.v.get_state().items()}

    def get_map(self, which):
        return self.v.get_map(which).cpu().numpy()

    def set_map(self, which, arr):
        self.v.set_map(which, np.asarray(arr, np.float64))

    def step(self, actions, uniforms=None, autoreset=False):
        a = actions if isinstance(actions, torch.Tensor) else np.asarray(actions)
        obs, r, d, info = self.v.step(a, uniforms, autoreset=autoreset)
        torch.cuda.synchronize()
        self.last_obs = obs.cpu().numpy()
        self.last_info = {k: t.cpu().numpy() for k, t in info.items()}
        return r.cpu().numpy(), d.cpu().numpy(), info['constraints'].cpu().numpy(), info['success'].cpu().numpy()

    def observe(self):
        return self.v.observe().cpu().numpy()
