"""Host side of the vectorised MEDA environment: `VecMEDA` drives E lock-step chips through the C
ABI in include/meda_vec.h (kernels in marl_dmfb_amd/csrc/meda_*.h*); `MEDAEnv` is the
reference-shaped single-chip facade (env/MEDA/meda.py:457-681).

Two deliberate differences from the reference, both stated by SURVEY.md 8(d)/(f3):
  * observations are int8 (the reference builds float64 arrays whose values are small integers);
    the facade widens them back to float64 so callers see the reference's dtype;
  * `get_env_info()['obs_shape']` of the facade is the reference's (an int, meda.py:676-681);
    `VecMEDA.get_env_info()` returns the tuple shape the networks need."""
import ctypes as C

import numpy as np
import torch

from .. import _lib

MEDA_STEP_AUTORESET = 2
MEDA_ACT_I32, MEDA_ACT_I8, MEDA_ACT_I64 = 0, 16, 32
MAPS = {'health': 0, 'usage': 1, 'degrade': 2}
_ERRORS = {
    -1: (ValueError, 'bad argument'),
    -3: (RuntimeError, 'Too many droplets in the MEDA array'),   # env/MEDA/meda.py:151-154
    -4: (AssertionError, 'w > 0 and l > 0'),                     # env/MEDA/meda.py:472
    -5: (AssertionError, 'n_agents > 0'),                        # env/MEDA/meda.py:473
    -6: (NotImplementedError, 'configuration outside the build limits (include/meda_vec.h)'),
    -8: (RuntimeError, 'env was created without health/usage/degrade maps (pass with_maps=True)'),
}


def _check(rc):
    if rc == 0:
        return
    if rc == -100:
        raise RuntimeError('HIP runtime error %d in meda_vec' % _lib.meda_vec().meda_vec_last_hip_error())
    exc, msg = _ERRORS.get(rc, (RuntimeError, 'meda_vec error %d' % rc))
    raise exc(msg)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class VecMEDA:
    """E independent MEDA chips advanced in lock-step on one MI355X (ctor of MEDAEnv, meda.py:469,
    plus n_envs / seed / env_id0 / with_maps)."""

    def __init__(self, width, length, n_agents, n_blocks=0, fov=19, stall=True, b_degrade=False, per_degrade=0.1,
                 n_envs=1, seed=0, with_maps=False, env_id0=0, device=None, version=0):
        self.lib = _lib.meda_vec()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('VecMEDA runs on the GPU only (no CPU fallback)')
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.width, self.length, self.n_agents, self.fov, self.n_envs = width, length, n_agents, fov, n_envs
        self.cfg = _lib.MedaVecConfig(width, length, n_agents, fov, int(bool(b_degrade)), int(bool(with_maps)),
                                      float(per_degrade), n_envs, env_id0, seed, self.device.index or 0, int(version))
        self.version = int(version)
        _check(self.lib.meda_vec_check_config(C.byref(self.cfg)))
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.meda_vec_create(C.byref(self.cfg), self._stream(), C.byref(self.h)))
        self.obs_len = (3 if self.version == 2 else 4) * fov * fov + 2
        self.max_step = width + length
        self.timing = None
        E, n, dev = n_envs, n_agents, self.device
        self.obs = torch.zeros((E, n, self.obs_len), dtype=torch.int8, device=dev)
        self.rewards = torch.zeros((E, n), dtype=torch.float64, device=dev)
        self.dones = torch.zeros((E, n), dtype=torch.uint8, device=dev)
        self.fail = torch.zeros((E,), dtype=torch.float64, device=dev)
        self.success = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self.team_reward = torch.zeros((E,), dtype=torch.float64, device=dev)
        self.terminated = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self._out = _lib.MedaVecStepOut(self.rewards.data_ptr(), self.dones.data_ptr(), self.fail.data_ptr(),
                                        self.success.data_ptr(), self.obs.data_ptr(), self.team_reward.data_ptr(),
                                        self.terminated.data_ptr())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.meda_vec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_env_info(self):
        return {'n_actions': 9, 'n_agents': self.n_agents,
                'obs_shape': (3 if self.version == 2 else 4, self.fov, self.fov, 2, self.obs_len),
                'episode_limit': self.max_step}

    def _dev(self, a, dtype):
        if a is None:
            return None
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def reset(self, mask=None, new=False, obs=None):
        obs = self.obs if obs is None else obs
        _check(self.lib.meda_vec_reset(self.h, _ptr(self._dev(mask, torch.uint8)), _ptr(obs), self._stream()))
        return obs

    def restart(self, mask=None, obs=None):
        obs = self.obs if obs is None else obs
        _check(self.lib.meda_vec_restart(self.h, _ptr(self._dev(mask, torch.uint8)), _ptr(obs), self._stream()))
        return obs

    def set_task(self, starts, ends):
        s = self._dev(starts, torch.int32).reshape(self.n_envs, self.n_agents, 2)
        e = self._dev(ends, torch.int32).reshape(self.n_envs, self.n_agents, 2)
        _check(self.lib.meda_vec_set_task(self.h, _ptr(s), _ptr(e), self._stream()))

    def get_task(self):
        s = torch.empty((self.n_envs, self.n_agents, 2), dtype=torch.int32, device=self.device)
        e = torch.empty_like(s)
        _check(self.lib.meda_vec_get_task(self.h, _ptr(s), _ptr(e), self._stream()))
        return s, e

    def step(self, actions, uniforms=None, record=True, autoreset=False, active=None, out=None):
        """MEDAEnv.step for all envs (meda.py:513-539); returns (obs, rewards, dones, info) device
        tensors reused by the next call; info = dict(constraints (= fail, float64), success,
        team_reward, terminated)."""
        if not isinstance(actions, torch.Tensor) or actions.device != self.device:
            actions = self._dev(actions, torch.int32)
        flag = {torch.int64: MEDA_ACT_I64, torch.int8: MEDA_ACT_I8, torch.int32: MEDA_ACT_I32}.get(actions.dtype)
        if flag is None:
            actions, flag = actions.to(torch.int32), MEDA_ACT_I32
        actions = actions.contiguous()
        if actions.numel() != self.n_envs * self.n_agents:
            raise RuntimeError('The number of actions is not the same as n_droplets')  # meda.py:242-244
        u = self._dev(uniforms, torch.float64)
        act = self._dev(active, torch.uint8)
        flags = flag | (MEDA_STEP_AUTORESET if autoreset else 0)
        if self.timing is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _check(self.lib.meda_vec_step(self.h, _ptr(actions), _ptr(u), _ptr(act), flags, C.byref(out or self._out),
                                      self._stream()))
        if self.timing is not None:
            ev1.record()
            self.timing.append((ev0, ev1))
        info = {'constraints': self.fail, 'success': self.success, 'team_reward': self.team_reward,
                'terminated': self.terminated}
        return self.obs, self.rewards, self.dones, info

    def observe(self, mask=None, obs=None):
        obs = self.obs if obs is None else obs
        _check(self.lib.meda_vec_observe(self.h, _ptr(self._dev(mask, torch.uint8)), _ptr(obs), self._stream()))
        return obs

    def launch_shape(self):
        """Chips per workgroup of the launches the handle makes (include/meda_vec.h: meda_vec_launch_shape)."""
        out = (C.c_int32 * 4)()
        _check(self.lib.meda_vec_launch_shape(self.h, C.byref(out)))
        return {'step_tile': out[0], 'observe_tile': out[1], 'observe_block': out[2], 'observe_workgroups': out[3]}

    def observe_timing(self, enable):
        """Start/stop collecting the dispatch time stamps of the observation kernel (meda_vec_observe_timing)."""
        _check(self.lib.meda_vec_observe_timing(self.h, int(bool(enable))))

    def observe_timing_read(self):
        """(summed kernel duration in microseconds, launches) since the last read; synchronises the host."""
        us, n = C.c_double(0.0), C.c_int(0)
        _check(self.lib.meda_vec_observe_timing_read(self.h, C.byref(us), C.byref(n)))
        return us.value, n.value

    def get_state(self):
        E, n, dev = self.n_envs, self.n_agents, self.device
        pos = torch.empty((E, n, 2), dtype=torch.int32, device=dev)
        status = torch.empty((E, n), dtype=torch.uint8, device=dev)
        sc = torch.empty((E,), dtype=torch.int32, device=dev)
        failed = torch.empty((E,), dtype=torch.uint8, device=dev)
        _check(self.lib.meda_vec_get_state(self.h, _ptr(pos), _ptr(status), _ptr(sc), _ptr(failed), self._stream()))
        return {'pos': pos, 'status': status, 'step_count': sc, 'failed': failed}

    def get_map(self, which):
        buf = torch.empty((self.n_envs, self.width, self.length), dtype=torch.float64, device=self.device)
        _check(self.lib.meda_vec_get_map(self.h, MAPS[which], _ptr(buf), self._stream()))
        return buf

    def set_map(self, which, arr):
        t = self._dev(arr, torch.float64).expand(self.n_envs, self.width, self.length).contiguous()
        _check(self.lib.meda_vec_set_map(self.h, MAPS[which], _ptr(t), self._stream()))


class MEDAEnv:
    """Single-chip facade with the reference's protocol (env/MEDA/meda.py:457-681)."""
    VERSION = 0

    def __init__(self, w, l, n_agents, n_blocks=0, fov=19, stall=True, b_degrade=False, per_degrade=0.1, show=False,
                 savemp4=False, seed=0, device=None):
        assert w > 0 and l > 0
        assert n_agents > 0
        if show or savemp4:
            raise NotImplementedError('rendering is out of scope (SURVEY.md section 2, row 4)')
        self.agents = ['player_{}'.format(i) for i in range(n_agents)]
        self.possible_agents = self.agents[:]
        self.width, self.length, self.fov = w, l, fov
        self.max_step = w + l
        self._vec = VecMEDA(w, l, n_agents, fov=fov, b_degrade=b_degrade, per_degrade=per_degrade, n_envs=1, seed=seed,
                            with_maps=True, device=device, version=self.VERSION)
        self.rewards = {i: 0. for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        self.fails = 0

    m_health = property(lambda self: self._vec.get_map('health')[0].cpu().numpy(),
                        lambda self, v: self._vec.set_map('health', np.asarray(v)))
    m_usage = property(lambda self: self._vec.get_map('usage')[0].cpu().numpy(),
                       lambda self, v: self._vec.set_map('usage', np.asarray(v)))
    m_degrade = property(lambda self: self._vec.get_map('degrade')[0].cpu().numpy(),
                         lambda self, v: self._vec.set_map('degrade', np.asarray(v)))

    def _obs_list(self, obs):
        o = obs[0].cpu().numpy()
        if self.VERSION == 0:
            o = o.astype(np.float64)   # the base env returns float64 rows; v0_2 returns int8 (meda.py:860)
        return [o[i].copy() for i in range(len(self.agents))]

    def step(self, actions):
        acts = [actions[a] for a in self.agents] if isinstance(actions, dict) else list(actions)
        if len(acts) != len(self.agents):
            raise RuntimeError('The number of actions is not the same as n_droplets')
        obs, rewards, dones, info = self._vec.step(np.asarray(acts, np.int32)[None])
        r, d = rewards[0].cpu().numpy(), dones[0].cpu().numpy()
        self.step_count += 1
        fail = float(info['constraints'][0].item())
        self.fails += fail
        for k, a in enumerate(self.agents):
            self.rewards[a] = float(r[k])
            self.dones[a] = bool(d[k])
        return self._obs_list(obs), self.rewards, self.dones, {'constraints': fail, 'success': int(info['success'][0].item())}

    def reset(self):
        self.rewards = {i: 0. for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        self.fails = 0
        return self._obs_list(self._vec.reset())

    def restart(self, index=None):
        self.rewards = {i: 0. for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        return self._obs_list(self._vec.restart())

    def getObs(self):
        return self._obs_list(self._vec.observe())

    def get_env_info(self):
        return {'n_actions': 9, 'n_agents': len(self.agents), 'obs_shape': self._vec.obs_len,
                'episode_limit': self.max_step}

    def seed(self, seed=None):
        pass

    def render(self, close=False):
        pass

    def close(self):
        pass


class MEDAEnv_v0_2(MEDAEnv):
    """env/MEDA/meda.py:846-897: 3-layer int8 observation with the direction zoomed to 30x30."""
    VERSION = 2
