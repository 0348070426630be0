"""Host side of the vectorised DMFB environment.

`VecDMFB` drives E lock-step chips through the C ABI in include/dmfb_vec.h (HIP kernels in
marl_dmfb_amd/csrc/dmfb_vec.hip); every array it hands back is a torch tensor living in HBM.
`DMFBenv` is the reference-shaped single-chip facade with the object protocol of the
reference's `DMFBenv` (env/DMFB/dmfb.py:474-640), so code written against the reference
(`RolloutWorker`, `evaDegre.py`, ...) can drive the HIP path unchanged.  PyTorch is used
only for device memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

DMFB_STEP_RECORD, DMFB_STEP_AUTORESET = 1, 2
DMFB_ACT_I32, DMFB_ACT_I8, DMFB_ACT_I64 = 0, 16, 32
MAPS = {'health': 0, 'usage': 1, 'degrade': 2}

# error code -> the exception the reference raises for the same condition
_ERRORS = {
    -1: (ValueError, 'bad argument'),
    -2: (RuntimeError, 'Fov is too large'),             # env/DMFB/dmfb.py:139-140
    -3: (TypeError, 'Too many droplets for DMFB'),       # env/DMFB/dmfb.py:144-146
    -4: (AssertionError, 'width >= 5 and length >= 5'),  # env/DMFB/dmfb.py:489
    -5: (AssertionError, 'n_agents > 0'),                # env/DMFB/dmfb.py:490
    -6: (NotImplementedError, 'configuration outside the build limits (include/dmfb_vec.h)'),
    -7: (TypeError, 'action is illegal'),                # env/DMFB/dmfb.py:116
    -8: (RuntimeError, 'env was created without health/usage/degrade maps (pass with_maps=True)'),
}


def _check(rc):
    if rc == 0:
        return
    if rc == -100:
        raise RuntimeError('HIP runtime error %d in dmfb_vec' % _lib.dmfb_vec().dmfb_vec_last_hip_error())
    exc, msg = _ERRORS.get(rc, (RuntimeError, 'dmfb_vec error %d' % rc))
    raise exc(msg)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class VecDMFB:
    """E independent DMFB chips advanced in lock-step on one MI355X.

    Constructor arguments are those of the reference's DMFBenv (dmfb.py:487) plus the batch:
    n_envs, seed (Philox key), env_id0 (global index of env 0 when a batch is sharded over
    ranks), with_maps (keep health/usage/degrade maps although b_degrade is False)."""

    def __init__(self, width, length, n_agents, n_blocks=0, fov=5, stall=True, b_degrade=False,
                 per_degrade=0.1, n_envs=1, seed=0, with_maps=False, env_id0=0, device=None):
        self.lib = _lib.dmfb_vec()
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('VecDMFB runs on the GPU only (no CPU fallback)')
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.width, self.length, self.n_agents, self.fov = width, length, n_agents, fov
        self.n_envs, self.stall, self.b_degrade = n_envs, bool(stall), bool(b_degrade)
        self.n_blocks = n_blocks
        self.has_maps = bool(b_degrade or with_maps)
        self.cfg = _lib.DmfbVecConfig(width, length, n_agents, n_blocks, fov, int(bool(stall)), int(bool(b_degrade)),
                                      int(bool(with_maps)), float(per_degrade), n_envs, env_id0, seed,
                                      self.device.index or 0)
        _check(self.lib.dmfb_vec_check_config(C.byref(self.cfg)))
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.dmfb_vec_create(C.byref(self.cfg), self._stream(), C.byref(self.h)))
        self.obs_len = 3 * fov * fov + 2
        self.max_step = 2 * (width + length)
        self.timing = None  # set to [] to collect (start, end) HIP event pairs around every step launch
        E, n, dev = n_envs, n_agents, self.device
        # outputs of a transition, allocated once and reused every step
        self.obs = torch.zeros((E, n, self.obs_len), dtype=torch.int8, device=dev)
        self.rewards = torch.zeros((E, n), dtype=torch.float64, device=dev)
        self.dones = torch.zeros((E, n), dtype=torch.uint8, device=dev)
        self.constraints = torch.zeros((E,), dtype=torch.int32, device=dev)
        self.success = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self.team_reward = torch.zeros((E,), dtype=torch.float64, device=dev)
        self.terminated = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self._out = _lib.DmfbVecStepOut(self.rewards.data_ptr(), self.dones.data_ptr(), self.constraints.data_ptr(),
                                        self.success.data_ptr(), self.obs.data_ptr(), self.team_reward.data_ptr(),
                                        self.terminated.data_ptr(), None)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.dmfb_vec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def state_bytes(self):
        return int(self.lib.dmfb_vec_state_bytes(self.h))

    def get_env_info(self):
        """DMFBenv.get_env_info (dmfb.py:633-640)."""
        return {'n_actions': 5, 'n_agents': self.n_agents,
                'obs_shape': (3, self.fov, self.fov, 2, self.obs_len), 'episode_limit': self.max_step}

    # ------------------------------------------------------------------ helpers
    def _dev(self, a, dtype):
        if a is None:
            return None
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _mask(self, mask):
        return self._dev(mask, torch.uint8)

    # ------------------------------------------------------------------ episode control
    def reset(self, mask=None, new=False, obs=None):
        """DMFBenv.reset(new) for the masked envs (all when mask is None); returns self.obs with
        the rows of the reset envs refreshed."""
        m = self._mask(mask)
        obs = self.obs if obs is None else obs
        _check(self.lib.dmfb_vec_reset(self.h, _ptr(m), int(bool(new)), _ptr(obs), self._stream()))
        return obs

    def restart(self, mask=None, obs=None):
        m = self._mask(mask)
        obs = self.obs if obs is None else obs
        _check(self.lib.dmfb_vec_restart(self.h, _ptr(m), _ptr(obs), self._stream()))
        return obs

    def set_task(self, starts, ends):
        s = self._dev(starts, torch.int32).reshape(self.n_envs, self.n_agents, 2)
        e = self._dev(ends, torch.int32).reshape(self.n_envs, self.n_agents, 2)
        _check(self.lib.dmfb_vec_set_task(self.h, _ptr(s), _ptr(e), self._stream()))

    def get_task(self):
        s = torch.empty((self.n_envs, self.n_agents, 2), dtype=torch.int32, device=self.device)
        e = torch.empty_like(s)
        _check(self.lib.dmfb_vec_get_task(self.h, _ptr(s), _ptr(e), self._stream()))
        return s, e

    def set_blocks(self, blocks):
        """Obstacle injection: blocks [E, nb, 4] = (x_min, x_max, y_min, y_max) (dmfb.py:34-41)."""
        b = self._dev(blocks, torch.int32).reshape(self.n_envs, -1, 4)
        _check(self.lib.dmfb_vec_set_blocks(self.h, _ptr(b) if b.shape[1] else None, b.shape[1], self._stream()))

    def get_blocks(self):
        nb = C.c_int(0)
        buf = torch.zeros((self.n_envs, max(1, self.n_blocks), 4), dtype=torch.int32, device=self.device)
        _check(self.lib.dmfb_vec_get_blocks(self.h, _ptr(buf), C.byref(nb), self._stream()))
        return buf[:, :nb.value]

    # ------------------------------------------------------------------ transition
    def step(self, actions, uniforms=None, record=True, autoreset=False, active=None, out=None):
        """DMFBenv.step for all envs.  `actions`: int8/int32/int64 tensor [E, n] on the device (or
        anything array-like); `active` (uint8/bool [E], optional) freezes the envs whose entry is 0.  Returns (obs, rewards, dones, info) as device tensors that are
        REUSED by the next call; info = dict(constraints, success, team_reward, terminated)."""
        if not isinstance(actions, torch.Tensor) or actions.device != self.device:
            actions = self._dev(actions, torch.int32)
        if actions.dtype == torch.int64:
            flag = DMFB_ACT_I64
        elif actions.dtype == torch.int8:
            flag = DMFB_ACT_I8
        elif actions.dtype == torch.int32:
            flag = DMFB_ACT_I32
        else:
            actions, flag = actions.to(torch.int32), DMFB_ACT_I32
        actions = actions.contiguous()
        if actions.numel() != self.n_envs * self.n_agents:
            raise RuntimeError('The number of actions is not the same as n_droplets')  # dmfb.py:272-274
        u = self._dev(uniforms, torch.float64)
        flags = flag | (DMFB_STEP_RECORD if record else 0) | (DMFB_STEP_AUTORESET if autoreset else 0)
        act = self._mask(active)
        if self.timing is not None:  # bench.py: HIP events on the launch stream around the kernel
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _check(self.lib.dmfb_vec_step(self.h, _ptr(actions), _ptr(u), _ptr(act), flags, C.byref(out or self._out),
                                      self._stream()))
        if self.timing is not None:
            ev1.record()
            self.timing.append((ev0, ev1))
        info = {'constraints': self.constraints, 'success': self.success, 'team_reward': self.team_reward,
                'terminated': self.terminated}
        return self.obs, self.rewards, self.dones, info

    def observe(self, mask=None, obs=None):
        obs = self.obs if obs is None else obs
        _check(self.lib.dmfb_vec_observe(self.h, _ptr(self._mask(mask)), _ptr(obs), self._stream()))
        return obs

    # ------------------------------------------------------------------ introspection
    def get_state(self):
        E, n, dev = self.n_envs, self.n_agents, self.device
        pos = torch.empty((E, n, 2), dtype=torch.int32, device=dev)
        dist = torch.empty((E, n), dtype=torch.int32, device=dev)
        sc = torch.empty((E,), dtype=torch.int32, device=dev)
        cons = torch.empty((E,), dtype=torch.int64, device=dev)
        _check(self.lib.dmfb_vec_get_state(self.h, _ptr(pos), _ptr(dist), _ptr(sc), _ptr(cons), self._stream()))
        return {'pos': pos, 'dist': dist, 'step_count': sc, 'constraints': cons}

    def get_map(self, which):
        buf = torch.empty((self.n_envs, self.width, self.length), dtype=torch.float64, device=self.device)
        _check(self.lib.dmfb_vec_get_map(self.h, MAPS[which], _ptr(buf), self._stream()))
        return buf

    def set_map(self, which, arr):
        t = self._dev(arr, torch.float64)
        t = t.expand(self.n_envs, self.width, self.length).contiguous()
        _check(self.lib.dmfb_vec_set_map(self.h, MAPS[which], _ptr(t), self._stream()))

    def launch_shape(self):
        """Chips per workgroup of the launches the handle makes (include/dmfb_vec.h: dmfb_vec_launch_shape)."""
        out = (C.c_int32 * 6)()
        _check(self.lib.dmfb_vec_launch_shape(self.h, C.byref(out)))
        return {'fused_tile': out[0], 'observe_tile': out[1], 'split_min_envs': out[2], 'step_only_tile': out[3],
                'observe_workgroups': out[4], 'observe_block': out[5]}

    def observe_timing(self, enable):
        """Start/stop collecting the dispatch time stamps of the observation kernel (dmfb_vec_observe_timing)."""
        _check(self.lib.dmfb_vec_observe_timing(self.h, int(bool(enable))))

    def observe_timing_read(self):
        """(summed kernel duration in microseconds, launches) since the last read; synchronises the host."""
        us, n = C.c_double(0.0), C.c_int(0)
        _check(self.lib.dmfb_vec_observe_timing_read(self.h, C.byref(us), C.byref(n)))
        return us.value, n.value

    def zoom_lut(self):
        out = np.zeros((2, 511), np.int8)
        _check(self.lib.dmfb_vec_zoom_lut(self.h, out.ctypes.data_as(C.c_void_p)))
        return out


class _RoutingManagerView:
    """What callers read from `env.routing_manager` in the reference (evaDegre.py:21 reads
    m_health; the golden harness assigns starts/ends and the maps)."""

    def __init__(self, env):
        self._env = env

    def _map(self, which):
        return self._env._vec.get_map(which)[0].cpu().numpy()

    m_health = property(lambda self: self._map('health'), lambda self, v: self._env._vec.set_map('health', v))
    m_usage = property(lambda self: self._map('usage'), lambda self, v: self._env._vec.set_map('usage', v))
    m_degrade = property(lambda self: self._map('degrade'), lambda self, v: self._env._vec.set_map('degrade', v))

    @property
    def starts(self):
        return self._env._vec.get_task()[0][0].cpu().numpy().astype(int)

    @property
    def ends(self):
        return self._env._vec.get_task()[1][0].cpu().numpy().astype(int)

    @property
    def distances(self):
        return self._env._vec.get_state()['dist'][0].cpu().numpy().astype(int)

    @property
    def blocks(self):
        """(x_min, x_max, y_min, y_max) per block, like the reference's Block objects (dmfb.py:34-41)."""
        return [tuple(int(v) for v in b) for b in self._env._vec.get_blocks()[0].cpu().numpy()]

    @blocks.setter
    def blocks(self, value):
        rows = [(b.x_min, b.x_max, b.y_min, b.y_max) if hasattr(b, 'x_min') else tuple(b) for b in value]
        self._env._vec.set_blocks(np.asarray(rows, np.int32).reshape(1, -1, 4))

    def set_task(self, starts, ends):
        """starts/ends assignment + restartforall (dmfb.py:185-190)."""
        self._env._vec.set_task(np.asarray(starts)[None], np.asarray(ends)[None])

    def getTaskStatus(self):
        return [bool(d == 0) for d in self.distances]


class DMFBenv:
    """Single-chip facade over the HIP path with the reference's protocol
    (env/DMFB/dmfb.py:474-640): same constructor, reset/step/restart/get_env_info, `.agents`,
    `.width/.length`, `.max_step`, `.routing_manager.m_health`.  One chip = a batch of one, so
    every call synchronises; use VecDMFB for throughput."""

    def __init__(self, width, length, n_agents, n_blocks=0, fov=5, stall=True, b_degrade=False,
                 per_degrade=0.1, show=False, savemp4=False, seed=0, with_maps=True, device=None):
        assert width >= 5 and length >= 5
        assert n_agents > 0
        if show or savemp4:
            raise NotImplementedError('rendering is out of scope (SURVEY.md section 2, rows 3-4)')
        self.agents = ['player_{}'.format(i) for i in range(n_agents)]
        self.possible_agents = self.agents[:]
        self.width, self.length = width, length
        self.max_step = (width + length) * 2
        self._vec = VecDMFB(width, length, n_agents, n_blocks, fov, stall, b_degrade, per_degrade, n_envs=1,
                            seed=seed, with_maps=with_maps, device=device)
        self.routing_manager = _RoutingManagerView(self)
        self.rewards = {i: 0. for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        self.constraints = 0

    def _obs_list(self, obs):
        o = obs[0].cpu().numpy()
        return [o[i].copy() for i in range(len(self.agents))]

    def step(self, actions, record=True):
        if isinstance(actions, dict):
            acts = [actions[a] for a in self.agents]
        elif isinstance(actions, list):
            acts = actions
        else:
            raise TypeError('wrong actions')
        if len(acts) != len(self.agents):
            raise RuntimeError('The number of actions is not the same as n_droplets')
        if any(int(a) < 0 or int(a) > 4 for a in acts):
            raise TypeError('action is illegal')
        obs, rewards, dones, info = self._vec.step(np.asarray(acts, np.int32)[None], record=record)
        r = rewards[0].cpu().numpy()
        d = dones[0].cpu().numpy()
        self.step_count += 1
        c = int(info['constraints'][0].item())
        self.constraints += c
        for k, a in enumerate(self.agents):
            self.rewards[a] = np.float64(r[k])
            self.dones[a] = bool(d[k])
        return self._obs_list(obs), self.rewards, self.dones, {'constraints': c, 'success': int(info['success'][0].item())}

    def reset(self, new=False):
        self.rewards = {i: 0 for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        self.constraints = 0
        return self._obs_list(self._vec.reset(new=new))

    def restart(self, index=None):
        self.rewards = {i: 0.0 for i in self.agents}
        self.dones = {i: False for i in self.agents}
        self.step_count = 0
        self.constraints = 0
        return self._obs_list(self._vec.restart())

    def getObs(self):
        return self._obs_list(self._vec.observe())

    def get_env_info(self):
        return self._vec.get_env_info()

    def seed(self, seed=None):
        pass

    def render(self, close=False):
        pass

    def close(self):
        pass
