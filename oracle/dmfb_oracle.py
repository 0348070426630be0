"""ctypes binding of oracle/dmfb_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py).

`DmfbOracle` steps E independent chips one after another on the CPU, exactly as the
reference's per-droplet Python loops do (env/DMFB/dmfb.py); it exists to check the HIP
kernels and to serve as bench.py's `cpu_baseline` (kind "port").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERRORS = {
    -1: (ValueError, 'bad argument'),
    -2: (RuntimeError, 'Fov is too large'),             # env/DMFB/dmfb.py:139-140
    -3: (TypeError, 'Too many droplets for DMFB'),       # env/DMFB/dmfb.py:144-146
    -4: (AssertionError, 'width >= 5 and length >= 5'),  # env/DMFB/dmfb.py:489
    -5: (AssertionError, 'n_agents > 0'),                # env/DMFB/dmfb.py:490
    -6: (NotImplementedError, 'configuration outside the build limits'),
    -7: (TypeError, 'action is illegal'),                # env/DMFB/dmfb.py:116
}


def build(force=False):
    so = os.path.join(_HERE, '_build', 'libdmfb_oracle.so')
    src = os.path.join(_HERE, 'dmfb_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, u32, u64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
        L.dmfb_oracle_create.argtypes = [i32, i32, i32, i32, i32, i32, i32, f64, i32, u64, i32, u32, C.POINTER(vp)]
        L.dmfb_oracle_create.restype = i32
        L.dmfb_oracle_destroy.argtypes = [vp]
        L.dmfb_oracle_destroy.restype = None
        L.dmfb_oracle_check_cfg.argtypes = [i32] * 5
        L.dmfb_oracle_reset.argtypes = [vp, vp, i32]
        L.dmfb_oracle_reset.restype = None
        L.dmfb_oracle_restart.argtypes = [vp, vp]
        L.dmfb_oracle_restart.restype = None
        L.dmfb_oracle_set_task.argtypes = [vp, vp, vp]
        L.dmfb_oracle_set_task.restype = None
        L.dmfb_oracle_set_blocks.argtypes = [vp, vp, i32]
        L.dmfb_oracle_get_blocks.argtypes = [vp, vp, i32]
        L.dmfb_oracle_get_task.argtypes = [vp, vp, vp]
        L.dmfb_oracle_get_task.restype = None
        L.dmfb_oracle_get_state.argtypes = [vp, vp, vp, vp, vp]
        L.dmfb_oracle_get_state.restype = None
        L.dmfb_oracle_get_map.argtypes = [vp, i32, vp]
        L.dmfb_oracle_set_map.argtypes = [vp, i32, vp]
        L.dmfb_oracle_step.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp]
        L.dmfb_oracle_observe.argtypes = [vp, vp]
        L.dmfb_oracle_observe.restype = None
        L.dmfb_oracle_zoom_lut.argtypes = [vp, vp]
        L.dmfb_oracle_zoom_lut.restype = None
        L.dmfb_oracle_philox.argtypes = [u32, u32, vp, vp]
        L.dmfb_oracle_philox.restype = None
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        exc, msg = ERRORS.get(rc, (RuntimeError, 'oracle error %d' % rc))
        raise exc(msg)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def philox(k0, k1, ctr):
    c = np.asarray(ctr, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().dmfb_oracle_philox(k0, k1, _p(c), _p(out))
    return out


class DmfbOracle:
    MAP = {'health': 0, 'usage': 1, 'degrade': 2}

    def __init__(self, width, length, n_agents, n_blocks=0, fov=5, stall=True, b_degrade=False,
                 per_degrade=0.1, n_envs=1, seed=0, with_maps=False, env_id0=0):
        self.W, self.L, self.n, self.fov, self.E = width, length, n_agents, fov, n_envs
        self.n_blocks = n_blocks
        self.h = C.c_void_p()
        _check(lib().dmfb_oracle_create(width, length, n_agents, n_blocks, fov, int(stall), int(b_degrade),
                                        float(per_degrade), int(with_maps), seed, n_envs, env_id0,
                                        C.byref(self.h)))
        self.max_step = 2 * (width + length)
        self.obs_len = 3 * fov * fov + 2

    def __del__(self):
        if getattr(self, 'h', None) is not None and self.h:
            lib().dmfb_oracle_destroy(self.h)
            self.h = None

    def reset(self, mask=None, new=False):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().dmfb_oracle_reset(self.h, _p(m), int(new))

    def restart(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().dmfb_oracle_restart(self.h, _p(m))

    def set_task(self, starts, ends):
        s = np.ascontiguousarray(starts, dtype=np.int32).reshape(self.E, self.n, 2)
        e = np.ascontiguousarray(ends, dtype=np.int32).reshape(self.E, self.n, 2)
        lib().dmfb_oracle_set_task(self.h, _p(s), _p(e))

    def set_blocks(self, blocks):
        b = np.ascontiguousarray(blocks, dtype=np.int32).reshape(self.E, -1, 4)
        _check(lib().dmfb_oracle_set_blocks(self.h, _p(b), b.shape[1]))

    def get_blocks(self):
        cap = max(1, self.n_blocks)
        b = np.zeros((self.E, cap, 4), np.int32)
        nb = lib().dmfb_oracle_get_blocks(self.h, _p(b), cap)
        return b[:, :nb]

    def get_task(self):
        s = np.zeros((self.E, self.n, 2), np.int32)
        e = np.zeros((self.E, self.n, 2), np.int32)
        lib().dmfb_oracle_get_task(self.h, _p(s), _p(e))
        return s, e

    def get_state(self):
        pos = np.zeros((self.E, self.n, 2), np.int32)
        dist = np.zeros((self.E, self.n), np.int32)
        sc = np.zeros(self.E, np.int32)
        cons = np.zeros(self.E, np.int64)
        lib().dmfb_oracle_get_state(self.h, _p(pos), _p(dist), _p(sc), _p(cons))
        return dict(pos=pos, dist=dist, step_count=sc, constraints=cons)

    def get_map(self, which):
        buf = np.zeros((self.E, self.W, self.L), np.float64)
        _check(lib().dmfb_oracle_get_map(self.h, self.MAP[which], _p(buf)))
        return buf

    def set_map(self, which, arr):
        buf = np.ascontiguousarray(np.broadcast_to(np.asarray(arr, np.float64), (self.E, self.W, self.L)))
        _check(lib().dmfb_oracle_set_map(self.h, self.MAP[which], _p(buf)))

    def step(self, actions, uniforms=None, record=True):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.n)
        u = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float64).reshape(self.E, self.n)
        rewards = np.zeros((self.E, self.n), np.float64)
        dones = np.zeros((self.E, self.n), np.uint8)
        cons = np.zeros(self.E, np.int32)
        succ = np.zeros(self.E, np.uint8)
        _check(lib().dmfb_oracle_step(self.h, _p(a), _p(u), int(record), _p(rewards), _p(dones), _p(cons), _p(succ)))
        return rewards, dones, cons, succ

    def observe(self):
        obs = np.zeros((self.E, self.n, self.obs_len), np.int8)
        lib().dmfb_oracle_observe(self.h, _p(obs))
        return obs

    def zoom_lut(self):
        out = np.zeros((2, 511), np.int8)
        lib().dmfb_oracle_zoom_lut(self.h, _p(out))
        return out
