"""ctypes binding of oracle/meda_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ERRORS = {
    -1: (ValueError, 'bad argument'),
    -3: (RuntimeError, 'Too many droplets in the MEDA array'),   # env/MEDA/meda.py:151-154
    -4: (AssertionError, 'w > 0 and l > 0'),                     # env/MEDA/meda.py:472
    -5: (AssertionError, 'n_agents > 0'),                        # env/MEDA/meda.py:473
    -6: (NotImplementedError, 'configuration outside the build limits'),
}


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, '_build', 'libmeda_oracle.so')
        src = os.path.join(_HERE, 'meda_oracle.c')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['make', '-C', _HERE, '-s'])
        L = C.CDLL(so)
        vp, i32, u32, u64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
        L.meda_oracle_create.argtypes = [i32, i32, i32, i32, i32, f64, i32, u64, i32, u32, i32, C.POINTER(vp)]
        L.meda_oracle_check_cfg.argtypes = [i32, i32, i32]
        for name, args in [('destroy', [vp]), ('reset', [vp, vp]), ('restart', [vp, vp]), ('set_task', [vp, vp, vp]),
                           ('get_task', [vp, vp, vp]), ('get_state', [vp, vp, vp, vp, vp]),
                           ('step', [vp, vp, vp, vp, vp, vp, vp]), ('observe', [vp, vp])]:
            f = getattr(L, 'meda_oracle_' + name)
            f.argtypes = args
            f.restype = None
        L.meda_oracle_get_map.argtypes = [vp, i32, vp]
        L.meda_oracle_set_map.argtypes = [vp, i32, vp]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        exc, msg = ERRORS.get(rc, (RuntimeError, 'oracle error %d' % rc))
        raise exc(msg)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MedaOracle:
    MAP = {'health': 0, 'usage': 1, 'degrade': 2}

    def __init__(self, width, length, n_agents, fov=19, b_degrade=False, per_degrade=0.1, n_envs=1, seed=0,
                 with_maps=False, env_id0=0, version=0):
        self.W, self.L, self.n, self.fov, self.E = width, length, n_agents, fov, n_envs
        self.h = C.c_void_p()
        _check(lib().meda_oracle_create(width, length, n_agents, fov, int(b_degrade), float(per_degrade),
                                        int(with_maps), seed, n_envs, env_id0, int(version), C.byref(self.h)))
        self.max_step = width + length
        self.obs_len = (3 if version == 2 else 4) * fov * fov + 2

    def __del__(self):
        if getattr(self, 'h', None) is not None and self.h:
            lib().meda_oracle_destroy(self.h)
            self.h = None

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().meda_oracle_reset(self.h, _p(m))

    def restart(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().meda_oracle_restart(self.h, _p(m))

    def set_task(self, starts, ends):
        s = np.ascontiguousarray(starts, dtype=np.int32).reshape(self.E, self.n, 2)
        e = np.ascontiguousarray(ends, dtype=np.int32).reshape(self.E, self.n, 2)
        lib().meda_oracle_set_task(self.h, _p(s), _p(e))

    def get_task(self):
        s = np.zeros((self.E, self.n, 2), np.int32)
        e = np.zeros((self.E, self.n, 2), np.int32)
        lib().meda_oracle_get_task(self.h, _p(s), _p(e))
        return s, e

    def get_state(self):
        pos = np.zeros((self.E, self.n, 2), np.int32)
        status = np.zeros((self.E, self.n), np.uint8)
        sc = np.zeros(self.E, np.int32)
        failed = np.zeros(self.E, np.uint8)
        lib().meda_oracle_get_state(self.h, _p(pos), _p(status), _p(sc), _p(failed))
        return dict(pos=pos, status=status, step_count=sc, failed=failed)

    def get_map(self, which):
        buf = np.zeros((self.E, self.W, self.L), np.float64)
        _check(lib().meda_oracle_get_map(self.h, self.MAP[which], _p(buf)))
        return buf

    def set_map(self, which, arr):
        buf = np.ascontiguousarray(np.broadcast_to(np.asarray(arr, np.float64), (self.E, self.W, self.L)))
        _check(lib().meda_oracle_set_map(self.h, self.MAP[which], _p(buf)))

    def step(self, actions, uniforms=None):
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.E, self.n)
        u = None if uniforms is None else np.ascontiguousarray(uniforms, dtype=np.float64).reshape(self.E, self.n)
        rewards = np.zeros((self.E, self.n), np.float64)
        dones = np.zeros((self.E, self.n), np.uint8)
        fail = np.zeros(self.E, np.float64)
        succ = np.zeros(self.E, np.uint8)
        lib().meda_oracle_step(self.h, _p(a), _p(u), _p(rewards), _p(dones), _p(fail), _p(succ))
        return rewards, dones, fail, succ

    def observe(self):
        obs = np.zeros((self.E, self.n, self.obs_len), np.int8)
        lib().meda_oracle_observe(self.h, _p(obs))
        return obs
