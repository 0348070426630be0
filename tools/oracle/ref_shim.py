"""Container-only harness: import the Python reference from /root/reference.

TEST INFRASTRUCTURE, not product code.  Nothing under marl_dmfb_amd/, bench.py or
__graft_entry__.py imports this module; it is used only by the golden-vector
generators in this directory.

The reference imports a few third-party packages that are absent from this image
(gym, pettingzoo, cv2) but never uses them on the env/policy arithmetic path:
spaces are only constructed in the env constructors, ParallelEnv is only a base
class, cv2 is only touched when savemp4=True.  They are registered here as inert
module objects so that the reference's own source files can be imported unchanged
from where they lie.  No reference source is copied.
"""
import random
import sys
import types
import warnings

REFERENCE_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AnySpace:
    def __init__(self, *a, **k):
        pass


class _ParallelEnv:
    pass


def install():
    """Register inert stand-ins for absent third-party modules, then put the
    reference root on sys.path.  Idempotent."""
    if getattr(install, 'done', False):
        return
    import numpy as np
    import typing_extensions
    warnings.filterwarnings('ignore')
    if 'gym' not in sys.modules:
        spaces = _mod('gym.spaces', Discrete=_AnySpace, Box=_AnySpace)
        wrappers = _mod('gym.wrappers')
        seeding = _mod('gym.utils.seeding')
        utils = _mod('gym.utils', seeding=seeding)
        error = _mod('gym.error')
        _mod('gym', spaces=spaces, wrappers=wrappers, utils=utils, error=error)
    if 'pettingzoo' not in sys.modules:
        env = _mod('pettingzoo.utils.env', ParallelEnv=_ParallelEnv)
        u = _mod('pettingzoo.utils', env=env)
        _mod('pettingzoo', utils=u)
    if 'cv2' not in sys.modules:
        _mod('cv2')
    if 'numpy.lib.function_base' not in sys.modules:
        _mod('numpy.lib.function_base', select=np.select)
    if not hasattr(typing_extensions, 'runtime'):
        typing_extensions.runtime = typing_extensions.runtime_checkable
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install.done = True


class DrawQueue:
    """Replacement for the global `random.random` the reference envs call
    (env/DMFB/dmfb.py:335, env/MEDA/meda.py:280): pops injected float64 draws and
    records how many were consumed."""

    def __init__(self):
        self.q = []
        self.consumed = 0

    def feed(self, values):
        self.q = list(values)
        self.consumed = 0

    def __call__(self):
        self.consumed += 1
        return self.q.pop(0)


def patch_random(queue):
    """Route `random.random()` to `queue` (the env modules call it through the
    module attribute, so rebinding it is enough)."""
    random.random = queue
