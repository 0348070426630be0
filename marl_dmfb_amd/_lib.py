"""ctypes loader for the HIP libraries built in-tree (marl_dmfb_amd/lib/*.so).

Fails loudly when a library is missing: the product path has no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, 'lib')
_CACHE = {}


class HipLibraryMissing(RuntimeError):
    pass


def load(name):
    if name in _CACHE:
        return _CACHE[name]
    # MARL_DMFB_VARIANT_<NAME>=_tag loads lib<name>_tag.so instead: same-box A/B timing of a kernel variant (development aid)
    path = os.path.join(LIB_DIR, 'lib%s%s.so' % (name, os.environ.get('MARL_DMFB_VARIANT_' + name.upper(), '')))
    if not os.path.exists(path):
        raise HipLibraryMissing(
            '%s not found: build the HIP extension first (python -c "import __graft_entry__ as g; g.build()" '
            'or make -C marl_dmfb_amd/csrc). There is no CPU fallback.' % path)
    # PyTorch bundles its own HIP runtime (SONAME libamdhip64.so.7).  It must be in the process
    # BEFORE our library is opened, so that the library's NEEDED entry binds to that same runtime:
    # two HIP runtimes in one process cannot share devices, streams or pointers.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    _CACHE[name] = lib
    return lib


class DmfbVecConfig(C.Structure):
    """include/dmfb_vec.h: dmfb_vec_config"""
    _fields_ = [('width', C.c_int32), ('length', C.c_int32), ('n_agents', C.c_int32), ('n_blocks', C.c_int32),
                ('fov', C.c_int32), ('stall', C.c_int32), ('b_degrade', C.c_int32), ('with_maps', C.c_int32),
                ('per_degrade', C.c_double), ('n_envs', C.c_int32), ('env_id0', C.c_uint32), ('seed', C.c_uint64),
                ('device', C.c_int32)]


class DmfbVecStepOut(C.Structure):
    """include/dmfb_vec.h: dmfb_vec_step_out"""
    _fields_ = [('d_rewards', C.c_void_p), ('d_dones', C.c_void_p), ('d_constraints', C.c_void_p),
                ('d_success', C.c_void_p), ('d_obs', C.c_void_p), ('d_team_reward', C.c_void_p),
                ('d_terminated', C.c_void_p), ('d_obs_terminal', C.c_void_p)]


DMFB_VEC_SYMBOLS = [
    'dmfb_vec_check_config', 'dmfb_vec_create', 'dmfb_vec_destroy', 'dmfb_vec_state_bytes', 'dmfb_vec_obs_len',
    'dmfb_vec_max_step', 'dmfb_vec_n_envs', 'dmfb_vec_n_agents', 'dmfb_vec_reset', 'dmfb_vec_restart',
    'dmfb_vec_set_task', 'dmfb_vec_get_task', 'dmfb_vec_set_blocks', 'dmfb_vec_get_blocks', 'dmfb_vec_step', 'dmfb_vec_observe', 'dmfb_vec_get_state',
    'dmfb_vec_get_map', 'dmfb_vec_set_map', 'dmfb_vec_launch_shape', 'dmfb_vec_observe_timing', 'dmfb_vec_observe_timing_read', 'dmfb_vec_zoom_lut', 'dmfb_vec_strerror', 'dmfb_vec_last_hip_error',
]


def dmfb_vec():
    lib = load('dmfb_vec')
    if getattr(lib, '_typed', False):
        return lib
    vp, i32, u32 = C.c_void_p, C.c_int, C.c_uint32
    cfgp = C.POINTER(DmfbVecConfig)
    lib.dmfb_vec_check_config.argtypes = [cfgp]
    lib.dmfb_vec_create.argtypes = [cfgp, vp, C.POINTER(vp)]
    lib.dmfb_vec_destroy.argtypes = [vp]
    lib.dmfb_vec_state_bytes.argtypes = [vp]
    lib.dmfb_vec_state_bytes.restype = C.c_size_t
    for f in ('dmfb_vec_obs_len', 'dmfb_vec_max_step', 'dmfb_vec_n_envs', 'dmfb_vec_n_agents'):
        getattr(lib, f).argtypes = [vp]
    lib.dmfb_vec_reset.argtypes = [vp, vp, i32, vp, vp]
    lib.dmfb_vec_restart.argtypes = [vp, vp, vp, vp]
    lib.dmfb_vec_set_task.argtypes = [vp, vp, vp, vp]
    lib.dmfb_vec_get_task.argtypes = [vp, vp, vp, vp]
    lib.dmfb_vec_set_blocks.argtypes = [vp, vp, i32, vp]
    lib.dmfb_vec_get_blocks.argtypes = [vp, vp, C.POINTER(C.c_int), vp]
    lib.dmfb_vec_step.argtypes = [vp, vp, vp, vp, u32, C.POINTER(DmfbVecStepOut), vp]
    lib.dmfb_vec_observe.argtypes = [vp, vp, vp, vp]
    lib.dmfb_vec_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dmfb_vec_get_map.argtypes = [vp, i32, vp, vp]
    lib.dmfb_vec_set_map.argtypes = [vp, i32, vp, vp]
    lib.dmfb_vec_zoom_lut.argtypes = [vp, vp]
    lib.dmfb_vec_launch_shape.argtypes = [vp, C.POINTER(C.c_int32 * 6)]
    lib.dmfb_vec_observe_timing.argtypes = [vp, i32]
    lib.dmfb_vec_observe_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.dmfb_vec_strerror.argtypes = [i32]
    lib.dmfb_vec_strerror.restype = C.c_char_p
    lib.dmfb_vec_last_hip_error.argtypes = []
    lib._typed = True
    return lib


class MedaVecConfig(C.Structure):
    """include/meda_vec.h: meda_vec_config"""
    _fields_ = [('width', C.c_int32), ('length', C.c_int32), ('n_agents', C.c_int32), ('fov', C.c_int32),
                ('b_degrade', C.c_int32), ('with_maps', C.c_int32), ('per_degrade', C.c_double), ('n_envs', C.c_int32),
                ('env_id0', C.c_uint32), ('seed', C.c_uint64), ('device', C.c_int32), ('obs_version', C.c_int32)]


class MedaVecStepOut(C.Structure):
    """include/meda_vec.h: meda_vec_step_out"""
    _fields_ = [('d_rewards', C.c_void_p), ('d_dones', C.c_void_p), ('d_fail', C.c_void_p), ('d_success', C.c_void_p),
                ('d_obs', C.c_void_p), ('d_team_reward', C.c_void_p), ('d_terminated', C.c_void_p)]


MEDA_VEC_SYMBOLS = [
    'meda_vec_check_config', 'meda_vec_create', 'meda_vec_destroy', 'meda_vec_state_bytes', 'meda_vec_obs_len',
    'meda_vec_max_step', 'meda_vec_n_envs', 'meda_vec_n_agents', 'meda_vec_reset', 'meda_vec_restart',
    'meda_vec_set_task', 'meda_vec_get_task', 'meda_vec_step', 'meda_vec_observe', 'meda_vec_get_state',
    'meda_vec_get_map', 'meda_vec_set_map', 'meda_vec_launch_shape', 'meda_vec_observe_timing', 'meda_vec_observe_timing_read', 'meda_vec_strerror', 'meda_vec_last_hip_error',
]


def meda_vec():
    lib = load('meda_vec')
    if getattr(lib, '_typed', False):
        return lib
    vp, i32, u32 = C.c_void_p, C.c_int, C.c_uint32
    cfgp = C.POINTER(MedaVecConfig)
    lib.meda_vec_check_config.argtypes = [cfgp]
    lib.meda_vec_create.argtypes = [cfgp, vp, C.POINTER(vp)]
    lib.meda_vec_destroy.argtypes = [vp]
    lib.meda_vec_state_bytes.argtypes = [vp]
    lib.meda_vec_state_bytes.restype = C.c_size_t
    for f in ('meda_vec_obs_len', 'meda_vec_max_step', 'meda_vec_n_envs', 'meda_vec_n_agents'):
        getattr(lib, f).argtypes = [vp]
    lib.meda_vec_reset.argtypes = [vp, vp, vp, vp]
    lib.meda_vec_restart.argtypes = [vp, vp, vp, vp]
    lib.meda_vec_set_task.argtypes = [vp, vp, vp, vp]
    lib.meda_vec_get_task.argtypes = [vp, vp, vp, vp]
    lib.meda_vec_step.argtypes = [vp, vp, vp, vp, u32, C.POINTER(MedaVecStepOut), vp]
    lib.meda_vec_observe.argtypes = [vp, vp, vp, vp]
    lib.meda_vec_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.meda_vec_get_map.argtypes = [vp, i32, vp, vp]
    lib.meda_vec_set_map.argtypes = [vp, i32, vp, vp]
    lib.meda_vec_launch_shape.argtypes = [vp, C.POINTER(C.c_int32 * 4)]
    lib.meda_vec_observe_timing.argtypes = [vp, C.c_int]
    lib.meda_vec_observe_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.meda_vec_strerror.argtypes = [i32]
    lib.meda_vec_strerror.restype = C.c_char_p
    lib.meda_vec_last_hip_error.argtypes = []
    lib._typed = True
    return lib


def crnn_ops():
    """include/crnn_ops.h"""
    lib = load('crnn_ops')
    if getattr(lib, '_typed', False):
        return lib
    vp, i64 = C.c_void_p, C.c_int64
    lib.crnn_conv9_forward.argtypes = [vp, i64, i64, vp, vp, vp, vp, C.c_int, vp, i64, vp]
    lib.crnn_front9_forward.argtypes = [vp, i64, vp, C.c_int, i64, vp, vp, vp, vp, vp, vp, C.c_int, vp, i64, C.c_int, vp]
    lib.crnn_front9_forward_live.argtypes = [vp, i64, vp, C.c_int, i64, vp, vp, vp, vp, vp, vp, C.c_int, vp, i64, C.c_int, vp, vp, C.c_int, vp]
    lib.crnn_front19_forward.argtypes = [vp, i64, vp, C.c_int, i64, vp, vp, vp, vp, vp, vp, C.c_int, vp, i64, C.c_int, vp]
    lib.crnn_front_padded_cols.argtypes = [C.c_int]
    lib.crnn_conv9_backward_parts.argtypes = [C.c_int]
    lib.crnn_conv9_backward.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
    lib.crnn_mlp_backward_parts.argtypes = []
    lib.crnn_mlp_backward.argtypes = [vp, i64, C.c_int, vp, C.c_int, i64, vp, i64, vp, i64, C.c_int, vp, vp, vp, vp]
    lib.crnn_conv19_backward_parts.argtypes = [C.c_int]
    lib.crnn_conv19_backward.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
    lib.crnn_last_hip_error.argtypes = []
    lib.gru_seq_forward.argtypes = [vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, vp, vp, vp]
    lib.gru_seq_backward.argtypes = [vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, vp, vp, vp, vp, vp]
    ip = C.POINTER(C.c_int32)
    lib.gru_seq_forward_packed.argtypes = [vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, ip, vp, vp, vp]
    lib.gru_seq_forward_packed_pair.argtypes = [vp] * 14 + [C.c_int, i64, C.c_int, ip, vp]
    lib.gru_seq_backward_packed.argtypes = [vp, vp, vp, vp, vp, C.c_int, i64, C.c_int, ip, vp, vp, vp, vp, vp, vp]
    lib.gru_seq_row_blocks.argtypes = [i64]
    lib.gru_seq_row_blocks.restype = i64
    lib.gru_last_hip_error.argtypes = []
    lib._typed = True
    return lib


class RolloutStage(C.Structure):
    """include/rollout_ops.h: rollout_stage"""
    _fields_ = [('d_t_ep', C.c_void_p), ('d_o0', C.c_void_p), ('d_o_next', C.c_void_p), ('d_u', C.c_void_p),
                ('d_onehot', C.c_void_p), ('d_r', C.c_void_p), ('d_ep_acc', C.c_void_p), ('d_chip_acc', C.c_void_p),
                ('d_close_slot', C.c_void_p), ('d_state_alt', C.c_void_p)]


class RolloutRing(C.Structure):
    """include/rollout_ops.h: rollout_ring"""
    _fields_ = [('slots', C.c_int32), ('d_o', C.c_void_p), ('d_o_next', C.c_void_p), ('d_u', C.c_void_p),
                ('d_u_onehot', C.c_void_p), ('d_avail_u', C.c_void_p), ('d_avail_u_next', C.c_void_p), ('d_r', C.c_void_p),
                ('d_padded', C.c_void_p), ('d_terminated', C.c_void_p), ('d_len', C.c_void_p), ('d_stats', C.c_void_p),
                ('d_state', C.c_void_p)]


def rollout_ops():
    """include/rollout_ops.h"""
    lib = load('rollout_ops')
    if getattr(lib, '_typed', False):
        return lib
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float
    lib.rollout_select_actions.argtypes = [vp, i32, i32, i32, vp, i32, u64, vp, vp, vp, vp, vp, i32, i32, vp]
    lib.rollout_gru_head_select.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, u64, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.rollout_gru_head_select_live.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, u64, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.rollout_compact_alive.argtypes = [i32, vp, vp, vp, vp]
    lib.rollout_post_step.argtypes = [i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, i32, vp, vp, vp]
    lib.rollout_gru_head_select_stream.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, u64, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    ringp, stagep = C.POINTER(RolloutRing), C.POINTER(RolloutStage)
    lib.rollout_stream_step.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, stagep, ringp, i32, vp, vp, vp, f32, f32, vp, vp]
    lib.rollout_last_hip_error.argtypes = []
    lib._typed = True
    return lib


def vdn_ops():
    """include/vdn_ops.h"""
    lib = load('vdn_ops')
    if getattr(lib, '_typed', False):
        return lib
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.vdn_td_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp]
    lib.vdn_td_backward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.vdn_td_forward_packed.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]
    lib.vdn_td_backward_packed.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp]
    lib.vdn_gather_units.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp]
    pp, pl = C.POINTER(C.c_void_p), C.POINTER(C.c_int64)
    f64 = C.c_double
    lib.vdn_clip_adam_step.argtypes = [i32, pp, pp, pp, pp, pl, f32, f64, f64, f64, f64, f64, f64, vp, vp, vp, vp]
    lib.vdn_last_hip_error.argtypes = []
    lib._typed = True
    return lib
