"""CPU-side checks of the C ABI: the in-tree HIP library loads without a GPU, exports every
symbol include/dmfb_vec.h declares, and the config guards return the reference's error classes."""
import ctypes as C
import os
import re

import pytest

from marl_dmfb_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    prefix = header.split('.')[0]
    return sorted(set(re.findall(r'\b(%s_[a-z_0-9]+)\s*\(' % prefix, txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.dmfb_vec()
    names = _declared('dmfb_vec.h')
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.DMFB_VEC_SYMBOLS) == names


def _cfg(**kw):
    base = dict(width=10, length=10, n_agents=4, n_blocks=0, fov=9, stall=1, b_degrade=0, with_maps=0,
                per_degrade=0.1, n_envs=8, env_id0=0, seed=0, device=0)
    base.update(kw)
    return _lib.DmfbVecConfig(**base)


@pytest.mark.parametrize('kw,code', [
    ({}, 0),
    ({'fov': 11}, -2),                      # RuntimeError('Fov is too large')        dmfb.py:139-140
    ({'n_agents': 14}, -3),                 # TypeError('Too many droplets for DMFB') dmfb.py:144-146
    ({'width': 4}, -4),                     # assert width >= 5                        dmfb.py:489
    ({'n_agents': 0}, -5),                  # assert n_agents > 0                      dmfb.py:490
    ({'width': 50, 'length': 50, 'n_agents': 17}, -6),
    ({'n_envs': 0}, -1),
])
def test_check_config_codes(kw, code):
    lib = _lib.dmfb_vec()
    c = _cfg(**kw)
    assert lib.dmfb_vec_check_config(C.byref(c)) == code


def test_strerror():
    lib = _lib.dmfb_vec()
    assert lib.dmfb_vec_strerror(-2) == b'Fov is too large'
    assert lib.dmfb_vec_strerror(-3) == b'Too many droplets for DMFB'


def test_meda_library_exports_every_declared_symbol():
    lib = _lib.meda_vec()
    names = _declared('meda_vec.h')
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.MEDA_VEC_SYMBOLS) == names


def test_meda_check_config_codes():
    lib = _lib.meda_vec()

    def cfg(**kw):
        base = dict(width=30, length=30, n_agents=4, fov=19, b_degrade=0, with_maps=0, per_degrade=0.1, n_envs=8,
                    env_id0=0, seed=0, device=0)
        base.update(kw)
        return _lib.MedaVecConfig(**base)
    assert lib.meda_vec_check_config(C.byref(cfg())) == 0
    assert lib.meda_vec_check_config(C.byref(cfg(width=10, length=10))) == -3   # meda.py:151-154
    assert lib.meda_vec_check_config(C.byref(cfg(n_agents=5))) == -3
    assert lib.meda_vec_check_config(C.byref(cfg(n_agents=0))) == -5
    assert lib.meda_vec_check_config(C.byref(cfg(width=0))) == -4


def test_crnn_ops_library_exports():
    lib = _lib.crnn_ops()
    txt = open(os.path.join(ROOT, 'include', 'crnn_ops.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = sorted(set(re.findall(r'\b(crnn_[a-z_0-9]+)\s*\(', txt)))
    assert names == ['crnn_conv19_backward', 'crnn_conv19_backward_parts', 'crnn_conv9_backward', 'crnn_conv9_backward_parts', 'crnn_conv9_forward', 'crnn_front19_forward',
                     'crnn_front9_forward', 'crnn_front9_forward_live',
                     'crnn_front_padded_cols', 'crnn_last_hip_error', 'crnn_mlp_backward', 'crnn_mlp_backward_parts']
    gru = sorted(set(re.findall(r'\b(gru_[a-z_0-9]+)\s*\(', txt)))
    assert gru == ['gru_last_hip_error', 'gru_seq_backward', 'gru_seq_backward_packed', 'gru_seq_forward', 'gru_seq_forward_packed', 'gru_seq_forward_packed_pair',
                   'gru_seq_row_blocks']
    assert lib.gru_seq_forward_packed(None, None, None, None, None, 4, 8, 128, None, None, None, None) == -1
    for n in names + gru:
        assert hasattr(lib, n)
    # argument guards run on the host before anything touches the GPU
    assert lib.crnn_conv9_forward(None, 245, 4, None, None, None, None, 24, None, 600, None) == -1
    assert lib.gru_seq_forward(None, None, None, None, None, 4, 8, 128, None, None, None) == -1
    assert lib.crnn_front19_forward(None, 1085, None, 9, 4, None, None, None, None, None, None, 32, None, 810, 0, None) == -1
    assert lib.crnn_front_padded_cols(24) == 640 and lib.crnn_front_padded_cols(32) == 832 and lib.crnn_front_padded_cols(16) < 0


def test_rollout_ops_library_exports():
    lib = _lib.rollout_ops()
    txt = open(os.path.join(ROOT, 'include', 'rollout_ops.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = sorted(set(re.findall(r'\b(rollout_[a-z_0-9]+)\s*\(', txt)))
    assert names == ['rollout_compact_alive', 'rollout_gru_head_select', 'rollout_gru_head_select_live', 'rollout_gru_head_select_stream',
                     'rollout_last_hip_error', 'rollout_post_step', 'rollout_select_actions', 'rollout_stream_step']
    assert lib.rollout_stream_step(4, 2, 5, 10, 490, 128, None, None, None, None, None, None, 0, None, None, None, 0, None, None, None, 0.0, 0.0, None, None) == -1
    assert lib.rollout_compact_alive(4, None, None, None, None) == -1
    for n in names:
        assert hasattr(lib, n)
    assert lib.rollout_select_actions(None, 4, 2, 5, None, 1, 0, None, None, None, None, None, 10, 0, None) == -1


def test_vdn_ops_library_exports():
    lib = _lib.vdn_ops()
    txt = open(os.path.join(ROOT, 'include', 'vdn_ops.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = sorted(set(re.findall(r'\b(vdn_[a-z_0-9]+)\s*\(', txt)))
    assert names == ['vdn_clip_adam_step', 'vdn_gather_units', 'vdn_last_hip_error', 'vdn_td_backward', 'vdn_td_backward_packed', 'vdn_td_forward',
                     'vdn_td_forward_packed']
    assert lib.vdn_gather_units(None, 980, None, 4, 0, 0, None, None) == -1
    assert lib.vdn_td_forward_packed(None, None, None, 4, None, None, None, None, None, 4, 5, 0.99, None, None, None, None) == -1
    for n in names:
        assert hasattr(lib, n)
    # argument guards run on the host before anything touches the GPU
    assert lib.vdn_td_forward(None, None, None, None, None, None, None, 4, 3, 8, 2, 5, 0.99, None, None, None, None) == -1
    assert lib.vdn_clip_adam_step(0, None, None, None, None, None, 10.0, 1e-3, 0.9, 0.99, 1e-8, 0.1, 0.01, None, None, None, None) == -1
    assert lib.vdn_clip_adam_step(33, None, None, None, None, None, 10.0, 1e-3, 0.9, 0.99, 1e-8, 0.1, 0.01, None, None, None, None) == -1
    assert lib.vdn_td_backward(None, None, None, None, 4, 3, 8, 2, 5, None, None) == -1
