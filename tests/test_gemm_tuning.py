"""common/gemm_tuning.py decision table without a GPU: the shipped rocBLAS / hipBLASLt solution file is used read-only, a file
whose validators do not match (another ROCm build) or that cannot be read falls back to the library defaults WITH a stated
reason, MARL_DMFB_GEMM_TUNING=0 switches it off, and nothing is ever written in the read-only mode."""
import pytest

from marl_dmfb_amd.common import gemm_tuning


class _Tun:
    def __init__(self, read_ok=True, raises=False):
        self.calls, self.read_ok, self.raises = [], read_ok, raises

    def enable(self, v):
        self.calls.append(('enable', v))

    def tuning_enable(self, v):
        self.calls.append(('tuning_enable', v))

    def write_file_on_exit(self, v):
        self.calls.append(('write_file_on_exit', v))

    def set_filename(self, name, insert_device_ordinal=False):
        self.calls.append(('set_filename', name))

    def read_file(self, name):
        self.calls.append(('read_file', name))
        if self.raises:
            raise OSError('boom')
        return self.read_ok


def test_shipped_file_read_only(tmp_path):
    f = tmp_path / 'r.csv'
    f.write_text('Validator,PT_VERSION,2.10.0\n')
    t = _Tun()
    ok, mode = gemm_tuning._decide(t, True, env={}, results=str(f))
    assert ok and mode == 'tuned (shipped choices)'
    assert ('tuning_enable', False) in t.calls and ('write_file_on_exit', False) in t.calls
    assert not any(c[0] == 'set_filename' for c in t.calls)          # no redirected output file either
    del _Tun.write_file_on_exit                                      # torch builds without that switch: output goes to os.devnull
    try:
        t = _Tun()
        assert gemm_tuning._decide(t, True, env={}, results=str(f))[0]
        import os
        assert ('set_filename', os.devnull) in t.calls
    finally:
        _Tun.write_file_on_exit = lambda self, v: self.calls.append(('write_file_on_exit', v))


@pytest.mark.parametrize('tun, why', [(_Tun(read_ok=False), 'validators'), (_Tun(raises=True), 'unusable')])
def test_bad_file_falls_back_with_reason(tmp_path, tun, why):
    f = tmp_path / 'r.csv'
    f.write_text('garbage')
    ok, mode = gemm_tuning._decide(tun, True, env={}, results=str(f))
    assert not ok and mode.startswith('library default') and why in mode
    assert tun.calls[-1] == ('enable', False)                        # TunableOp switched off again


def test_switch_missing_file_and_no_gpu(tmp_path):
    t = _Tun()
    assert gemm_tuning._decide(t, True, env={'MARL_DMFB_GEMM_TUNING': '0'}, results=str(tmp_path / 'x'))[1] == \
        'library default (MARL_DMFB_GEMM_TUNING=0)'
    assert gemm_tuning._decide(t, True, env={}, results=str(tmp_path / 'missing.csv')) == (False, 'library default (no results file)')
    assert gemm_tuning._decide(None, False, env={}) == (False, 'library default (no GPU)')
    assert t.calls == []


def test_tune_to_turns_online_tuning_on(tmp_path):
    t = _Tun()
    out = str(tmp_path / 'new.csv')
    ok, mode = gemm_tuning._decide(t, True, env={'MARL_DMFB_GEMM_TUNE_TO': out}, results=str(tmp_path / 'x'))
    assert ok and mode == 'tuning to ' + out
    assert ('tuning_enable', True) in t.calls and ('set_filename', out) in t.calls


def test_shipped_file_exists_and_names_gfx950():
    import os
    assert os.path.exists(gemm_tuning.RESULTS)
    head = open(gemm_tuning.RESULTS).read(2000)
    assert 'Validator' in head and 'gfx950' in head
