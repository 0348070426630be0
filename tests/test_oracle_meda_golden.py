"""Pins the MEDA CPU oracle (oracle/meda_oracle.c) to the reference: every committed golden
episode captured from the real MEDAEnv must replay bit-exactly; plus the reference's ctor guard."""
import os

import pytest

from oracle.meda_oracle import MedaOracle
from meda_replay import golden_files, replay


@pytest.mark.parametrize('path', golden_files(), ids=os.path.basename)
def test_oracle_replays_reference_golden(path):
    assert replay(path, lambda **kw: MedaOracle(**kw)) > 0


def test_meda_10x10_is_rejected_like_the_reference():
    # BASELINE.json config 3 "MEDA 10x10, drop_num=4": the reference raises RuntimeError (meda.py:151-154)
    with pytest.raises(RuntimeError):
        MedaOracle(10, 10, 4)
    MedaOracle(30, 30, 4)
    with pytest.raises(RuntimeError):
        MedaOracle(30, 30, 5)
