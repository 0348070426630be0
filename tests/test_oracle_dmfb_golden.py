"""Pins the CPU oracle (oracle/dmfb_oracle.c) to the reference: every committed golden
episode captured from the real DMFBenv must replay bit-exactly."""
import os

import numpy as np
import pytest

from oracle.dmfb_oracle import DmfbOracle, philox
from dmfb_replay import golden_files, replay


def make_oracle(**kw):
    return DmfbOracle(**kw)


@pytest.mark.parametrize('path', golden_files(), ids=os.path.basename)
def test_oracle_replays_reference_golden(path):
    steps = replay(path, make_oracle)
    assert steps > 0


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(v) for v in philox(0, 0, [0, 0, 0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in philox(0xffffffff, 0xffffffff, [0xffffffff] * 4)] == [
        '0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in philox(0xa4093822, 0x299f31d0, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344])] == [
        '0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']
