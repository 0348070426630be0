"""VDN.learn parity (floating point, fp32): two consecutive learns on a fixed minibatch with
deterministic weights must reproduce the reference's clipped gradients, gradient norms and
updated weights (policy/vdn.py:79-132 run in this container, see tools/oracle/gen_vdn_golden.py).

Tolerance: rtol 2e-4 on values, atol 2e-5 x max|grad| per tensor.  The build batches the
convolutions over all T steps and runs the target net under no_grad, so summation order differs
from the reference's per-step loop; everything else is the same arithmetic."""
import glob
import os

import pytest

from vdn_helpers import learn_golden_check

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'vdn_learn_*.npz')))


@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_matches_reference_cpu(path):
    learn_golden_check(path, 'cpu', rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_matches_reference_gpu(path):
    learn_golden_check(path, 'cuda:0', rtol=2e-3, atol=2e-4)
