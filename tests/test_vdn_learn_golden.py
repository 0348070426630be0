"""VDN.learn parity (floating point, fp32): two consecutive learns on a fixed minibatch with
deterministic weights must reproduce the reference's clipped gradients, gradient norms and
updated weights (policy/vdn.py:79-132 run in this container, see tools/oracle/gen_vdn_golden.py).

Tolerance (1e-5 relative), the same on CPU and GPU: |g - ref| <= 1e-5 x max|ref| + 1e-5 x |ref| per
sampled gradient element, 1e-5 relative on the gradient norm.  The build batches the convolutions over all T steps, runs
the target net under no_grad, and on the GPU goes through the HIP front end, split-K weight gradients and the one-launch
GRU kernels, so summation order differs from the reference's per-step loop; everything else is the same arithmetic.
Measured (tools/diag_learn_tol.py): max|g - ref| / max|ref| <= 1.8e-6 on the CPU and <= 1.4e-6 on MI355X for every
parameter tensor, with the HIP GRU or the per-step aten cell, with the HIP conv front end or the im2col GEMM path."""
import glob
import os

import numpy as np
import pytest

from vdn_helpers import learn_golden_check

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'vdn_learn_*.npz')))


@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_matches_reference_cpu(path):
    learn_golden_check(path, 'cpu', rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('replay_dtypes', [True, False], ids=['fused_td', 'tensor_op_td'])
@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_matches_reference_gpu(path, replay_dtypes):
    """fused_td: the batch in the replay buffer's dtypes -> k_td_forward/backward + time-major Q values, i.e. what Trainer.run
    and bench.py execute; tensor_op_td: the golden's float64 rewards -> the torch TD block.  Both against the reference's numbers."""
    learn_golden_check(path, 'cuda:0', rtol=1e-5, atol=1e-5, replay_dtypes=replay_dtypes)


@pytest.mark.gpu
@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_with_host_length_bound_matches_reference_gpu(path):
    """The shipped default of Trainer.collect_and_learn (args.host_len_bound): the learn is handed max_len = the slots of the
    batch instead of reading the batch's own length back, through the fused TD block -- the path bench.py times."""
    learn_golden_check(path, 'cuda:0', rtol=1e-5, atol=1e-5, replay_dtypes=True, host_len_bound=True)


@pytest.mark.gpu
@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_packed_matches_reference_gpu(path):
    """VDN.learn_packed (what Trainer runs in continuous mode): conv front end, GRU input projection and head on the valid
    (episode, step) rows only, GRU sequence kernels that stop every row at its own length, TD block indexing the replay tensors
    in place -- against the reference's numbers for batches whose episodes have 3 .. 80 valid steps."""
    learn_golden_check(path, 'cuda:0', rtol=1e-5, atol=1e-5, replay_dtypes=True, packed=True,
                       atol_step1=5e-5 if 'meda' in os.path.basename(path) else None)


@pytest.mark.gpu
@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_packed_valu_recurrence_matches_reference_gpu(path, monkeypatch):
    """The same with the two networks' recurrences through the 8-row VALU kernels (MARL_DMFB_GRU_PAIR=0) instead of the one
    matrix-core launch: every golden at 1e-5 on both learns.  Why the MEDA golden's SECOND learn gets 5e-5 above: that golden (20
    rows, 5 to 16 steps) is 40 x more sensitive to rounding than the DMFB ones -- one-ulp noise on weight_hh moves its second-learn
    gradients by 1.5e-6 of the tensor scale against 3.8e-8 for vdn_learn_4d_od24_b64 (first learn: 1.2e-6 / 2.6e-8) -- and the two
    recurrence kernels sum the 128 products of a gate in different orders (both within 5e-7 of a float64 GRU, rms 5e-8,
    tools/dbg_gru_err.py); the first learn and all weights agree at 1e-5 / 2e-6 on either kernel."""
    monkeypatch.setenv('MARL_DMFB_GRU_PAIR', '0')
    learn_golden_check(path, 'cuda:0', rtol=1e-5, atol=1e-5, replay_dtypes=True, packed=True)


@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_learn_with_host_length_bound_matches_reference_cpu(path):
    learn_golden_check(path, 'cpu', rtol=1e-5, atol=1e-5, host_len_bound=True)


@pytest.mark.parametrize('path', FILES, ids=os.path.basename)
def test_goldens_hold_padded_and_early_terminated_episodes(path):
    """Every learn golden pins mask = 1 - padded, (1 - terminated) and the length trim (policy/vdn.py:115-122,
    agent/agent.py:51-61) to the reference: at least a third of its episodes carry padded steps, lengths differ, and an
    episode that ends early has terminated = 1 on its last valid step."""
    g = np.load(path)
    padded, term = g['padded'][:, :, 0].astype(bool), g['terminated'][:, :, 0].astype(bool)
    lens = (~padded).sum(1)
    T = padded.shape[1]
    assert (lens < T).sum() * 3 >= len(lens), lens
    assert len(set(lens.tolist())) >= 3, lens
    for b, ln in enumerate(lens):
        assert not padded[b, :ln].any() and padded[b, ln:].all()
        assert term[b, ln - 1] and not term[b, :ln - 1].any() and term[b, ln:].all()
    if os.path.basename(path) == 'vdn_learn_4d_od24_short.npz':
        assert lens.max() < T          # the batch's own length is below the limit: max_len = T adds all-padded steps
