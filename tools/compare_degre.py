"""Print a degradation sweep of this build (rewards/steps/success/health.npy written by tools/train_degre.py or
marl_dmfb_amd/evaDegre.py) beside the reference-held curves of the same chip shape (container-only: reads
/root/reference/DegreData/<shape>/*.npy, the saved output of the reference's evaDegre.py:29-56).

    python tools/compare_degre.py profiles/r04/degre/DegreData_5chips/20by20-10d0b [20by20-10d0b]

Per epoch: success, steps, reward (mean over chips; the reference's own min..max over its 5 chips in brackets) and the mean
electrode health at the START of the epoch (evaDegre.py:21); then the epoch of collapse (first epoch with mean success < 0.1)
and the final mean health.  The reference's evaluate_task is not stored with its data; it is inferred from the granularity of
its success values (20by20-10d0b: multiples of 1/20)."""
import os
import sys
from fractions import Fraction

import numpy as np


def load(d):
    return {k: np.load(os.path.join(d, k + '.npy')) for k in ('success', 'steps', 'rewards', 'health')}


def infer_tasks(success):
    den = 1
    for v in np.unique(np.round(success, 6)):
        den = np.lcm(den, Fraction(float(v)).limit_denominator(200).denominator)
    return int(den)


def collapse_epoch(success_mean):
    below = np.nonzero(success_mean < 0.1)[0]
    return int(below[0]) if len(below) else None


def main():
    ours_dir = sys.argv[1]
    shape = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.normpath(ours_dir))
    ref_dir = os.path.join('/root/reference/DegreData', shape)
    a, b = load(ours_dir), load(ref_dir)
    n_ep = min(a['success'].shape[1], b['success'].shape[1])
    print('ours: %s  chips=%d epochs=%d   reference: %s  chips=%d epochs=%d  (reference evaluate_task inferred: %d)' % (
        ours_dir, a['success'].shape[0], a['success'].shape[1], ref_dir, b['success'].shape[0], b['success'].shape[1],
        infer_tasks(b['success'])))
    print('%5s | %-31s | %-29s | %-31s | %s' % ('epoch', 'success  ours  ref [min..max]', 'steps  ours  ref [min..max]',
                                              'reward  ours  ref [min..max]', 'mean health  ours  ref'))
    for e in range(n_ep):
        row = '%5d |' % e
        for k, f in (('success', '%5.2f'), ('steps', '%5.1f'), ('rewards', '%6.2f')):
            row += (' ' + f + '  ' + f + ' [' + f + '..' + f + '] |') % (a[k][:, e].mean(), b[k][:, e].mean(), b[k][:, e].min(), b[k][:, e].max())
        row += ' %.3f  %.3f' % (a['health'][:, e].mean(), b['health'][:, e].mean())
        print(row)
    sa, sb = a['success'].mean(0), b['success'].mean(0)
    print('epoch-0 success: ours %.3f, reference %.3f (its chips %.2f..%.2f)' % (sa[0], sb[0], b['success'][:, 0].min(), b['success'][:, 0].max()))
    print('epoch of collapse (mean success < 0.1): ours %s, reference %s' % (collapse_epoch(sa), collapse_epoch(sb)))
    print('final mean health: ours %.3f, reference %.3f' % (a['health'][:, n_ep - 1].mean(), b['health'][:, n_ep - 1].mean()))


if __name__ == '__main__':
    main()
