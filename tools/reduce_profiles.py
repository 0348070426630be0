"""Reduce rocprofv3 CSV output (gpurun_out/ scratch) to the small summaries kept under profiles/.

  kernel trace  -> <out>/<tag>_kernels_by_grid.csv : per (kernel, grid size): calls, AverageNs, MinNs, MaxNs, registers, LDS.
                   rocprofv3's own --stats table aggregates by name only; the env kernels run at several batch sizes in one
                   process (k_observe<4> at 4 096 chips inside the loop and at 262 144 as the roofline kernel), so the judged
                   averages are the per-grid rows.
  --pmc passes  -> profiles/traffic.json : HBM bytes per launch per (kernel, grid), FETCH_SIZE doubled (gfx950 note in
                   MI355X_MICROARCH.md, HBM section), WRITE_SIZE as read; both counters are reported in KB by rocprofv3.

usage: reduce_profiles.py trace <dir with *_kernel_trace.csv> <out csv> [name filter regex]
       reduce_profiles.py traffic <fetch dir> <write dir> <traffic.json> <labels.json>
       reduce_profiles.py pmc <dir with one sub-directory per --pmc pass> <out json> [name filter regex]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'((?:dmfbk|medak|crnn_mfma|\(anonymous namespace\))::k_\w+(?:<[^>]*>)?|Cijk_\w{0,40}|k_\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:60]


def trace_rows(d):
    for f in sorted(glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True)):
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                yield r


def reduce_trace(d, out, flt=None):
    agg = defaultdict(list)
    meta = {}
    for r in trace_rows(d):
        name = short(r['Kernel_Name'])
        if flt and not re.search(flt, name):
            continue
        grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        wg = int(r['Workgroup_Size_X'])
        key = (name, grid, wg)
        agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[key] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'])
    total = sum(sum(v) for v in agg.values()) or 1
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Kernel', 'Grid_Size', 'Workgroup_Size', 'Workgroups', 'Calls', 'TotalDurationNs', 'AverageNs', 'MedianNs', 'MinNs', 'MaxNs',
                    'Percentage', 'VGPR', 'AGPR', 'SGPR', 'LDS_Block_Size'])
        for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            v2 = sorted(v)
            w.writerow([key[0], key[1], key[2], key[1] // max(1, key[2]), len(v), sum(v), round(sum(v) / len(v), 1), v2[len(v2) // 2], v2[0], v2[-1],
                        round(100.0 * sum(v) / total, 2)] + list(meta[key]))
    return out


def pmc_rows(d):
    for f in sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)):
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                yield r


def reduce_traffic(fetch_dir, write_dir, out_json, labels_json):
    """labels.json (written by tools/bench_env.py --labels): {"<short kernel name>|<grid size>": [{"key":
    "k_observe_10x10_4d_E262144", "algo_bytes": N}, ...]}; several keys on one row = configurations that make the very
    same launch (e.g. D and E share k_observe<10>)."""
    labels = json.load(open(labels_json))
    acc = defaultdict(lambda: defaultdict(list))
    for d, counter in ((fetch_dir, 'FETCH_SIZE'), (write_dir, 'WRITE_SIZE')):
        for r in pmc_rows(d):
            if r['Counter_Name'] != counter:
                continue
            acc['%s|%s' % (short(r['Kernel_Name']), r['Grid_Size'])][counter].append(float(r['Counter_Value']))
    res = {}
    detail = []
    for k, labs in labels.items():
        if k not in acc or not acc[k]['FETCH_SIZE'] or not acc[k]['WRITE_SIZE']:
            continue
        lab = labs[0]
        f = acc[k]['FETCH_SIZE']
        w = acc[k]['WRITE_SIZE']
        # drop the first dispatches (cold caches / first-touch page faults)
        f, w = f[len(f) // 4:], w[len(w) // 4:]
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        hbm = int(round((2.0 * fk + wk) * 1024))
        for l2 in labs:
            res[l2['key']] = hbm
        detail.append({'key': '+'.join(l2['key'] for l2 in labs), 'kernel': k.split('|')[0], 'grid_size': int(k.split('|')[1]), 'n_dispatch': len(f),
                       'FETCH_SIZE_KB_avg': round(fk, 3), 'WRITE_SIZE_KB_avg': round(wk, 3), 'hbm_bytes_per_launch': hbm,
                       'algo_bytes_per_launch': lab['algo_bytes'], 'ratio': round(hbm / lab['algo_bytes'], 3)})
    res['detail'] = detail
    res['note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/bench_env.py (env-only tier, '
                   'shipped kernels); both counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 tallies '
                   '128-B requests at 64 B); hbm_bytes_per_launch = (2*FETCH + WRITE)*1024 averaged over the last 3/4 of the dispatches. '
                   'Calibrated for the two read patterns of these kernels with tools/probe/fetch_calib.hip (TCC_EA0_RDREQ_128B/_64B/_32B): '
                   'a 16-B-per-lane stream and isolated 8-byte gathers BOTH issue 128-byte requests only, tallied at 64 B, so the '
                   'doubling holds for the health gathers of k_step<N,true> too (one 128-byte line per gathered float64)')
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)
    return res


def reduce_pmc(root, out_json, flt=None):
    """Every *_counter_collection.csv below `root` (one sub-directory per --pmc pass) -> {kernel|grid: {counter: average per
    dispatch, 'dispatches': n, 'avg_ns': kernel time from the same pass}} ; the first quarter of the dispatches is dropped."""
    acc = defaultdict(lambda: defaultdict(list))
    for r in pmc_rows(root):
        name = short(r['Kernel_Name'])
        if flt and not re.search(flt, name):
            continue
        key = '%s|%s' % (name, r['Grid_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        if 'Start_Timestamp' in r and r.get('End_Timestamp'):
            acc[key]['_ns_' + r['Counter_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    res = {}
    for key, cs in acc.items():
        row = {}
        for c, v in cs.items():
            v = v[len(v) // 4:]
            if c.startswith('_ns_'):
                row.setdefault('avg_ns_by_pass', {})[c[4:]] = round(sum(v) / len(v), 1)
            else:
                row[c] = round(sum(v) / len(v), 2)
                row['dispatches'] = len(v)
        res[key] = row
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    return res


if __name__ == '__main__':
    if sys.argv[1] == 'pmc':
        r = reduce_pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
        print(json.dumps(sorted(r)))
    elif sys.argv[1] == 'trace':
        print(reduce_trace(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None))
    elif sys.argv[1] == 'traffic':
        r = reduce_traffic(*sys.argv[2:6])
        print(json.dumps([(d['key'], d['ratio']) for d in r['detail']]))
    else:
        sys.exit(__doc__)
