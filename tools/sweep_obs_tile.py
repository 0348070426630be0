"""Sweep of k_observe's launch shape (chips per tile x persistent workgroups per CU) at a batch whose output is beyond the
256 MiB Infinity Cache: `python tools/sweep_obs_tile.py A 655360`.  Uses the library's tuning knobs DMFB_VEC_OBS_TILE /
DMFB_VEC_OBS_PER_CU; times the kernel by the dispatch time stamps of its launches inside the env-only lock-step loop."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_env  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'A'
E = int(sys.argv[2]) if len(sys.argv) > 2 else 655360
tiles = [int(t) for t in (sys.argv[3].split(',') if len(sys.argv) > 3 else '4,8,16,32'.split(','))]
percu = [int(t) for t in (sys.argv[4].split(',') if len(sys.argv) > 4 else '0,1,2,4'.split(','))]
oneshot = os.environ.get('SWEEP_ONESHOT', '0')
for t in tiles:
    for pc in percu:
        os.environ['DMFB_VEC_OBS_TILE'] = str(t)
        os.environ['DMFB_VEC_OBS_PER_CU'] = str(pc)
        os.environ['DMFB_VEC_OBS_ONESHOT'] = oneshot
        r = bench_env.run(name, E, 30, observe=True)
        o = r['observe']
        print(json.dumps({'cfg': name, 'E': E, 'tile': t, 'per_cu': pc, 'oneshot': int(oneshot), 'us': o['us_per_launch'], 'b2b_us': o['back_to_back_us_per_launch'],
                          'TBps': round(o['algo_GBps'] / 1e3, 2), 'frac': o['frac_of_8TBps'], 'lockstep_us': r['us_per_launch']}), flush=True)
