"""profiles/<round>/traffic.json = the capture's traffic.json with the MEDA observation rows of the mixed-size pass (whose persistent
grid is the same at every batch size, so its launches cannot be told apart) replaced by the single-size pass
traffic_meda_E163840.json (tools/capture_extra_r04.sh); also written to profiles/traffic.json, which bench.py reads.
    python tools/merge_traffic.py r04"""
import json
import os
import sys

rnd = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cap = json.load(open(os.path.join(root, 'gpurun_out', rnd, 'summary', 'traffic.json')))
meda = json.load(open(os.path.join(root, 'profiles', rnd, 'traffic_meda_E163840.json')))
old_path = os.path.join(root, 'profiles', rnd, 'traffic.json')
old = json.load(open(old_path)) if os.path.exists(old_path) else {}
drop = [d['key'] for d in cap['detail'] if 'meda_observe' in d['key'] and '+' in d['key'] and 'E163840' in d['key']]
cap['detail'] = [d for d in cap['detail'] if d['key'] not in drop]
for k in list(cap):
    if 'meda_observe' in k and any(k in x.split('+') for x in drop):
        del cap[k]
for k, v in meda.items():
    if k not in ('detail', 'note'):
        cap[k] = v
cap['detail'] += meda['detail']
if 'note_r04' in old:
    cap['note_r04'] = old['note_r04']
for path in (old_path, os.path.join(root, 'profiles', 'traffic.json')):
    json.dump(cap, open(path, 'w'), indent=1)
print('dropped', drop)
