#!/bin/bash
# Headline loop (DMFB 10x10, 4 droplets, fov 9) at different chip counts per GPU; learns scale with the batch (4 learns x n_envs/8 episodes):
#   tools/sweep_n_envs.sh  -> gpurun_out/r04/summary/sweep_n_envs.jsonl
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04/summary; O=gpurun_out/r04/summary/sweep_n_envs.jsonl; : > $O
for E in 512 1024 2048 4096 8192 16384 32768; do
  python bench.py --n_envs $E --batch_size $((E / 8)) --train_time 4 --buffer_size $((E * 4)) --steps 8 --warmup 3 --no_cpu_baseline --no_tiers 2>/dev/null | tail -1 >> $O
done
python - <<'P'
import json
for l in open('gpurun_out/r04/summary/sweep_n_envs.jsonl'):
    j = json.loads(l)
    print(j['config']['workload'][:60], '| %.3g env-steps/s | %.2f ms per round' % (j['value'], j['ms_per_step']))
P
