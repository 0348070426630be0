#!/bin/bash
# Same-box A/B of the default bench loop under an environment switch:  tools/ab_env.sh VAR [bench flags]  (VAR=0 against VAR=1, twice)
set -e
VAR=$1; shift || true
for i in 1 2; do
  for v in 0 1; do
    printf "%s=%s " $VAR $v
    env $VAR=$v python bench.py --steps 12 --warmup 3 --no_cpu_baseline --no_tiers "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
  done
done
