#!/bin/bash
# Kernel-level A/B under an environment switch: average duration of the kernels matching PATTERN in the default bench loop
#   tools/ab_kernel.sh VAR PATTERN
set -e
VAR=$1; PAT=$2
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  rm -rf /tmp/abk$v
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk$v -- python3 $REPO/bench.py --steps 6 --warmup 2 --no_cpu_baseline --no_tiers > /dev/null 2>&1
  echo "$VAR=$v"; python3 -c "
import csv, glob, re, sys
for r in csv.DictReader(open(glob.glob('/tmp/abk$v/*/*_kernel_stats.csv')[0])):
    if re.search(sys.argv[1], r['Name']): print('%-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
" "$PAT"
done
