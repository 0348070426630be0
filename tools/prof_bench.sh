#!/bin/bash
# rocprofv3 kernel trace of the default bench loop (no tiers, no CPU baseline), reduced per (kernel, grid):
#   tools/prof_bench.sh <tag> [extra bench.py flags]     -> gpurun_out/<tag>/{bench.json,kernels_by_grid.csv,kernel_stats.csv}
set -eo pipefail
TAG=${1:-prof}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_tiers "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 $REPO/tools/reduce_profiles.py trace $OUT/trace $OUT/kernels_by_grid.csv
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
