#!/bin/bash
# Second half of the round-4 capture: SQ counters of the conv kernels, MEDA traffic at the beyond-the-cache batch (a run of its own: the
# persistent observation grid is the same at every batch size, so one process = one batch size), bench lines of the other configs.
set -o pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/r04; SUM=$OUT/summary; mkdir -p $SUM
export TMPDIR=/tmp
tools/pmc_conv.sh r04/pmc_conv > $OUT/pmc_conv.log 2>&1
python3 tools/reduce_profiles.py pmc $OUT/pmc_conv $SUM/pmc_conv_summary.json 'k_conv9' > /dev/null
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_m -- python3 $REPO/tools/bench_env.py --cfg M30,M30v2 --sizes 163840 --msizes 163840 --iters 24 --observe \
    --labels $OUT/labels_m.json > $OUT/pmc_fetch_m.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_m -- python3 $REPO/tools/bench_env.py --cfg M30,M30v2 --sizes 163840 --msizes 163840 --iters 24 --observe \
    > $OUT/pmc_write_m.log 2>&1
python3 $REPO/tools/reduce_profiles.py traffic $OUT/pmc_fetch_m $OUT/pmc_write_m $SUM/traffic_meda_E163840.json $OUT/labels_m.json
cd $REPO
GRAFT_REPO_ROOT=$REPO bash tools/bench_configs.sh > $OUT/bench_configs.log 2>&1
echo done
