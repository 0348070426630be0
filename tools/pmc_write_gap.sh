#!/bin/bash
# TCC (L2 <-> fabric) counters of the write-only probe kernels and of k_observe<4> at 655 360 chips (642 MB of output), one
# rocprofv3 --pmc pass per counter set (kernel trace only: gpurun refuses --pmc together with other trace domains).
#   tools/pmc_write_gap.sh <out dir under gpurun_out>     then: python tools/reduce_profiles.py pmc <dir> <summary.json>
set -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-pmc_write}
mkdir -p $OUT
make -C $REPO/tools/probe -s bin/write_probe_pmc
export TMPDIR=/tmp
cd /tmp
SETS=(
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"
 "TCC_EA0_WRREQ_LEVEL_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum"
 "TCC_REQ_sum TCC_WRITE_REQ_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum"
 "WRITE_SIZE"
 "FETCH_SIZE"
)
k=0
for s in "${SETS[@]}"; do
  echo "== set $k: $s" ; date
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/probe_$k -- $REPO/tools/probe/bin/write_probe_pmc 642 > $OUT/probe_$k.log 2>&1 || echo "probe set $k failed"
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/obs_$k -- python3 $REPO/tools/bench_env.py --cfg A --sizes 655360 --iters 12 --observe > $OUT/obs_$k.log 2>&1 || echo "observe set $k failed"
  k=$((k+1))
done
echo done; date
