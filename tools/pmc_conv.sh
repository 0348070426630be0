#!/bin/bash
# SQ counters of the Q-network front-end kernels (k_conv9_mfma<24> forward, k_conv9_bwd<24> backward) over tools/bench_conv.py, one
# rocprofv3 --pmc pass per counter set (kernel trace only).   tools/pmc_conv.sh <dir under gpurun_out> ; then
#   python tools/reduce_profiles.py pmc gpurun_out/<dir> profiles/r03/pmc_conv_summary.json 'k_conv9'
set -o pipefail
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-pmc_conv}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SETS=(
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
k=0
for s in "${SETS[@]}"; do
  echo "== set $k: $s"; date
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $OUT/conv_$k -- python3 $REPO/tools/bench_conv.py > $OUT/conv_$k.log 2>&1 || echo "set $k failed"
  k=$((k+1))
done
echo done; date
