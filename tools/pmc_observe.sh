#!/bin/bash
# SQ counter passes over the observation kernel (tools/exp_observe.py)
set -eo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_obs; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CFG=${1:-D}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $REPO/tools/exp_observe.py 262144 $CFG > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $REPO/tools/exp_observe.py 262144 $CFG > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p3 -- python3 $REPO/tools/exp_observe.py 262144 $CFG > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ('p1','p2','p3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$OUT/%s/**/*_counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_observe' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:30] + '|' + r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
