#!/bin/bash
# Capture the measurement records of a round on the MI355X box (run through gpurun from the repo root):
#   tools/capture_profiles.sh r04          (ONLY_BENCH=1 ...: the bench.py sections only; SKIP_PROBES=1: no write-pattern / calibration probes)
# Writes raw rocprofv3 output under gpurun_out/<round>/ (scratch) and the reduced summaries under
# gpurun_out/<round>/summary/ -- copy those into profiles/<round>/ and profiles/traffic.json and commit them.
# Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
set -eo pipefail
ROUND=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$ROUND
SUM=$OUT/summary
mkdir -p "$SUM"
export TMPDIR=/tmp
cd /tmp

echo "== bench.py (plain) ==" ; date
python3 $REPO/bench.py > $SUM/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $SUM/bench.json

echo "== bench.py under rocprofv3 --kernel-trace --stats ==" ; date
# (the persistent k_observe<4> grid is the same at every large batch: this run times ONLY the 655 360-chip roofline launches at it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no_cpu_baseline \
    --roofline_envs_cached 0 --trained_tier_rounds 0 > $SUM/bench_under_rocprof.json 2> $OUT/bench_trace.err
python3 $REPO/tools/reduce_profiles.py trace $OUT/bench_trace $SUM/bench_kernels_by_grid.csv
cp $(ls $OUT/bench_trace/*/*_kernel_stats.csv | head -1) $SUM/bench_kernel_stats.csv

if [ -n "$ONLY_BENCH" ]; then echo "== ONLY_BENCH: the env-kernel sections are skipped =="; exit 0; fi   # the env kernels did not change
echo "== env tiers (plain) ==" ; date
python3 $REPO/tools/bench_env.py --cfg A,D,E,M30,M60,M30v2 --sizes 4096,65536,262144,655360 --msizes 4096,65536,163840 --iters 100 --observe > $SUM/env_tiers.jsonl
cat $SUM/env_tiers.jsonl

echo "== env kernels under rocprofv3 --kernel-trace ==" ; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env_trace -- python3 $REPO/tools/bench_env.py --cfg A,D,E,M30,M60,M30v2 \
    --sizes 4096,262144 --msizes 65536,163840 --iters 40 --observe > $OUT/env_trace.log 2>&1
python3 $REPO/tools/reduce_profiles.py trace $OUT/env_trace $SUM/env_kernels_by_grid.csv '(dmfbk|medak)::|k_meda_observe'
# the roofline batch on its own (same persistent grid as 262 144 chips, so a run of its own): A at 655 360 chips
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/env_trace_big -- python3 $REPO/tools/bench_env.py --cfg A --sizes 655360 --iters 40 --observe \
    > $OUT/env_trace_big.log 2>&1
python3 $REPO/tools/reduce_profiles.py trace $OUT/env_trace_big $SUM/env_A655360_kernels_by_grid.csv 'dmfbk::'

echo "== HBM traffic counters ==" ; date
# sizes 4096 + 262144 (MEDA: 65536) only: the fused 4096-chip launch and the 65536-chip step-only launch share a grid size,
# and so do the persistent MEDA observation launches of all batch sizes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/bench_env.py --cfg A,D,E,M30,M60,M30v2 \
    --sizes 4096,262144 --msizes 65536,163840 --iters 24 --observe --labels $OUT/labels.json > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/bench_env.py --cfg A,D,E,M30,M60,M30v2 \
    --sizes 4096,262144 --msizes 65536,163840 --iters 24 --observe > $OUT/pmc_write.log 2>&1
python3 $REPO/tools/reduce_profiles.py traffic $OUT/pmc_fetch $OUT/pmc_write $SUM/traffic.json $OUT/labels.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_big -- python3 $REPO/tools/bench_env.py --cfg A --sizes 655360 --iters 24 --observe \
    --labels $OUT/labels_big.json > $OUT/pmc_fetch_big.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_big -- python3 $REPO/tools/bench_env.py --cfg A --sizes 655360 --iters 24 --observe \
    > $OUT/pmc_write_big.log 2>&1
python3 $REPO/tools/reduce_profiles.py traffic $OUT/pmc_fetch_big $OUT/pmc_write_big $OUT/traffic_big.json $OUT/labels_big.json
python3 - <<PY
import json
a = json.load(open('$SUM/traffic.json')); b = json.load(open('$OUT/traffic_big.json'))
a['detail'] += b.pop('detail'); b.pop('note', None); a.update(b)
json.dump(a, open('$SUM/traffic.json', 'w'), indent=1)
PY
if [ -n "$SKIP_PROBES" ]; then echo "== SKIP_PROBES: the write-pattern and calibration probes are not re-run =="; exit 0; fi
echo "== write-pattern and counter-calibration probes ==" ; date
make -C $REPO/tools/probe -s bin/write_probe bin/fetch_calib || true   # built from source, never a checked-in binary
if [ -x $REPO/tools/probe/bin/write_probe ]; then
  $REPO/tools/probe/bin/write_probe 642 1840 4 > $SUM/write_probe_642MB.txt 2>&1 || true
  $REPO/tools/probe/bin/write_probe 257 1840 4 > $SUM/write_probe_257MB.txt 2>&1 || true
fi
if [ -x $REPO/tools/probe/bin/fetch_calib ]; then
  rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv \
      -d $OUT/calib_req -- $REPO/tools/probe/bin/fetch_calib > $OUT/calib_req.log 2>&1 || true
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- $REPO/tools/probe/bin/fetch_calib > $OUT/calib_fetch.log 2>&1 || true
  python3 - <<PY > $SUM/fetch_calibration.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for d in ('calib_req', 'calib_fetch'):
    for f in glob.glob('$OUT/%s/**/*_counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Kernel_Name'].startswith('k_'):
                acc[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
print('k_stream reads 1 GiB per launch (16 B per lane, coalesced); k_gather8 makes 16 Mi isolated 8-byte gathers per launch')
for k, v in sorted(acc.items()):
    print('%-10s %-26s %.1f' % (k[0], k[1], sum(v) / len(v)))
PY
  cat $SUM/fetch_calibration.txt
fi
echo "== done ==" ; date
