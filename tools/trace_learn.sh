set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/trace_learn; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_tiers > $OUT/bench.json 2> $OUT/err.txt
f=$(ls $OUT/trace/*/*_kernel_trace.csv | head -1)
python3 - "$f" "$OUT/tail.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-1400:]
with open(sys.argv[2], 'w') as f:
    w = csv.writer(f)
    t0 = int(rows[0]['Start_Timestamp'])
    for r in rows:
        w.writerow([r['Kernel_Name'][:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - int(r['Start_Timestamp'])])
P
rm -rf $OUT/trace
