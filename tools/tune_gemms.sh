#!/bin/bash
# Regenerate marl_dmfb_amd/tuning/gemm_gfx950.csv on an MI355X box (through gpurun from the repo root):
#   tools/tune_gemms.sh          -> gpurun_out/gemm_gfx950.csv ; copy it to marl_dmfb_amd/tuning/ and commit
# PyTorch TunableOp measures every rocBLAS / hipBLASLt solution for each GEMM shape the BASELINE configurations issue
# (rollout and learn) and keeps the fastest; later runs load the file with on-line tuning OFF (common/gemm_tuning.py).
set -eo pipefail
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
OUT=$REPO/gpurun_out/gemm_gfx950.csv
rm -f $OUT
export MARL_DMFB_GEMM_TUNE_TO=$OUT
cd /tmp
run() { python3 $REPO/bench.py --no_cpu_baseline --no_tiers "$@" 2>/dev/null | cut -c1-160; wc -l $OUT; }
run --steps 3 --warmup 2                                                                  # configs[1]: DMFB 10x10, 4 droplets, 4096 chips
run --env meda --width 30 --length 30 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 2 --warmup 1
run --env meda --width 30 --length 60 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 2 --warmup 1
run --width 50 --length 50 --drop_num 10 --n_envs 1024 --batch_size 128 --train_time 2 --buffer_size 2048 --steps 2 --warmup 1
run --width 20 --length 20 --drop_num 10 --degrade --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 2 --warmup 1
cat $OUT | cut -c1-140
