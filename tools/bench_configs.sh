set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/summary; O=gpurun_out/r04/summary/bench_configs.jsonl; : > $O
# BASELINE configs[2]: MEDA (10x10 is rejected by the reference, meda.py:151-154): smallest legal 4-droplet chip and the CLI default
python bench.py --env meda --width 30 --length 30 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 4 --warmup 1 --no_cpu_baseline 2>/dev/null >> $O
python bench.py --env meda --width 30 --length 60 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 4 --warmup 1 --no_cpu_baseline 2>/dev/null >> $O
# configs[3]: DMFB 50x50, 10 droplets, one GPU's share (1024 of 8192 chips)
python bench.py --width 50 --length 50 --drop_num 10 --n_envs 1024 --batch_size 128 --train_time 2 --buffer_size 2048 --steps 4 --warmup 1 --no_cpu_baseline 2>/dev/null >> $O
# configs[4]: DMFB 20x20, 10 droplets, degradation, 4096 chips: the evaDegre path (greedy evaluation episodes, chips keep ageing) and training
python bench.py --width 20 --length 20 --drop_num 10 --degrade --eval_only --n_envs 4096 --steps 6 --warmup 2 2>/dev/null >> $O
python bench.py --width 20 --length 20 --drop_num 10 --degrade --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 4 --warmup 1 --no_cpu_baseline 2>/dev/null >> $O
cut -c1-420 $O
