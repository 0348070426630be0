mkdir -p gpurun_out/m1
timeout -k 10 300 python -m pytest tests/test_vdn_learn_golden.py -q -m gpu -k packed > gpurun_out/m1/pytest.txt 2>&1; tail -5 gpurun_out/m1/pytest.txt
python bench.py --env meda --width 30 --length 30 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 4 --warmup 2 --no_cpu_baseline > gpurun_out/m1/meda_stream.json 2> gpurun_out/m1/err1.txt
python bench.py --env meda --width 30 --length 30 --drop_num 4 --n_envs 4096 --batch_size 256 --train_time 2 --buffer_size 8192 --steps 4 --warmup 2 --no_cpu_baseline --no_stream > gpurun_out/m1/meda_nostream.json 2> gpurun_out/m1/err2.txt
cut -c1-700 gpurun_out/m1/meda_stream.json; cut -c1-300 gpurun_out/m1/meda_nostream.json; tail -3 gpurun_out/m1/err1.txt
