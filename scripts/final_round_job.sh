cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/r04_gpu_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.json 2> gpurun_out/r04_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd_2.json 2>> gpurun_out/r04_bench.err
VFN_WINOGRAD=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_driver_cmd_winograd_off.json 2>> gpurun_out/r04_bench.err
python bench.py > gpurun_out/r04_bench_default.json 2>> gpurun_out/r04_bench.err
VFN_DIST_BACKEND=gloo VFN_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_2ranks_one_device_gloo.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C3 --precision bf16x3 --no-cpu-baseline > gpurun_out/r04_bench_c3_bf16x3.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C3 --precision bf16 --no-cpu-baseline > gpurun_out/r04_bench_c3_bf16.json 2>> gpurun_out/r04_bench.err
python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/r04_train_step.txt 2>&1
python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/r04_train_step.txt 2>&1
bash scripts/profile_train.sh r04
python scripts/profile_round.py fp32 > gpurun_out/r04_profile_round.log 2>&1
python scripts/profile_layers.py > gpurun_out/r04_layers.txt 2>&1
