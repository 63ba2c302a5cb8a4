cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_gpu_suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.json 2> gpurun_out/r04_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd_2.json 2>> gpurun_out/r04_bench.err
python bench.py > gpurun_out/r04_bench_default.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C3 --precision bf16x3 > gpurun_out/r04_bench_c3_bf16x3.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C3 --precision bf16 > gpurun_out/r04_bench_c3_bf16.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C5 --precision bf16x3 --steps 2000 --warmup 2 > gpurun_out/r04_bench_c5_bf16x3.json 2>> gpurun_out/r04_bench.err
python bench.py --workload C5 --precision bf16 --steps 2000 --warmup 2 > gpurun_out/r04_bench_c5_bf16.json 2>> gpurun_out/r04_bench.err
VFN_DIST_BACKEND=gloo VFN_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_2ranks_one_device_gloo.json 2>> gpurun_out/r04_bench.err
python scripts/bench_train_step.py > gpurun_out/r04_train_step.txt 2>&1
