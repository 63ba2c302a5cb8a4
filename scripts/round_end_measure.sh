set -x
timeout 900 python3 scripts/profile_round.py fp32 > gpurun_out/profile_round_fp32.log 2>&1
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_cmd.json 2> gpurun_out/driver.err
timeout 400 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/default.err
timeout 400 python3 bench.py --workload C3 --precision bf16x3 > gpurun_out/r03_bench_c3_bf16x3.json 2> gpurun_out/c3x3.err
timeout 400 python3 bench.py --workload C3 --precision bf16 > gpurun_out/r03_bench_c3_bf16.json 2> gpurun_out/c3.err
timeout 900 python3 bench.py --workload C5 --precision bf16x3 --steps 2000 --warmup 2 > gpurun_out/r03_bench_c5_bf16x3.json 2> gpurun_out/c5.err
timeout 900 python3 bench.py --workload C5 --precision bf16 --steps 2000 --warmup 2 > gpurun_out/r03_bench_c5_bf16.json 2> gpurun_out/c5b.err
for P in 0 2 1; do PREC=$P timeout 200 python3 scripts/bench_bank_kernels.py 56000 250000 1200000 2>&1 | grep "B=" > gpurun_out/r03_bank_kernels_prec$P.txt; done
tail -c 300 gpurun_out/*.err
timeout 300 python3 scripts/profile_layers.py > gpurun_out/r03_layers.txt 2>&1
hipcc -O3 --offload-arch=gfx950 scripts/clock_probe_bf16.hip -o /tmp/cp16 2>/dev/null && /tmp/cp16 > gpurun_out/r03_clock_probe_bf16.json
timeout 300 python3 scripts/bench_train_step.py > gpurun_out/r03_train_step.txt 2>&1
