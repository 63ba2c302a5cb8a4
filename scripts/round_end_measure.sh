# Round-5 evidence set (run on the GPU box through gpurun; everything lands in gpurun_out/, copy into profiles/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd.json 2> gpurun_out/driver.err
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_2.json 2>> gpurun_out/driver.err
VFN_WINOGRAD=0 timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_driver_cmd_winograd_off.json 2>> gpurun_out/driver.err
timeout 400 python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/default.err
timeout 400 python3 bench.py --workload C3 --precision bf16x3 > gpurun_out/r05_bench_c3_bf16x3.json 2> gpurun_out/c3x3.err
timeout 400 python3 bench.py --workload C3 --precision bf16 > gpurun_out/r05_bench_c3_bf16.json 2> gpurun_out/c3.err
timeout 400 python3 bench.py --workload C3 --precision fp32 > gpurun_out/r05_bench_c3_fp32.json 2> gpurun_out/c3f.err
timeout 900 python3 bench.py --workload C5 --precision bf16x3 --steps 2000 --warmup 2 > gpurun_out/r05_bench_c5_bf16x3.json 2> gpurun_out/c5.err
timeout 900 python3 bench.py --workload C5 --precision bf16 --steps 2000 --warmup 2 > gpurun_out/r05_bench_c5_bf16.json 2> gpurun_out/c5b.err
timeout 1500 python3 scripts/bf16_trained_margins.py > gpurun_out/r05_margins.log 2>&1
for P in bf16 bf16x3; do
timeout 400 python3 bench.py --workload C3 --precision $P --checkpoint /tmp/vfn_trained.pth > gpurun_out/r05_bench_c3_${P}_trained.json 2>> gpurun_out/trained.err
timeout 900 python3 bench.py --workload C5 --precision $P --steps 2000 --warmup 2 --checkpoint /tmp/vfn_trained.pth > gpurun_out/r05_bench_c5_${P}_trained.json 2>> gpurun_out/trained.err
done
timeout 400 python3 bench.py --workload C3 --precision fp32 --checkpoint /tmp/vfn_trained.pth > gpurun_out/r05_bench_c3_fp32_trained.json 2>> gpurun_out/trained.err
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --checkpoint /tmp/vfn_trained.pth > gpurun_out/r05_bench_c2_fp32_trained.json 2>> gpurun_out/trained.err
timeout 900 python3 scripts/profile_round.py fp32 > gpurun_out/profile_round_fp32.log 2>&1
timeout 300 python3 scripts/profile_layers.py > gpurun_out/r05_layers.txt 2>&1
timeout 300 python3 scripts/bench_train_step.py > gpurun_out/r05_train_step.txt 2>&1
timeout 400 python3 scripts/main_throughput.py > gpurun_out/r05_main_throughput.txt 2>&1
timeout 300 python3 scripts/bench_wino_transforms.py > gpurun_out/r05_transforms_final.txt 2>&1
tail -c 300 gpurun_out/*.err
for f in gpurun_out/r05_bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['dtype'][:12], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('all_conv_frac'), d.get('parity'))
except Exception as e: print('$f', 'FAILED', e)"; done
