# Round-6 evidence set (run on the GPU box through gpurun; everything lands in gpurun_out/, copy into profiles/ afterwards)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
B="python3 bench.py"
timeout 300 $B --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd.json 2> $O/driver.err
timeout 300 $B --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_driver_cmd_2.json 2>> $O/driver.err
timeout 400 $B > $O/r06_bench_default.json 2> $O/default.err
timeout 400 $B --workload C3 --precision bf16x3 > $O/r06_bench_c3_bf16x3.json 2> $O/c3x3.err
timeout 400 $B --workload C3 --precision bf16 > $O/r06_bench_c3_bf16.json 2> $O/c3.err
timeout 400 $B --workload C3 --precision fp32 > $O/r06_bench_c3_fp32.json 2> $O/c3f.err
# the same workload with the frames of a key-frame interval as one batched pass (ClipRunner.launch_group)
for P in bf16 bf16x3 fp32; do timeout 400 $B --workload C3 --precision $P --group > $O/r06_bench_c3_${P}_grouped.json 2>> $O/c3g.err; done
timeout 900 $B --workload C5 --precision bf16x3 --steps 2000 --warmup 2 > $O/r06_bench_c5_bf16x3.json 2> $O/c5.err
timeout 900 $B --workload C5 --precision bf16 --steps 2000 --warmup 2 > $O/r06_bench_c5_bf16.json 2> $O/c5b.err
VFN_APPLY_PIPE=0 VFN_SCAN_PIPE=0 timeout 900 $B --workload C5 --precision bf16 --steps 2000 --warmup 2 --no-cpu-baseline > $O/r06_bench_c5_bf16_round5_kernels.json 2>> $O/c5b.err
for T in easy hard; do
# (each checkpoint is benchmarked on ITS task's frames: the hard-task checkpoint on the easy clip is out of distribution)
if [ $T = easy ]; then N=trained_easy; else N=hard_task; fi
timeout 600 python3 scripts/train_ckpt.py $T 3000 /tmp/vfn_trained_$T.pth > $O/r06_train_ckpt_$T.log 2>&1
timeout 400 $B --workload C3 --precision bf16 --clip $T --checkpoint /tmp/vfn_trained_$T.pth > $O/r06_bench_c3_bf16_$N.json 2>> $O/trained.err
timeout 400 $B --workload C3 --precision bf16 --group --clip $T --checkpoint /tmp/vfn_trained_$T.pth > $O/r06_bench_c3_bf16_${N}_grouped.json 2>> $O/trained.err
timeout 900 $B --workload C5 --precision bf16 --steps 2000 --warmup 2 --clip $T --checkpoint /tmp/vfn_trained_$T.pth > $O/r06_bench_c5_bf16_$N.json 2>> $O/trained.err
done
timeout 400 $B --workload C3 --precision bf16x3 --clip hard --checkpoint /tmp/vfn_trained_hard.pth > $O/r06_bench_c3_bf16x3_hard_task.json 2>> $O/trained.err
timeout 400 $B --workload C3 --precision bf16x3 --group --clip hard --checkpoint /tmp/vfn_trained_hard.pth > $O/r06_bench_c3_bf16x3_hard_task_grouped.json 2>> $O/trained.err
timeout 400 $B --gpus 1 --steps 20 --warmup 5 --clip hard --checkpoint /tmp/vfn_trained_hard.pth > $O/r06_bench_c2_fp32_hard_task.json 2>> $O/trained.err
timeout 900 $B --workload C5 --precision bf16x3 --steps 2000 --warmup 2 --checkpoint /tmp/vfn_trained_easy.pth > $O/r06_bench_c5_bf16x3_trained_easy.json 2>> $O/trained.err
timeout 400 $B --gpus 1 --steps 20 --warmup 5 --checkpoint /tmp/vfn_trained_easy.pth > $O/r06_bench_c2_fp32_trained_easy.json 2>> $O/trained.err
timeout 900 python3 scripts/profile_round.py fp32 > $O/profile_round_fp32.log 2>&1
timeout 900 python3 scripts/profile_round.py bf16 > $O/profile_round_bf16.log 2>&1
timeout 300 python3 scripts/profile_layers.py > $O/r06_layers.txt 2>&1
timeout 300 python3 scripts/bench_train_step.py > $O/r06_train_step.txt 2>&1
timeout 300 python3 scripts/bench_apply_bf16.py > $O/r06_apply_pipe_ab.txt 2>&1
timeout 300 python3 scripts/bench_scan_bf16.py > $O/r06_scan_pipe_ab.txt 2>&1
(VFN_CKPT=/tmp/vfn_trained_easy.pth timeout 600 python3 scripts/bench_group.py bf16 5 100; timeout 600 python3 scripts/bench_group.py fp32 5 100) > $O/r06_group_vs_frame_by_frame.txt 2>&1
timeout 600 python3 scripts/group_parity_oracle.py 41 > $O/group_parity_oracle.log 2>&1
timeout 600 python3 scripts/pmc_apply_bf16.py 660000 > $O/pmc_apply.log 2>&1
(timeout 300 python3 scripts/main_throughput.py 100; timeout 300 python3 scripts/main_throughput.py 200; timeout 300 python3 scripts/main_throughput.py 200 0) 2>&1 | grep "frames/s" > $O/r06_main_throughput_final.txt
tail -c 300 $O/*.err
for f in $O/r06_bench_*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], d['dtype'][:12], r.get('kernel'), r.get('frac'), r.get('all_conv_frac'), d.get('parity'))
except Exception as e: print('$f', 'FAILED', e)"; done
cat $O/r06_main_throughput_final.txt
