#!/bin/bash
# rocprofv3 kernel stats of the training step (scripts/bench_train_step.py) -> gpurun_out/<tag>_train_kernel_stats.csv
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vfn_prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vfn_prof_train -o train -- python3 $GRAFT_REPO_ROOT/scripts/bench_train_step.py 6 400 400 2 4 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_train_step_profiled.txt 2>&1
cp $(find /tmp/vfn_prof_train -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${TAG}_train_kernel_stats.csv
