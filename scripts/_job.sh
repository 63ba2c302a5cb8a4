cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "column_sums" 2>&1 | tail -8 > gpurun_out/j20_tests.log
python -m pytest tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/j20_tests.log
python scripts/bench_train_step.py 6 400 400 2 6 > gpurun_out/j20_train.txt 2>&1
VFN_COLSUM_BLOCKS=64 python scripts/bench_train_step.py 6 400 400 2 6 >> gpurun_out/j20_train.txt 2>&1
bash scripts/profile_train.sh j20
