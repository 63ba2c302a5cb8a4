cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "data_gradient" 2>&1 | tail -30 > gpurun_out/j31_tests.log
VFN_SIDE_PRIORITY=-1 python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j31_train.txt 2>&1
python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/j31_train.txt 2>&1
