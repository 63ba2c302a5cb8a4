cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/j50_tests.log
python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j50_train.txt 2>&1
VFN_WINOGRAD_WGRAD=0 python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/j50_train.txt 2>&1
