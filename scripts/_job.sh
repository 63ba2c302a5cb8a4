cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "wgrad or column" 2>&1 | tail -12 > gpurun_out/j39_tests.log
python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j39_train.txt 2>&1
VFN_WGRAD_INLAUNCH=0 python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/j39_train.txt 2>&1
VFN_WGRAD_INLAUNCH=0 VFN_TRAIN_CHAIN_PRIORITY=-1 python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/j39_train.txt 2>&1
python -m pytest tests/test_round4_gpu.py tests/test_backward_gpu.py -q -m gpu --tb=short 2>&1 | tail -12 >> gpurun_out/j39_tests.log
