cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/j26_tests.log
python -m pytest tests/test_backward_gpu.py tests/test_conv_gpu.py -x -q -m gpu 2>&1 | tail -30 >> gpurun_out/j26_tests.log
python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j26_train.txt 2>&1
VFN_WINOGRAD_TRAIN=0 python scripts/bench_train_step.py 6 400 400 2 8 >> gpurun_out/j26_train.txt 2>&1
python scripts/tune_winograd.py 400x400 > gpurun_out/j26_tune_wino.txt 2>&1
