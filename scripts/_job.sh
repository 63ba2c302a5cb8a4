cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py tests/test_backward_gpu.py -q -m gpu --tb=short 2>&1 | tail -12 > gpurun_out/j61_tests.log
python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j61_train.txt 2>&1
