cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=short 2>&1 | tail -30 > gpurun_out/j45_tests.log
python scripts/tune_train.py > gpurun_out/j45_tune_train.txt 2>&1
