cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py -x -q -m gpu -k "refresh" 2>&1 | tail -15 > gpurun_out/j22_tests.log
python scripts/bench_train_step.py 6 400 400 2 6 > gpurun_out/j22_train.txt 2>&1
bash scripts/profile_train.sh j22
python -m pytest tests/test_backward_gpu.py -x -q -m gpu 2>&1 | tail -8 >> gpurun_out/j22_tests.log
