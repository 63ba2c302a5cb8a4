cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/j51_tests.log
bash scripts/profile_train.sh j51
VFN_SIDE_DROP=1 python scripts/bench_train_step.py 6 400 400 2 8 > gpurun_out/j51_train.txt 2>&1
