cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short -k "reproducible or refresh" 2>&1 | tail -15 >> gpurun_out/j35_tests.log; done
