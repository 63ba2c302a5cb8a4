cd $GRAFT_REPO_ROOT
python -m pytest tests/test_jpeg.py tests/test_linknet.py tests/test_round4_gpu.py -q -m gpu --durations=6 2>&1 | tail -16 > gpurun_out/j56_tests.log
