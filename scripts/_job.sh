cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -80 > gpurun_out/r04_gpu_suite.log
