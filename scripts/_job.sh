cd $GRAFT_REPO_ROOT
python scripts/main_throughput.py 100 1 > gpurun_out/j43_main.txt 2>&1
python scripts/main_throughput.py 100 1 >> gpurun_out/j43_main.txt 2>&1
