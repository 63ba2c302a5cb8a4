#!/bin/bash
# HBM-side fetch bytes of the bank kernels at a given bank size / precision (run on the GPU box):
#   scripts/pmc_fetch_bank.sh <entries> <prec 0|1|2>   -> stdout
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf
PREC=${2:-0} timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf -o b -- python3 $GRAFT_REPO_ROOT/scripts/bench_bank_kernels.py ${1:-56000} > /tmp/pmcf.log 2>&1
tail -n 2 /tmp/pmcf.log
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/pmcf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
for k in acc:
    if 'bank_scan' in k or 'memread_apply' in k:
        # FETCH_SIZE is in KiB-like units of 1 KB on this stack and under-reports by 2x on gfx950 (MI355X_MICROARCH.md): x2
        print('%-70s launches %4d  fetch per launch %.3f GB' % (k[:70], n[k], 2 * acc[k] * 1024 / n[k] / 1e9))
PY
