#!/bin/bash
# PMC pass over the bank-kernel micro-benchmark (run on the GPU box): per-kernel SQ counters -> gpurun_out/r02_pmc_bank.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcb
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcb -o b -- python3 $GRAFT_REPO_ROOT/scripts/bench_bank_kernels.py 56000 > /tmp/pmcb.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('/tmp/pmcb/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']][r['Counter_Name']] += 1
for k in acc:
    a = acc[k]; c = max(n[k].values())
    if 'bank_scan' not in k and 'memread_apply' not in k: continue
    wc = a['SQ_WAVE_CYCLES'] or 1
    print(k[:70], 'launches', c)
    print('   mfma_util %.3f  (MFMA_BUSY/1024 / (GUI_ACTIVE/8))' % ((a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (a['GRBM_GUI_ACTIVE'] / 8)))
    print('   per wave-cycle: wait_any %.3f  wait_inst_any %.3f  active_inst_any %.3f  wait_inst_lds %.3f ; lds_conflict cycles/launch %.0f ; gui_active/8 per launch %.0f cycles' % (
        a['SQ_WAIT_ANY'] / wc, a['SQ_WAIT_INST_ANY'] / wc, a['SQ_ACTIVE_INST_ANY'] / wc, a['SQ_WAIT_INST_LDS'] / wc, a['SQ_LDS_BANK_CONFLICT'] / c, a['GRBM_GUI_ACTIVE'] / 8 / c))
PY
