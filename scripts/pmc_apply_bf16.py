"""rocprofv3 --pmc passes over scripts/bench_apply_bf16.py (the plain-bf16 apply kernels alone at a C5-size bank): matrix-pipe busy
fraction, wave-cycle shares, LDS conflicts, HBM-side bytes per launch.  This process never touches the GPU.
Usage: pmc_apply_bf16.py [entries]   -> gpurun_out/r06_pmc_apply_bf16.json"""
import csv, glob, json, os, shutil, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out')
os.makedirs(out, exist_ok=True)
B = sys.argv[1] if len(sys.argv) > 1 else '660000'
env = dict(os.environ, TMPDIR='/tmp')
passes = {'sq': ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'],
          'sq2': ['SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM'],
          'fetch': ['FETCH_SIZE'], 'write': ['WRITE_SIZE'], 'tcc': ['TCC_HIT_sum', 'TCC_MISS_sum']}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for tag, counters in passes.items():
    d = f'/tmp/vfn_pmc_apply_{tag}'
    shutil.rmtree(d, ignore_errors=True)
    cmd = ['rocprofv3', '--pmc'] + counters + ['--output-format', 'csv', '-d', d, '-o', tag, '--', 'python3', os.path.join(root, 'scripts', 'bench_apply_bf16.py'), B]
    r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(tag, 'rc', r.returncode, r.stdout.strip().splitlines()[-1][:160] if r.stdout.strip() else '', flush=True)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'memread_apply' not in k:
                continue
            # only the launches at the big bank: the bit-identity part of the script runs tiny banks first (grid size tells)
            if int(row.get('Grid_Size', row.get('Grid_Size_X', 0)) or 0) < 100000:
                continue
            acc[k][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[k][row['Counter_Name']] += 1
res = {}
for k in acc:
    a, n = acc[k], cnt[k]
    per = lambda c: a[c] / n[c] if n.get(c) else None
    e = {'launches': max(n.values())}
    if per('GRBM_GUI_ACTIVE'):
        e['mfma_util'] = round((per('SQ_VALU_MFMA_BUSY_CYCLES') / 1024) / (per('GRBM_GUI_ACTIVE') / 8), 3)
        e['gui_active_cycles_per_xcd'] = round(per('GRBM_GUI_ACTIVE') / 8)
    if per('SQ_WAVE_CYCLES'):
        e['wait_any_per_wave_cycle'] = round(per('SQ_WAIT_ANY') / per('SQ_WAVE_CYCLES'), 3)
    if per('SQ_LDS_IDX_ACTIVE'):
        e['lds_bank_conflict_frac'] = round(per('SQ_LDS_BANK_CONFLICT') / per('SQ_LDS_IDX_ACTIVE'), 3)
    for c in ('SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM', 'TCC_HIT_sum', 'TCC_MISS_sum'):
        if per(c) is not None:
            e[c] = round(per(c))
    if per('FETCH_SIZE') is not None:
        e['hbm_fetch_bytes_per_launch'] = round(2 * 1024 * per('FETCH_SIZE'))      # KiB; doubled (MI355X_MICROARCH.md, section HBM)
    if per('WRITE_SIZE') is not None:
        e['hbm_write_bytes_per_launch'] = round(1024 * per('WRITE_SIZE'))
    res[k] = e
json.dump({'entries_per_object': int(B), 'note': 'rocprofv3 --pmc, one pass per counter group, launches of scripts/bench_apply_bf16.py at the given bank size only; '
           'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)', 'kernels': res}, open(os.path.join(out, 'r06_pmc_apply_bf16.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
