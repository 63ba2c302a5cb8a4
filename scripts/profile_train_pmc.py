"""PMC counters of the training step's kernels (rocprofv3 --pmc, one pass per counter group, as scripts/profile_round.py does for the
bench): HBM bytes per launch (FETCH_SIZE doubled: MI355X_MICROARCH.md, section HBM), MFMA utilisation, wait share -- with the weight
gradients on the main stream (VFN_SIDE_WGRAD=0) so that every kernel is alone on the device.  -> gpurun_out/r04_train_pmc.json"""
import os, sys, csv, glob, json, shutil, subprocess, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out')
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp', VFN_SIDE_WGRAD='0')
cmd_tail = ['python3', os.path.join(root, 'scripts', 'bench_train_step.py'), '6', '400', '400', '2', '2']
passes = {'fetch': ['FETCH_SIZE'], 'write': ['WRITE_SIZE'],
          'sq': ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'GRBM_GUI_ACTIVE']}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for tag, counters in passes.items():
    d = f'/tmp/vfn_train_pmc_{tag}'
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(['rocprofv3', '--pmc'] + counters + ['--output-format', 'csv', '-d', d, '-o', tag, '--'] + cmd_tail, cwd='/tmp', env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(tag, 'rc', r.returncode, [l for l in r.stdout.splitlines() if 'train step' in l][-1:], flush=True)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            k, c = row['Kernel_Name'], row['Counter_Name']
            acc[k][c] += float(row['Counter_Value'])
            launches[k][c] += 1
kern = {}
for k in acc:
    a, n = acc[k], launches[k]
    per = lambda c: a[c] / n[c] if n.get(c) else None
    e = {'launches': max(n.values())}
    f, w = per('FETCH_SIZE'), per('WRITE_SIZE')
    if f is not None and w is not None:
        e.update(fetch_kib_raw_per_launch=round(f, 1), write_kib_per_launch=round(w, 1), hbm_bytes_per_launch=round((2 * f + w) * 1024))
    if n.get('GRBM_GUI_ACTIVE') and a['GRBM_GUI_ACTIVE'] > 0:
        e['mfma_util'] = round((a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (a['GRBM_GUI_ACTIVE'] / 8), 3)
        e['gpu_active_cycles_total'] = a['GRBM_GUI_ACTIVE']
    if a.get('SQ_WAVE_CYCLES'):
        e['wait_any_per_wave_cycle'] = round(a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES'], 3)
    kern[k] = e
top = dict(sorted(kern.items(), key=lambda kv: -kv[1].get('gpu_active_cycles_total', 0))[:24])
note = ('rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ+GRBM, one pass each) of `VFN_SIDE_WGRAD=0 python3 scripts/bench_train_step.py 6 400 400 2 2`; '
        'per-kernel means per launch over the whole run (5 steps incl. the first); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); mfma_util = '
        'SQ_VALU_MFMA_BUSY_CYCLES/1024 SIMDs / (GRBM_GUI_ACTIVE/8 XCDs); the 24 kernels with the most active cycles')
json.dump({'note': note, 'kernels': top}, open(os.path.join(out, 'r04_train_pmc.json'), 'w'), indent=1)
print('wrote r04_train_pmc.json', len(kern), 'kernels')
