"""One steady-state frame as the device saw it: rocprofv3 --kernel-trace of a short bench run, then the kernels of the last
complete frame in start order with duration, queue and the gap to the previous kernel's end (run on the GPU box).
usage: trace_frame.py [out.txt] [extra bench args ...]   (e.g. --no-overlap)"""
import csv, glob, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'gpurun_out', 'trace_frame.txt')
extra = sys.argv[2:]
d = '/tmp/vfn_trace_frame'
shutil.rmtree(d, ignore_errors=True)
cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 't', '--', 'python3', os.path.join(root, 'bench.py'),
       '--steps', '12', '--warmup', '2', '--min-warm-s', '0', '--min-timed-s', '0', '--no-autotune', '--no-cpu-baseline', '--sample-every', '1000'] + extra
r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-600:])
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r_ in csv.DictReader(open(f)):
        rows.append((int(r_['Start_Timestamp']), int(r_['End_Timestamp']), r_['Kernel_Name'], r_.get('Queue_Id', '?'), r_.get('Stream_Id', '?')))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r_ in csv.DictReader(open(f)):
        rows.append((int(r_['Start_Timestamp']), int(r_['End_Timestamp']), 'MEMCPY ' + r_.get('Direction', '') + ' ' + r_.get('Bytes', r_.get('Size', '?')), 'copy', '-'))
rows.sort()
# frame boundary: memread_apply launches; take the span between the last two but one
idx = [i for i, r_ in enumerate(rows) if 'memread_apply' in r_[2]]
a, b = idx[-3], idx[-2]
with open(out, 'w') as fo:
    prev_end = rows[a][0]
    tot = 0
    fo.write('span %.1f us between two apply launches; %d kernels\n' % ((rows[b][0] - rows[a][0]) / 1e3, b - a))
    for s, e, n, q, st in rows[a:b]:
        short = n.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        fo.write('%9.1f  dur %8.1f  gap %7.1f  q%-3s s%-3s %s\n' % ((s - rows[a][0]) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st, short))
        prev_end = max(prev_end, e)
        tot += e - s
    fo.write('sum of durations %.1f us\n' % (tot / 1e3))
print(open(out).read()[-1500:])
