"""One training step as the device saw it: rocprofv3 --kernel-trace of scripts/bench_train_step.py, then the kernels between the last
two AdamW launches in start order with duration, stream and the gap to the previous kernel's end, and the totals by kernel name
(run on the GPU box).  usage: trace_train.py [out.txt]   (VFN_SIDE_DROP=1 in the environment: the dependent chain alone)"""
import csv, glob, os, shutil, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'gpurun_out', 'trace_train.txt')
d = '/tmp/vfn_trace_train'
shutil.rmtree(d, ignore_errors=True)
cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 't', '--', 'python3', os.path.join(root, 'scripts', 'bench_train_step.py'),
       '6', '400', '400', '2', '4']
r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-400:])
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r_ in csv.DictReader(open(f)):
        rows.append((int(r_['Start_Timestamp']), int(r_['End_Timestamp']), r_['Kernel_Name'], r_.get('Queue_Id', '?'), r_.get('Stream_Id', '?')))
rows.sort()
idx = [i for i, r_ in enumerate(rows) if 'adamw' in r_[2]]
a, b = idx[-2] + 1, idx[-1] + 1
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')
with open(out, 'w') as fo:
    span = (rows[b - 1][1] - rows[a][0]) / 1e3
    fo.write('span %.1f us between two AdamW launches; %d kernels\n' % (span, b - a))
    streams = collections.Counter(r_[4] for r_ in rows[a:b])
    main = streams.most_common(1)[0][0]
    by, gaps, prev_end = collections.defaultdict(lambda: [0, 0.0]), 0.0, rows[a][0]
    for s, e, n, q, st in rows[a:b]:
        if st != main:
            continue
        k = short(n).split('(')[0][:90]
        by[k][0] += 1
        by[k][1] += (e - s) / 1e3
        gaps += max(0, s - prev_end) / 1e3
        prev_end = max(prev_end, e)
    fo.write('main stream s%s: %d kernels, durations %.1f us, idle gaps %.1f us; other streams: %s\n' % (
        main, streams[main], sum(v[1] for v in by.values()), gaps, {k: v for k, v in streams.items() if k != main}))
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:60]:
        fo.write('  %9.1f us  %4d x  %s\n' % (t, c, k))
    fo.write('\n')
    prev_end = rows[a][0]
    for s, e, n, q, st in rows[a:b]:
        fo.write('%9.1f  dur %8.1f  gap %7.1f  s%-3s %s\n' % ((s - rows[a][0]) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, st, short(n)[:110]))
        prev_end = max(prev_end, e)
print(open(out).read()[:6000])
