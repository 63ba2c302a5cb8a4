"""One steady-state frame of video_seg.main (files -> files) as the device saw it: rocprofv3 --kernel-trace of scripts/main_throughput.py,
then the kernels between two late memory-read launches by stream: busy time per stream, totals by kernel name, and the launches in start
order (run on the GPU box).  usage: trace_main.py [out.txt]"""
import csv, glob, os, shutil, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'gpurun_out', 'trace_main.txt')
d = '/tmp/vfn_trace_main'
shutil.rmtree(d, ignore_errors=True)
cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 't', '--', 'python3', os.path.join(root, 'scripts', 'main_throughput.py'), '60']
r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print(r.stdout[-300:])
rows = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r_ in csv.DictReader(open(f)):
        rows.append((int(r_['Start_Timestamp']), int(r_['End_Timestamp']), r_['Kernel_Name'], r_.get('Queue_Id', '?'), r_.get('Stream_Id', '?')))
rows.sort()
idx = [i for i, r_ in enumerate(rows) if 'memread_apply' in r_[2]]
a, b = idx[-12], idx[-2]                       # ten frames near the end of the timed run
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')
with open(out, 'w') as fo:
    span = (rows[b][0] - rows[a][0]) / 1e3
    fo.write('span %.1f us over 10 frames (%.1f us per frame); %d kernels\n' % (span, span / 10, b - a))
    by_s = collections.defaultdict(lambda: [0, 0.0])
    by_k = collections.defaultdict(lambda: [0, 0.0, set()])
    for s, e, n, q, st in rows[a:b]:
        by_s[st][0] += 1; by_s[st][1] += (e - s) / 1e3
        k = short(n).split('(')[0][:80]
        by_k[k][0] += 1; by_k[k][1] += (e - s) / 1e3; by_k[k][2].add(st)
    for st, (c, t) in sorted(by_s.items(), key=lambda kv: -kv[1][1]):
        fo.write('stream s%s: %d kernels, busy %.1f us per frame\n' % (st, c, t / 10))
    fo.write('\n')
    for k, (c, t, sts) in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:50]:
        fo.write('  %9.1f us/frame  %5.1f x  s%-8s %s\n' % (t / 10, c / 10, ','.join(sorted(sts)), k))
    fo.write('\n')
    a2 = idx[-3]
    prev_end = rows[a2][0]
    for s, e, n, q, st in rows[a2:b]:
        fo.write('%9.1f  dur %8.1f  gap %7.1f  s%-3s %s\n' % ((s - rows[a2][0]) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, st, short(n)[:100]))
        prev_end = max(prev_end, e)
print(open(out).read()[:5000])
