"""GPU busy fraction of the frame loop from a rocprofv3 kernel trace (run on the GPU box):
union of kernel intervals / wall, and the distribution of idle gaps between consecutive kernels."""
import csv, glob, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = '/tmp/vfn_gap'
shutil.rmtree(d, ignore_errors=True)
extra = sys.argv[1:]
cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'g', '--', 'python3', os.path.join(root, 'bench.py'),
       '--steps', '40', '--warmup', '2', '--no-autotune', '--no-cpu-baseline', '--sample-every', '1000'] + extra
r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [(int(r_['Start_Timestamp']), int(r_['End_Timestamp']), r_['Kernel_Name']) for r_ in csv.DictReader(open(f))]
rows.sort()
# timed region = last 40 frames: take the last 75% of kernels
rows = rows[len(rows) // 4:]
t0, t1 = rows[0][0], max(e for _, e, _ in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('kernels', len(rows), 'wall %.2f ms' % ((t1 - t0) / 1e6), 'busy %.1f %%' % (100 * busy / (t1 - t0)))
gaps.sort()
n = len(gaps)
print('gaps: n %d, total %.2f ms, median %.2f us, p90 %.2f us, max %.1f us' % (n, sum(gaps) / 1e6, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[-1] / 1e3))
big = [g for g in gaps if g > 20000]
print('gaps > 20 us: %d totalling %.2f ms' % (len(big), sum(big) / 1e6))
