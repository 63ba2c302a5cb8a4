"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (all streams merged): how much of the wall
clock no kernel was running.  usage: gap_report.py <dir with *_kernel_trace.csv> [skip_first_n_kernels]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 3
rows = rows[skip:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy = 0; cur_end = rows[0][0]; gaps = []; after = collections.defaultdict(lambda: [0, 0])
prev_name = None
for s, e, n in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        if prev_name: after[prev_name[:60]][0] += s - cur_end; after[prev_name[:60]][1] += 1
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev_name = n
wall = t1 - t0
print('kernels %d  wall %.2f ms  busy %.2f ms (%.1f %%)  idle %.2f ms in %d gaps (median %.2f us, mean %.2f us)' % (
    len(rows), wall / 1e6, busy / 1e6, 100 * busy / wall, (wall - busy) / 1e6, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3, sum(gaps) / max(1, len(gaps)) / 1e3))
import numpy as np
g = np.array(gaps) / 1e3
for lo, hi in [(0, 1), (1, 2), (2, 4), (4, 8), (8, 20), (20, 100), (100, 1e9)]:
    m = (g >= lo) & (g < hi)
    print('  gaps %5g-%-5g us: %6d  total %.2f ms' % (lo, hi, m.sum(), g[m].sum() / 1e3))
print('idle after (top 12):')
for k, v in sorted(after.items(), key=lambda kv: -kv[1][0])[:12]:
    print('  %-60s %8.2f ms over %5d gaps (%.2f us each)' % (k, v[0] / 1e6, v[1], v[0] / v[1] / 1e3))
