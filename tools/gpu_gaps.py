"""GPU idle gaps of a rocprofv3 --kernel-trace run: union of the kernel intervals over all streams inside the last
`frac` of the trace, total busy / idle time and the largest gaps with the kernels around them.
    python tools/gpu_gaps.py <trace_dir> [frac=0.5]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - int((t1 - t0) * frac)
rows = [r for r in rows if r[0] >= lo]
busy, gaps = 0, []
cur_s, cur_e, last_name = rows[0][0], rows[0][1], rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = n
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f'window {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps; sum of kernel '
      f'durations {sum(e - s for s, e, _ in rows) / 1e6:.2f} ms')
short = lambda n: n.split('(')[0][-48:]
for g, a, b in sorted(gaps, reverse=True)[:14]:
    print(f'  {g / 1e3:9.1f} us  after {short(a):48s} before {short(b)}')
small = sum(g for g, _, _ in gaps if g < 20000)
print(f'  gaps < 20 us: {sum(1 for g, _, _ in gaps if g < 20000)} totalling {small / 1e6:.2f} ms')

import collections
tot = collections.Counter()
cnt = collections.Counter()
for s_, e_, n in rows:
    tot[n] += e_ - s_
    cnt[n] += 1
print('kernels inside the window by total time:')
for n, t in tot.most_common(34):
    print(f'  {t / 1e6:8.2f} ms  {cnt[n]:5d} x  {n.split("(")[0][-90:]}')
