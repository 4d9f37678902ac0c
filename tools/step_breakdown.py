"""Per-kernel time of ONE bench step from a rocprofv3 --kernel-trace CSV directory.
usage: python tools/step_breakdown.py <dir> <steps_in_trace> [out.csv]
The trace holds warm-up + timed steps; every step launches the same kernels, so per-step time = total / steps."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'emp::\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN3emp12_GLOBAL__N_1(\d+)(.*)', name)
    if m:
        n = int(m.group(1))
        return m.group(2)[:n] + ('<' + m.group(2)[n:][:24] + '>' if 'I' in m.group(2)[n:n + 2] else '')
    return re.sub(r'^void ', '', name)[:90]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cnt[k] += 1
    total = sum(tot.values())
    rows = sorted(tot.items(), key=lambda kv: -kv[1])
    lines = ['kernel,calls_per_step,us_per_step,share']
    for k, v in rows:
        lines.append(f'"{k}",{cnt[k] / steps:.1f},{v / steps:.1f},{v / total:.4f}')
    lines.append(f'"TOTAL kernel time",,{total / steps:.1f},1.0')
    out = '\n'.join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(out + '\n')


if __name__ == '__main__':
    main()
