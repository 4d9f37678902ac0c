"""Per-kernel time of ONE bench step from a rocprofv3 --kernel-trace CSV directory.
usage: python tools/step_breakdown.py <dir> [out.csv]
The trace holds every forward step bench.py ran (warm-up + timed + the uninstrumented repeat); the number of steps is read
from the trace itself -- one `stem_pool_kernel` launch per forward call (bench.py's own count of forward calls, the
`forward_steps_total` key of its JSON line, must agree; tests/test_profile_tools.py) -- never from a shell argument: round 4's
committed per-step figures were 1.77x high because the script's argument still said 13 when the bench had grown to 23."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

STEP_KERNEL = 'stem_pool_kernel'      # launched exactly once per forward call of the fp16 engine (csrc/stem.hip)


def short(name):
    name = re.sub(r'emp::\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN3emp12_GLOBAL__N_1(\d+)(.*)', name)
    if m:
        n = int(m.group(1))
        return m.group(2)[:n] + ('<' + m.group(2)[n:][:24] + '>' if 'I' in m.group(2)[n:n + 2] else '')
    return re.sub(r'^void ', '', name)[:90]


def trace_file(d):
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not f:
        raise FileNotFoundError(f'no *kernel_trace.csv under {d}')
    return f[0]


def steps_in_trace(rows):
    """Forward steps the trace holds = launches of the stem kernel.  rows: iterable of kernel names."""
    n = sum(1 for k in rows if STEP_KERNEL in k)
    if n == 0:
        raise ValueError(f'the trace holds no {STEP_KERNEL} launch: not a trace of the fp16 engine\'s forward')
    return n


def breakdown(d):
    """-> (steps, [(kernel, calls per step, us per step, share)], total us per step)"""
    tot, cnt = defaultdict(float), defaultdict(int)
    names = []
    for r in csv.DictReader(open(trace_file(d))):
        names.append(r['Kernel_Name'])
        k = short(r['Kernel_Name'])
        tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cnt[k] += 1
    steps = steps_in_trace(names)
    total = sum(tot.values())
    rows = [(k, cnt[k] / steps, v / steps, v / total) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])]
    return steps, rows, total / steps


def main():
    d = sys.argv[1]
    steps, rows, total = breakdown(d)
    lines = [f'# {steps} forward steps in the trace ({STEP_KERNEL} launches)', 'kernel,calls_per_step,us_per_step,share']
    for k, c, us, sh in rows:
        lines.append(f'"{k}",{c:.1f},{us:.1f},{sh:.4f}')
    lines.append(f'"TOTAL kernel time",,{total:.1f},1.0')
    out = '\n'.join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + '\n')


if __name__ == '__main__':
    main()
