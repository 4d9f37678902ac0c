"""HBM traffic per bench step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
  python tools/hbm_traffic.py <fetch_dir> <write_dir> [out.json]
Each dir is the output of  rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py
--steps S --warmup W --no-cpu-baseline --stack3d 0 --engine2d 0 --latency 0.  The number of forward steps a pass holds is
read from its own kernel trace (launches of `stem_pool_kernel`, one per forward call), never from an argument (round 4's
file was 1.75x high because the argument was stale).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests
as 64 B, MI355X_MICROARCH.md); both counters are in KiB; Infinity-Cache hits are included."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'emp::\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN3emp12_GLOBAL__N_1(\d+)(.*)', name)
    if m:
        return m.group(2)[:int(m.group(1))]
    base = re.sub(r'^void ', '', name).split('(')[0]
    if base.startswith('conv_igemm256_kernel<') and base.rstrip().endswith('true>'):
        return 'conv_igemm256_kernel<b2b>'      # the back-to-back instantiation is another kernel (bench.py counts the plain launches)
    if base.startswith('conv_igemm256w_kernel'):
        return 'conv_igemm256_kernel'           # round 5: the whole-line form of the same tile takes the plain launches
    return base.split('<')[0][:60]


STEP_KERNEL = 'stem_pool_kernel'


def collect(d, counter):
    """-> {kernel: (launches per step, mean counter bytes per captured launch)}.  Kernel names come from the kernel
    trace by dispatch id (the counter CSV's own name column is not reliable when a pass drops samples) and the
    per-step launch count from the trace as well."""
    fc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    ft = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
    name, launches = {}, defaultdict(int)
    for r in csv.DictReader(open(ft)):
        name[r['Dispatch_Id']] = short(r['Kernel_Name'])
        launches[short(r['Kernel_Name'])] += 1
    steps = launches.get(STEP_KERNEL, 0)
    if steps == 0:
        raise ValueError(f'{ft} holds no {STEP_KERNEL} launch: not a trace of the fp16 engine\'s forward')
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(fc)):
        if r['Counter_Name'] == counter and r['Dispatch_Id'] in name:
            k = name[r['Dispatch_Id']]
            tot[k] += float(r['Counter_Value']) * 1024.0
            cnt[k] += 1
    return {k: (launches[k] / steps, tot[k] / cnt[k]) for k in tot}, steps


def traffic(fd, wd):
    fe, fsteps = collect(fd, 'FETCH_SIZE')
    wr, wsteps = collect(wd, 'WRITE_SIZE')
    per = {}
    for k in set(fe) | set(wr):
        lps = fe[k][0] if k in fe else wr[k][0]
        per[k] = {'launches_per_step': lps, 'fetch_corrected': 2 * fe.get(k, (0, 0))[1] * lps,
                  'write': wr.get(k, (0, 0))[1] * lps}
    per = dict(sorted(per.items(), key=lambda kv: -(kv[1]['fetch_corrected'] + kv[1]['write'])))
    setup = {k: per.pop(k) for k in list(per) if 'fillBufferAligned' in k}      # one-time arena zeroing at reserve()
    return per, setup, (fsteps, wsteps)


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    per, setup, steps = traffic(fd, wd)
    out = {'note': __doc__.split('\n')[0] + ' FETCH_SIZE doubled (gfx950), Infinity-Cache hits included; bytes per step; '
                   'the runtime fill kernel (arena zeroing at reserve, once per process) is listed under setup.',
           'steps_in_trace': {'fetch_pass': steps[0], 'write_pass': steps[1]},
           'hbm_bytes_per_step': sum(v['fetch_corrected'] + v['write'] for v in per.values()),
           'setup': setup,
           'per_kernel': per}
    print(json.dumps({k: out[k] for k in ('hbm_bytes_per_step',)}))
    for k, v in list(per.items())[:12]:
        print(f"{k:40s} {v['launches_per_step']:6.1f} launches  fetch {v['fetch_corrected'] / 1e9:7.2f} GB  write {v['write'] / 1e9:7.2f} GB")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
