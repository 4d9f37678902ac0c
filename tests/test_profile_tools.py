"""tools/step_breakdown.py and tools/hbm_traffic.py fold a rocprofv3 trace of bench.py to per-step figures.  Round 4's
committed files were 1.77x / 1.75x high because the number of steps was a shell argument that went stale:
both tools now count the steps in the trace itself (stem_pool_kernel launches).  Here: synthetic traces through both tools,
and the committed profiles of the current round checked for physical plausibility."""
import csv
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

STEM = '_ZN3emp12_GLOBAL__N_116stem_pool_kernelIhEEvPKT_ffiiiiiPKfS6_PDF16_i'
CONV = 'void emp::(anonymous namespace)::conv_igemm256w_kernel(emp::ConvParams)'      # round 5: the whole-line form takes the plain launches
B2B = 'void emp::(anonymous namespace)::conv_igemm256_kernel<0, true>(emp::ConvParams)'
FILL = '__amd_rocclr_fillBufferAligned'


def _write_trace(d, steps, with_counter=None, conv_bytes_kib=1000.0):
    os.makedirs(d, exist_ok=True)
    rows, t, did = [], 1000, 0
    rows.append((did, FILL, t, t + 5000)); did += 1; t += 6000
    for _ in range(steps):
        rows.append((did, STEM, t, t + 500_000)); did += 1; t += 501_000
        for _ in range(29):
            rows.append((did, CONV, t, t + 400_000)); did += 1; t += 401_000
        for _ in range(2):
            rows.append((did, B2B, t, t + 540_000)); did += 1; t += 541_000
    with open(os.path.join(d, 'x_kernel_trace.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kind', 'Dispatch_Id', 'Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        for did, name, a, b in rows:
            w.writerow(['KERNEL_DISPATCH', did, name, a, b])
    if with_counter:
        with open(os.path.join(d, 'x_counter_collection.csv'), 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
            for did, name, a, b in rows:
                # the counter CSV's own name column is deliberately wrong: names come from the trace by dispatch id
                w.writerow([did, 'garbage', with_counter, conv_bytes_kib if name == CONV else 10.0])


@pytest.mark.parametrize('steps', [4, 7, 23])
def test_step_breakdown_counts_steps_in_the_trace(tmp_path, steps):
    import step_breakdown as sb
    d = str(tmp_path / 'ks')
    _write_trace(d, steps)
    n, rows, total = sb.breakdown(d)
    assert n == steps
    per = {k: (c, us) for k, c, us, _ in rows}
    conv = [v for k, v in per.items() if k.startswith('conv_igemm256w_kernel')][0]
    assert conv[0] == 29 and abs(conv[1] - 29 * 400.0) < 1e-6
    b2b = [v for k, v in per.items() if k.startswith('conv_igemm256_kernel<0, true>')][0]
    assert b2b[0] == 2
    # the fill kernel is setup: it ran once whatever the number of steps
    assert abs(total - (500.0 + 29 * 400.0 + 2 * 540.0 + 5.0 / steps)) < 1e-6


def test_step_breakdown_refuses_a_trace_without_the_stem(tmp_path):
    import step_breakdown as sb
    d = str(tmp_path / 'ks')
    _write_trace(d, 0)
    with pytest.raises(ValueError):
        sb.breakdown(d)


def test_hbm_traffic_counts_steps_per_pass(tmp_path):
    import hbm_traffic as ht
    fd, wd = str(tmp_path / 'pf'), str(tmp_path / 'pw')
    _write_trace(fd, 4, 'FETCH_SIZE', conv_bytes_kib=200.0)
    _write_trace(wd, 7, 'WRITE_SIZE', conv_bytes_kib=100.0)       # the two passes need not hold the same number of steps
    per, setup, steps = ht.traffic(fd, wd)
    assert steps == (4, 7)
    k = per['conv_igemm256_kernel']
    assert k['launches_per_step'] == 29
    assert abs(k['fetch_corrected'] - 2 * 200.0 * 1024 * 29) < 1e-6      # FETCH_SIZE doubled (gfx950), KiB -> B
    assert abs(k['write'] - 100.0 * 1024 * 29) < 1e-6
    assert per['conv_igemm256_kernel<b2b>']['launches_per_step'] == 2
    assert any('fillBufferAligned' in s for s in setup)


def _latest(pattern):
    f = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    return f[-1] if f else None


def test_committed_hbm_traffic_is_physical():
    """the newest committed traffic profile that bench.py may quote: 29 launches of the dominant kernel per step and a
    whole-step byte count below what HBM can move in the measured step time"""
    p = _latest('r*_hbm_traffic.json')
    tj = json.load(open(p))
    if 'steps_in_trace' not in tj:
        pytest.skip(f'{os.path.basename(p)} predates the trace-derived step count (bench.py never quotes it)')
    assert tj['per_kernel']['conv_igemm256_kernel']['launches_per_step'] == 29
    b = _latest('r*_bench.json')
    ms = json.load(open(b))['ms_per_step'] if b and os.path.basename(b)[:3] == os.path.basename(p)[:3] else 24.8
    assert tj['hbm_bytes_per_step'] / (ms * 1e-3) < 8e12, (tj['hbm_bytes_per_step'], ms)


def test_committed_step_breakdown_fits_in_the_step():
    p = _latest('r*_step_breakdown.csv')
    lines = open(p).read().splitlines()
    if not lines[0].startswith('#'):
        pytest.skip(f'{os.path.basename(p)} predates the trace-derived step count')
    rows = list(csv.reader(l for l in lines if not l.startswith('#')))
    dom = [r for r in rows if r[0].startswith('conv_igemm256w_kernel')][0]
    assert float(dom[1]) == 29.0
    total = [r for r in rows if r[0] == 'TOTAL kernel time'][0]
    b = _latest('r*_bench.json')
    if b and os.path.basename(b)[:3] == os.path.basename(p)[:3]:
        ms = json.load(open(b))['ms_per_step']
        assert float(total[2]) / 1e3 < 1.03 * ms, 'kernel time per step cannot exceed the step (one stream)'


def test_committed_fp16x3_profiles_recompute_the_roofline_fraction():
    """round 6: the default precision's records -- `rocprofv3 --kernel-trace --stats` of `bench.py --precision fp16x3 --batch 16`
    (12 forward steps) and the PMC traffic file -- must let a reader recompute the fraction the bench line states: 32 launches of
    conv16x3p_kernel per step, 20.05 TFLOP of fp16 MFMA per step at batch 16 (3 products x 417.7 GFLOP of the plane region per tile)"""
    p = _latest('r*_fp16x3_kernel_stats.csv')
    assert p, 'no committed fp16x3 kernel stats'
    rows = list(csv.DictReader(open(p)))
    x3p = [r for r in rows if 'conv16x3p_kernel' in r['Name']]
    stem = [r for r in rows if 'stem7x7_kernel' in r['Name'] or 'stem_pool32_kernel' in r['Name']]      # (fused with its max-pool since finding 67)
    assert x3p and stem
    steps = sum(int(r['Calls']) for r in stem)
    calls = sum(int(r['Calls']) for r in x3p)
    assert calls == 32 * steps, (calls, steps)
    ms_per_step = sum(int(r['TotalDurationNs']) for r in x3p) / steps / 1e6
    line = json.load(open(_latest('r*_bench_fp16x3_profiled.json')))
    flops = line['roofline']['kernel_flops_per_step']
    assert abs(flops - 20.05e12) < 0.02e12, flops
    frac = flops / (ms_per_step * 1e-3) / 2.5e15
    assert 0.45 < frac < 0.75, frac
    assert abs(frac - line['roofline']['frac']) < 0.03, (frac, line['roofline']['frac'])      # the trace and the HIP events agree
    t = json.load(open(_latest('r*_x3p_traffic.json')))
    assert t['dispatches'] % 32 == 0 and t['launches_per_step'] == 32
    assert 1.0 <= t['ratio'] < 3.0 and abs(t['algorithmic_bytes_per_launch'] - 507592704.0) < 1.0
