"""bench.py end to end on the GPU box: the JSON line's contract (roofline, cpu_baseline, parity), and the N > 1 form --
two ranks time-sharing the one GPU over gloo (EMP_BENCH_SHARE_GPU=1: the builder's boxes have one GPU) -- which must
also carry the `stack3d` block of the z-slab job (whatever `--gpus N` command the driver runs on an
8-GPU node records the 3-D scaling too)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=e,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_carries_roofline_cpu_baseline_and_parity():
    j = _bench(['--steps', '2', '--warmup', '1', '--batch', '4', '--size', '512', '--cpu-tiles', '1', '--stack3d', '0',
                '--engine2d', '0', '--latency', '0', '--fp32-mode', '2'])
    assert j['n_gpus'] == 1 and j['unit'] == 'tiles/s' and j['value'] > 0 and j['dtype'] == 'f16'
    f = j['fp32_mode']       # the rate of the tolerance-compliant mode rides on the same line
    assert 'error' not in f, f
    assert f['batch'] == 2 and 0 < f['tiles_per_s'] < j['value'] and 0 < f['frac_of_157TF'] < 1
    x = j['fp16x3_mode']      # ... and the rate of the fast tolerance-compliant mode (item 4)
    assert 'error' not in x, x
    assert x['batch'] == 4 and x['tiles_per_s'] > f['tiles_per_s'] and 0 < x['frac_of_fp16_peak'] < 1
    assert j['forward_steps_bench_loop'] == 1 + 2 * 2 and j['forward_calls_total'] >= j['forward_steps_bench_loop']
    assert j['roofline']['bound'] == 'mfma' and j['cpu_baseline']['kind'] == 'port' and j['cpu_baseline']['value'] > 0
    p = j['parity']
    assert 'error' not in p, p
    # the north star's gate on the float heat-maps, in rms, on the bench's own tile (tests/test_gpu_parity_fullsize.py
    # holds the gates at BASELINE's tile size)
    assert 0 < p['ctr_rms'] < 1e-3 and 0 < p['sem_rms'] < 1e-3, p
    assert p['ctr_max'] < 1e-2 and p['sem_max'] < 1e-2 and 0 <= p['fg_flip_frac'] < 1e-2, p


def test_two_ranks_sharing_the_gpu_emit_the_stack3d_block():
    j = _bench(['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '2', '--size', '256', '--slab-size', '512',
                '--slab-depth', '6', '--engine2d', '0', '--latency', '0', '--no-cpu-baseline'],
               env={'EMP_BENCH_SHARE_GPU': '1'})
    assert j['n_gpus'] == 2 and j['config']['ranks_sharing_one_gpu'] == 2
    s = j['stack3d']
    assert s is not None and 'error' not in s, s
    assert s['unit'] == 'voxels/s' and s['value'] > 0 and s['volume'] == [12, 512, 512] and s['ranks_sharing_one_gpu'] == 2
    assert s['slab_pipeline']['ranks'] == 2 and len(s['slab_pipeline']['per_rank']) == 2
    assert all(r['slices'] == 6 for r in s['slab_pipeline']['per_rank'])


def test_a_stuck_slab_job_costs_the_block_not_the_headline():
    """the z-slab job runs as a child job with a bound (bench.slab_job_child): one that does not finish is killed and
    leaves an `error` entry; the headline line of the N ranks is printed all the same"""
    j = _bench(['--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '2', '--size', '256', '--slab-size', '512',
                '--slab-depth', '6', '--engine2d', '0', '--latency', '0', '--no-cpu-baseline', '--slab-timeout', '1.5'],      # (a warm box finishes the job in under 4 s)
               env={'EMP_BENCH_SHARE_GPU': '1'})
    assert j['n_gpus'] == 2 and j['value'] > 0
    assert 'killed after' in j['stack3d']['error']


def test_profile_tools_read_the_step_count_off_the_trace(tmp_path):
    """the same command tools/refresh_profiles.sh profiles, under rocprofv3 --kernel-trace, at BASELINE's batch: the
    steps tools/step_breakdown.py finds in the trace are the forward calls bench.py says it made, the dominant kernel is
    launched 29 times per step, and the kernel time of a step fits in the step (round 4's committed
    per-step figures were 1.77x high)"""
    import shutil
    if shutil.which('rocprofv3') is None:
        pytest.skip('rocprofv3 not on PATH')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import step_breakdown as sb
    e = dict(os.environ, TMPDIR=str(tmp_path))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    d = str(tmp_path / 'ks')
    r = subprocess.run(['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'ks', '--', sys.executable,
                        os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--stack3d', '0',
                        '--engine2d', '0', '--latency', '0', '--fine-boundaries', '0', '--fp32-mode', '0'],
                       capture_output=True, text=True, env=e, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    steps, rows, total_us = sb.breakdown(d)
    assert steps == j['forward_steps_bench_loop'] == j['forward_calls_total'] == 5
    dom = [(c, us) for k, c, us, _ in rows if k.startswith('conv_igemm256w_kernel')][0]
    assert dom[0] == 29 == j['roofline']['launches_per_step']
    assert total_us / 1e3 < 1.05 * j['ms_per_step'], (total_us, j['ms_per_step'])
    # the HIP-event figure of the line and the trace agree on the dominant kernel's time per step
    assert abs(dom[1] / 1e3 - j['roofline']['kernel_ms_per_step']) < 0.08 * j['roofline']['kernel_ms_per_step']


def test_dominant_kernel_asks_l2_for_whole_lines(tmp_path):
    """Finding 52 as a test (profiles/r05_conv256_requests.txt): on the ASPP 3x3 shape the 256x256 tile's LDS-DMA used to ask
    L2 for half lines -- 64-byte row pieces: ~67 B per TCP->TCC read request -- and, with packed weight images and K-tile
    pairs for the pixels, asks for whole ones: >= 120 B per request for the same bytes.  rocprofv3 --pmc TCP_TCC_READ_REQ_sum
    around tools/one_conv.py; bytes through L1 per launch = workgroups x K-tiles x 32 KiB (every operand byte is staged once
    per workgroup)."""
    import csv
    import glob
    import shutil
    import subprocess
    if shutil.which('rocprofv3') is None:
        pytest.skip('rocprofv3 not on PATH')
    B, H, Cin, Cout, k = 8, 64, 2048, 256, 3
    l1_bytes = (B * H * H // 256) * (Cout // 256) * (k * k * Cin // 32) * 32768

    def requests(name, packed, env):
        d = str(tmp_path / name)
        e = dict(os.environ, TMPDIR=str(tmp_path), **env)
        r = subprocess.run(['rocprofv3', '--pmc', 'TCP_TCC_READ_REQ_sum', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p',
                            '--', sys.executable, os.path.join(ROOT, 'tools', 'one_conv.py'), str(H), str(H), str(Cin), str(Cout),
                            str(k), '4', '64', str(B), str(packed)], capture_output=True, text=True, env=e, cwd=str(tmp_path), timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        assert f, 'no counter file'
        per = {}
        for row in csv.DictReader(open(f[0])):
            if 'conv_igemm256' in row['Kernel_Name'] and row['Counter_Name'] == 'TCP_TCC_READ_REQ_sum':
                per[row['Dispatch_Id']] = per.get(row['Dispatch_Id'], 0.0) + float(row['Counter_Value'])
        assert per, 'the 256x256 tile did not run'
        return l1_bytes / (sum(per.values()) / len(per))

    whole = requests('whole', 1, {})
    half = requests('half', 0, {'EMP_CONV256_WIDE': '0'})
    print(f'bytes per L2 read request: whole-line kernel + packed weights {whole:.1f}, rounds 2-4 kernel {half:.1f}')
    assert whole >= 120.0, whole
    assert half <= 72.0, half
