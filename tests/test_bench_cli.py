"""bench.py's launcher contract on a machine without GPUs: `--gpus N` must never silently run a smaller job."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=e,
                          timeout=300)


def test_gpus_flag_without_enough_gpus_is_refused():
    import torch
    if torch.cuda.device_count() >= 4:
        import pytest
        pytest.skip('enough GPUs here')
    r = _run(['--gpus', '4', '--steps', '1', '--warmup', '0'])
    assert r.returncode == 2 and 'refusing' in r.stderr and '"n_gpus"' not in r.stdout


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '0'], env={'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '2'})
    assert r.returncode == 2 and 'does not match' in r.stderr and '"n_gpus"' not in r.stdout
