"""One launch per main conv shape (auto variant) for `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`, plus the join of the
per-dispatch counters with each shape's algorithmic bytes.
  run  : python tools/conv_traffic.py run [batch]         (under rocprofv3 --pmc ...)
  join : python tools/conv_traffic.py join <fetch_dir> <write_dir> [batch]"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from conv_bench import SHAPES  # noqa: E402


def alg_bytes(B):
    rows = []
    for name, H, W, Cin, Cout, k_, s, p, d in SHAPES:
        Ho = (H + 2 * p - d * (k_ - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k_ - 1) - 1) // s + 1
        rows.append((name, B * H * W * Cin * 2 + Cout * k_ * k_ * Cin * 2, B * Ho * Wo * Cout * 2))
    return rows


def run(B):
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import _abi
    lib = _abi.load()
    dev = torch.device('cuda:0')
    for name, H, W, Cin, Cout, k_, s, p, d in SHAPES:
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
        w = (torch.randn((Cout, k_ * k_, Cin), device=dev) / np.sqrt(Cin * k_ * k_)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        Ho = (H + 2 * p - d * (k_ - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k_ - 1) - 1) // s + 1
        out = torch.empty((B, Ho, Wo, Cout), device=dev, dtype=torch.float16)
        torch.cuda.synchronize()
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                           _abi.ptr(out), Cout, Cout, k_, k_, s, p, d, 1, 0, _abi.stream_ptr(dev)),
                   'conv')
        torch.cuda.synchronize()


def counters(d, name):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if ('conv_igemm' in r['Kernel_Name'] or 'conv3x3_c' in r['Kernel_Name']) and r['Counter_Name'] == name]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    return [float(r['Counter_Value']) for r in rows]


def join(fd, wd, B):
    fe, wr = counters(fd, 'FETCH_SIZE'), counters(wd, 'WRITE_SIZE')
    alg = alg_bytes(B)
    assert len(fe) == len(alg) == len(wr), (len(fe), len(wr), len(alg))
    print(f'{"shape":40s} {"alg read MB":>12s} {"fetched MB":>11s} {"ratio":>6s} {"alg write MB":>13s} {"written MB":>11s}')
    for (name, rb, wb), f, w in zip(alg, fe, wr):
        f = f * 1024 * 2          # KiB units, gfx950 request-size correction (MI355X_MICROARCH.md)
        w = w * 1024
        print(f'{name:40s} {rb/1e6:12.1f} {f/1e6:11.1f} {f/rb:6.2f} {wb/1e6:13.1f} {w/1e6:11.1f}')


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    else:
        join(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 32)
