"""A/B timing of the implicit-GEMM conv variants on the network's main shapes
(interleaved rounds in ONE process, HIP events).  Usage: python tools/conv_bench.py [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

SHAPES = [
    ('exp 64->64 1x1', 256, 256, 64, 64, 1, 1, 0, 1),
    ('exp 64->128 1x1', 256, 256, 64, 128, 1, 1, 0, 1),
    ('exp 64->512 1x1', 256, 256, 64, 512, 1, 1, 0, 1),
    ('exp 128->128 1x1', 256, 256, 128, 128, 1, 1, 0, 1),
  # name, H, W, Cin, Cout, k, stride, pad, dil  (spatial sizes for a 1024^2 tile)
    ('l1.conv1 256->64 1x1', 256, 256, 256, 64, 1, 1, 0, 1),
    ('l1.conv2 64->64 3x3', 256, 256, 64, 64, 3, 1, 1, 1),
    ('l1.conv3 64->256 1x1', 256, 256, 64, 256, 1, 1, 0, 1),
    ('l2.conv2 128 3x3', 128, 128, 128, 128, 3, 1, 1, 1),
    ('l2.conv3 128->512', 128, 128, 128, 512, 1, 1, 0, 1),
    ('l3.conv2 256 3x3', 64, 64, 256, 256, 3, 1, 1, 1),
    ('l3.conv3 256->1024', 64, 64, 256, 1024, 1, 1, 0, 1),
    ('l3.conv1 1024->256', 64, 64, 1024, 256, 1, 1, 0, 1),
    ('l4.conv2 512 3x3 d2', 64, 64, 512, 512, 3, 1, 2, 2),
    ('l4.conv3 512->2048', 64, 64, 512, 2048, 1, 1, 0, 1),
    ('l4.conv1 2048->512', 64, 64, 2048, 512, 1, 1, 0, 1),
    ('aspp 3x3 d4 2048->256', 64, 64, 2048, 256, 3, 1, 4, 4),
    ('aspp 3x3 d4 2048->512 (both decoders)', 64, 64, 2048, 512, 3, 1, 4, 4),
    ('aspp 1x1 2048->256', 64, 64, 2048, 256, 1, 1, 0, 1),
    ('fuse pw 320->256', 256, 256, 320, 256, 1, 1, 0, 1),
    ('head pw 256->256', 256, 256, 256, 256, 1, 1, 0, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lib = _abi.load()
    dev = torch.device('cuda:0')
    rows = []
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    for name, H, W, Cin, Cout, k_, s, p, d in SHAPES:
        if flt not in name:
            continue
        x = (torch.randn((B, H, W, Cin), device=dev) * 1.0).to(torch.float16)
        w = (torch.randn((Cout, k_ * k_, Cin), device=dev) / np.sqrt(Cin * k_ * k_)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        Ho = (H + 2 * p - d * (k_ - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k_ - 1) - 1) // s + 1
        out = torch.empty((B, Ho, Wo, Cout), device=dev, dtype=torch.float16)
        flops = 2.0 * B * Ho * Wo * Cout * Cin * k_ * k_
        VARS = [('128x128', 16 + 3), ('256x256', 64 if Cout % 256 == 0 and Cin * k_ * k_ >= 128 else 16 + 3), ('auto', 0)]
        times = {k: [] for k, _ in VARS}
        for rnd in range(5):
            for k, v in VARS:
                if v == 16 + 3 and Cout <= 32:
                    pass
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3
                e0.record()
                for _ in range(reps):
                    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None,
                                                       None, 0, _abi.ptr(out), Cout, Cout, k_, k_, s, p, d, 1, v,
                                                       _abi.stream_ptr(dev)), 'conv')
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    times[k].append(e0.elapsed_time(e1) / reps)
        gb = (x.numel() + out.numel() + w.numel()) * 2 / 1e9
        med = {k: np.median(v) for k, v in times.items()}
        best = min(med.values())
        print(f'{name:24s} {flops/1e9:8.1f} GF | ' + ' | '.join(f'{k} {t:6.3f} ms {flops/t/1e9:6.1f}' for k, t in med.items())
              + f' | traffic {gb/best*1e3:6.0f} GB/s', flush=True)


if __name__ == '__main__':
    main()
