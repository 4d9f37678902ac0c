"""A/B of the 256x256 conv tile's knobs on the network's MFMA-bound shapes (interleaved rounds in ONE process, HIP
events, random data): K-walk group (32-channel slabs per group) x DMA placement (mode).
    python tools/conv256_ab.py [batch] [filter]"""
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
    ('aspp 3x3 d4 2048->256', 64, 64, 2048, 256, 3, 1, 4, 4),
    ('aspp 3x3 d2 2048->256', 64, 64, 2048, 256, 3, 1, 2, 2),
    ('l4.conv2 512 3x3 d2', 64, 64, 512, 512, 3, 1, 2, 2),
    ('l3.conv2 256 3x3', 64, 64, 256, 256, 3, 1, 1, 1),
    ('l4.conv1 2048->512', 64, 64, 2048, 512, 1, 1, 0, 1),
    ('l3.conv1 1024->256', 64, 64, 1024, 256, 1, 1, 0, 1),
    ('aspp 1x1 2048->256', 64, 64, 2048, 256, 1, 1, 0, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    lib = _abi.load()
    dev = torch.device('cuda:0')
    for name, H, W, Cin, Cout, k_, s, p, d in SHAPES:
        if flt not in name:
            continue
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
        w = (torch.randn((Cout, k_ * k_, Cin), device=dev) / np.sqrt(Cin * k_ * k_)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        flops = 2.0 * B * H * W * Cout * Cin * k_ * k_
        kgs = [8] if k_ > 1 else [0]
        modes = [int(m) for m in os.environ.get('MODES', '0,3').split(',')]
        VARS = [(f'kg{kg} m{m}', 3 + 16 * 4 + 256 * kg + 65536 * m) for kg in kgs for m in modes]
        times = {k: [] for k, _ in VARS}
        ref = {}
        for rnd in range(6):
            for k, v in VARS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4
                e0.record()
                for _ in range(reps):
                    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None,
                                                       None, 0, _abi.ptr(out), Cout, Cout, k_, k_, s, p, d, 1, v,
                                                       _abi.stream_ptr(dev)), 'conv')
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    times[k].append(e0.elapsed_time(e1) / reps)
                kgname = k.split()[0]
                if kgname not in ref:
                    ref[kgname] = out.clone()
                elif not torch.equal(ref[kgname], out):      # every round: a race shows up as a rare wrong tile
                    bad = (ref[kgname] != out).float().mean().item()
                    print(f'!! {name} {k} round {rnd}: differs from mode 0 in {bad:.2e} of the elements', flush=True)
        print(f'{name:24s} {flops/1e9:8.1f} GF | ' +
              ' | '.join(f'{k} {np.median(t)*1e3:6.0f}us {flops/np.median(t)/1e9:5.0f}TF' for k, t in times.items()), flush=True)


if __name__ == '__main__':
    main()
