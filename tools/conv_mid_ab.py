"""A/B of conv tile variants on the network's memory-bound / short-K shapes (interleaved rounds in ONE process, HIP
events, random data, with and without the residual operand):
    python tools/conv_mid_ab.py [batch] [filter] [variants, comma separated tile codes e.g. 1,4,5]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

# name, H, W (per image, batch 32 -> M), Cin, Cout, residual
SHAPES = [
    ('l4.conv3 512->2048 +res', 64, 64, 512, 2048, True),
    ('l4.conv3 512->2048', 64, 64, 512, 2048, False),
    ('l3.conv3 256->1024 +res', 64, 64, 256, 1024, True),
    ('l2.conv3 128->512 +res', 128, 128, 128, 512, True),
    ('l1.conv3 64->256 +res', 256, 256, 64, 256, True),
    ('l3.0.conv1 512->256', 128, 128, 512, 256, False),
    ('l4.0.conv1 1024->512', 64, 64, 1024, 512, False),
    ('l2.conv1 512->128', 128, 128, 512, 128, False),
    ('l2.0.conv1 256->128', 256, 256, 256, 128, False),
    ('l1.conv1 256->64', 256, 256, 256, 64, False),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    tiles = [int(t) for t in (sys.argv[3] if len(sys.argv) > 3 else '1,4').split(',')]
    lib = _abi.load()
    dev = torch.device('cuda:0')
    for name, H, W, Cin, Cout, res in SHAPES:
        if flt not in name:
            continue
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
        w = (torch.randn((Cout, 1, Cin), device=dev) / np.sqrt(Cin)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        r = torch.randn((B, H, W, Cout), device=dev).to(torch.float16) if res else None
        out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        flops = 2.0 * B * H * W * Cout * Cin
        nbytes = 2.0 * B * H * W * (Cin + Cout * (2 if res else 1))
        times, ref = {}, None
        for rnd in range(6):
            for t in tiles:
                if (t == 4 and Cout % 256) or (t == 5 and Cout % 128):
                    continue
                v = 3 + 16 * t
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 4
                e0.record()
                for _ in range(reps):
                  try:
                    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None,
                                                       _abi.ptr(r) if res else None, Cout if res else 0, _abi.ptr(out), Cout,
                                                       Cout, 1, 1, 1, 0, 1, 1, v, _abi.stream_ptr(dev)), 'conv')
                  except Exception as e:
                    bad = True
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    times.setdefault(t, []).append(e0.elapsed_time(e1) / reps)
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    print(f'!! {name} tile {t} round {rnd}: differs in {(ref != out).float().mean().item():.2e} of the elements',
                          flush=True)
        print(f'{name:26s} {flops/1e9:7.1f} GF {nbytes/1e6:7.0f} MB | ' +
              ' | '.join(f't{t} {np.median(v)*1e3:5.0f}us {flops/np.median(v)/1e9:5.0f}TF {nbytes/np.median(v)/1e6:5.0f}GB/s'
                         for t, v in times.items()), flush=True)


if __name__ == '__main__':
    main()
