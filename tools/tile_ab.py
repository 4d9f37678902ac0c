"""A/B of the generic conv kernel's tile shapes (128x128 at 2 workgroups per CU, 128x64 at 3, 64x64 at 5) on the write- /
latency-bound 1x1 layers that stay on it (with their residual reads): python tools/tile_ab.py [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

SHAPES = [   # name, H, W, Cin, Cout, residual
    ('layer1.2.conv3 64->256 +res', 256, 256, 64, 256, True),
    ('layer2.x.conv3 128->512 +res', 128, 128, 128, 512, True),
    ('layer3.x.conv3 256->1024 +res', 64, 64, 256, 1024, True),
    ('layer2.0.conv1+project 256->176', 256, 256, 256, 176, False),
    ('layer2.x.conv1 512->128', 128, 128, 512, 128, False),
    ('layer3.x.conv1 1024->256', 64, 64, 1024, 256, False),
]
TILES = [('auto', 0), ('128x128', 16 * 1 + 3), ('128x64', 16 * 2 + 3), ('64x64', 16 * 3 + 3), ('h256', 16 * 5 + 3)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = _abi.load()
    dev = torch.device('cuda:0')
    for name, H, W, Cin, Cout, res in SHAPES:
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
        w = (torch.randn((Cout, 1, Cin), device=dev) / np.sqrt(Cin)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        r = torch.randn((B, H, W, Cout), device=dev).to(torch.float16) if res else None
        out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        line = f'{name:34s}'
        ref = None
        for tname, v in TILES:
            if tname == 'h256' and Cout % 128:
                continue
            ts = []
            ok = True
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    rc = lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None,
                                                 _abi.ptr(r) if res else None, Cout if res else 0, _abi.ptr(out), Cout, Cout, 1, 1, 1,
                                                 0, 1, 1, v, _abi.stream_ptr(dev))
                    if rc != 0:
                        ok = False
                        break
                e1.record()
                torch.cuda.synchronize()
                if not ok:
                    break
                if rnd:
                    ts.append(e0.elapsed_time(e1) / 5)
            if not ok:
                line += f' | {tname} n/a'
                continue
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(ref, out))
            line += f' | {tname} {1e3 * min(ts):6.1f} us{"" if same else " (differs)"}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
