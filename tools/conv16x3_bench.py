"""The fp16x3 mode's convolution (csrc/conv16x3.hip) and the fp32 mode's (ref32.hip) on the network's main shapes (HIP
events): ms, fp32-equivalent TFLOP/s (x 3 = the fp16 MFMA rate of the split products).  python tools/conv16x3_bench.py [batch]"""
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
    ('fuse pw 320->256', 256, 256, 320, 256, 1, 1, 0, 1),
    ('head pw 256->256', 256, 256, 256, 256, 1, 1, 0, 1),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    lib = _abi.load()
    dev = torch.device('cuda:0')
    tot = {'x3': 0.0, 'f32': 0.0}
    for name, H, W, Cin, Cout, k_, s, p, d in SHAPES:
        if flt not in name:
            continue
        x = torch.relu(torch.randn((B, H, W, Cin), device=dev))
        w = torch.randn((Cout, k_ * k_, Cin), device=dev) / np.sqrt(Cin * k_ * k_)
        b = torch.randn((Cout,), device=dev)
        Ho = (H + 2 * p - d * (k_ - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k_ - 1) - 1) // s + 1
        out = torch.empty((B, Ho, Wo, Cout), device=dev)
        flops = 2.0 * B * Ho * Wo * Cout * Cin * k_ * k_
        st = _abi.stream_ptr(dev)

        def x3():
            _abi.check(lib.emp_conv2d_nhwc_f16x3(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                                 _abi.ptr(out), Cout, Cout, k_, k_, s, p, d, 1, 1, 0, st), 'x3')

        def f32():
            _abi.check(lib.emp_conv2d_nhwc_f32(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                               _abi.ptr(out), Cout, Cout, k_, k_, s, p, d, 1, st), 'f32')
        ts = {}
        for nm, fn in (('x3', x3), ('f32', f32)):
            v = []
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    v.append(e0.elapsed_time(e1) / 3)
            ts[nm] = float(np.median(v))
            tot[nm] += ts[nm]
        gb = (x.numel() + out.numel()) * 4 / 1e9
        print(f"{name:24s} {flops / 1e9:8.1f} GF | x3 {ts['x3']:7.3f} ms {flops / ts['x3'] / 1e9:6.1f} TF-eq ({3 * flops / ts['x3'] / 1e9:6.0f} fp16) "
              f"{gb / ts['x3'] * 1e3:6.0f} GB/s | f32 {ts['f32']:7.3f} ms {flops / ts['f32'] / 1e9:6.1f} TF", flush=True)
    print(f"sum: x3 {tot['x3']:.3f} ms, f32 {tot['f32']:.3f} ms")


if __name__ == '__main__':
    main()
