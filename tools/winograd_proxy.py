"""Upper bound for a Winograd F(2x2,3x3) form of the dilated 3x3 convolutions, measured with the
library's own GEMM tiles -- no Winograd kernel is built; DESIGN.md finding 25 has the arithmetic.

A dilation-d 3x3 conv is d*d interleaved undilated sub-grids, and F(2x2,3x3) turns each 2x2 output tile into 16
element-wise products over channels: 16 independent GEMMs  M[xi] = V[xi] (tiles x Cin) . U[xi] (Cin x Cout)  with
tiles = pixels / 4, i.e. 16/36 of the direct MFMA work.  What the matrix pipe can make of them depends on the tile:

  * FUSED (transforms in LDS, as the verdict asks): a workgroup must hold all 16 products of its output pixels, so its
    65 536 accumulators (the register file of 8 waves) are 16 tiles of 64 x 64 instead of one of 256 x 256 -- 4x the
    operand bytes through LDS-DMA and LDS per MFMA.  Proxy: the 1x1 conv of the same GEMM volume (16 x 32768 rows,
    K = 2048, 512 couts) on the library's 64 x 64 tiles (two-stage and deep-ring) and, as a bound from above, 128 x 128.
  * UNFUSED (V and M through HBM, 256 x 256 tiles): the GEMMs run at the direct kernel's rate, but V is 4x the input
    (2.1 GB written + read per launch) and M is fp32 (1.07 GB written + read): >= 6.4 GB at the ~3.4 TB/s these
    mixed-traffic kernels reach = 1.9 ms on top.

python tools/winograd_proxy.py   (one MI355X)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

lib = _abi.load()
dev = torch.device('cuda:0')
st = _abi.stream_ptr(dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


def conv(B, H, W, Cin, Cout, k, d, variant):
    x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
    w = (torch.randn((Cout, k * k, Cin), device=dev) / np.sqrt(Cin * k * k)).to(torch.float16)
    b = torch.randn((Cout,), device=dev)
    out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
    p = d * (k - 1) // 2

    def run():
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                           _abi.ptr(out), Cout, Cout, k, k, 1, p, d, 1, variant, st), 'conv')
    return timeit(run)


direct = conv(32, 64, 64, 2048, 512, 3, 2, 0)
fl = 2.0 * 32 * 64 * 64 * 512 * 2048 * 9
print(f'direct merged ASPP 3x3 (2048 -> 512, d = 2, 32 x 64^2), 256 x 256 tile: {direct:.3f} ms = {fl / direct / 1e9:.0f} TFLOP/s')
wfl = fl * 16 / 36
for name, variant in (('64 x 64, two-stage', 3 << 4), ('64 x 64, deep ring', 7 << 4), ('128 x 128', 1 << 4), ('256 x 256', 4 << 4)):
    t = conv(128, 64, 64, 2048, 512, 1, 1, variant)      # 16 x 32768 rows: the 16 Winograd GEMMs stacked along M
    print(f'Winograd GEMM volume ({wfl / 1e12:.2f} TFLOP of MFMA work) on {name:20s} tiles: {t:.3f} ms = {wfl / t / 1e9:.0f} TFLOP/s '
          f'-> {"slower" if t > direct else "faster"} than the direct conv BEFORE any transform' )
