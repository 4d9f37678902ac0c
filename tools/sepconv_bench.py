"""Fused separable conv vs the unfused dwconv + 1x1 conv (+ head1x1) on the network's shapes (HIP events).
Usage: python tools/sepconv_bench.py [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

SHAPES = [('fuse0 128^2 320->256', 128, 128, 320, 256, 0), ('fuse1 256^2 320->256', 256, 256, 320, 256, 0),
          ('head 256^2 256->256 ->1', 256, 256, 256, 256, 1), ('head 256^2 256->256 ->2', 256, 256, 256, 256, 2)]


def timeit(fn, reps=5, rounds=4):
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lib = _abi.load()
    dev = torch.device('cuda:0')
    st = _abi.stream_ptr(dev)
    for name, H, W, Cc, Cout, hc in SHAPES:
        x = torch.randn((B, H, W, Cc), device=dev).to(torch.float16)
        dw = (torch.randn((25, Cc), device=dev) * 0.2).to(torch.float16)
        pw = (torch.randn((Cout, Cc), device=dev) / np.sqrt(Cc)).to(torch.float16)
        pwp = torch.empty_like(pw)
        _abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pw), Cc, Cc, Cout, _abi.ptr(pwp), st), 'pack')
        b = torch.randn((Cout,), device=dev)
        mid = torch.empty((B, H, W, Cc), device=dev, dtype=torch.float16)
        out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        hw = torch.randn((max(hc, 1), Cout), device=dev)
        hb = torch.randn((max(hc, 1),), device=dev)
        ho = torch.empty((B, max(hc, 1), H, W), device=dev)

        def unfused():
            _abi.check(lib.emp_dwconv_nhwc_f16(_abi.ptr(x), B, H, W, Cc, Cc, _abi.ptr(dw), 5, _abi.ptr(mid), Cc, st), 'dw')
            _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(mid), B, H, W, Cc, Cc, _abi.ptr(pw), _abi.ptr(b), None, None, 0,
                                               _abi.ptr(out), Cout, Cout, 1, 1, 1, 0, 1, 1, 0, st), 'pw')

        def fused():
            _abi.check(lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(x), B, H, W, Cc, Cc, _abi.ptr(dw), _abi.ptr(pwp),
                                                   _abi.ptr(b), Cout, 1, None if hc else _abi.ptr(out), Cout,
                                                   _abi.ptr(hw) if hc else None, _abi.ptr(hb) if hc else None, hc,
                                                   _abi.ptr(ho) if hc else None, st), 'fused')
        # the block with the exact depthwise half (sepconv_precise.hip): fp32 taps, depthwise result as fp16 hi + lo, 2 MFMAs per product
        dw32 = torch.randn((25, Cc), device=dev) * 0.2
        pw32 = torch.randn((Cout, Cc), device=dev) / np.sqrt(Cc)
        dwq, pwq = torch.empty_like(dw32), torch.empty((Cout, Cc), device=dev, dtype=torch.float16)
        _abi.check(lib.emp_sepconvp_pack_dw(_abi.ptr(dw32), 5, Cc, _abi.ptr(dwq), st), 'pack_dw')
        _abi.check(lib.emp_sepconvp_pack_pw(_abi.ptr(pw32), Cc, Cc, Cout, _abi.ptr(pwq), st), 'pack_pw')

        def precise():
            _abi.check(lib.emp_sepconvp_nhwc_f16(_abi.ptr(x), B, H, W, Cc, Cc, 5, _abi.ptr(dwq), _abi.ptr(pwq),
                                                 _abi.ptr(b), Cout, 1, None if hc else _abi.ptr(out), Cout,
                                                 _abi.ptr(hw) if hc else None, _abi.ptr(hb) if hc else None, hc,
                                                 _abi.ptr(ho) if hc else None, st), 'precise')
        tu, tf, tp = timeit(unfused), timeit(fused), timeit(precise)
        gb = (x.numel() + (0 if hc else out.numel())) * 2 / 1e9
        fl = 2.0 * B * H * W * Cc * (25 + Cout) / 1e9
        print(f'{name:26s} unfused(dw+pw) {tu:6.3f} ms | fused {tf:6.3f} ms | precise {tp:6.3f} ms  {gb/tf*1e3:6.0f} GB/s alg  {fl/tf:7.1f} GFLOP/ms',
              flush=True)


if __name__ == '__main__':
    main()
