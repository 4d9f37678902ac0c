"""Round 6: conv16x3p_kernel (256 x 256 tile over hl32 planes, csrc/conv16x3p.hip) against round 5's fp16x3 kernels (fp32 maps,
csrc/conv16x3.hip) on the shapes of the plane region (ResNet layer3 / layer4, ASPP) of a 1024^2 tile, HIP events.
  python tools/x3p_bench.py [batch] [name filter]
columns: GFLOP | new: ms, fp16-MFMA TFLOP/s (3 x the fp32-equivalent rate), workgroups | round 5: ms, TFLOP/s"""
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
    # name, H, W, Cin, Cout, k, stride, pad, dil, res
    ('l3.0.conv2 256 3x3 s2', 128, 128, 256, 256, 3, 2, 1, 1, False),
    ('l3.conv1 1024->256', 64, 64, 1024, 256, 1, 1, 0, 1, False),
    ('l3.conv2 256 3x3', 64, 64, 256, 256, 3, 1, 1, 1, False),
    ('l3.conv3 256->1024 +res', 64, 64, 256, 1024, 1, 1, 0, 1, True),
    ('l4.0.ds 1024->2048', 64, 64, 1024, 2048, 1, 1, 0, 1, False),
    ('l4.conv1 2048->512', 64, 64, 2048, 512, 1, 1, 0, 1, False),
    ('l4.conv2 512 3x3 d2', 64, 64, 512, 512, 3, 1, 2, 2, False),
    ('l4.conv3 512->2048 +res', 64, 64, 512, 2048, 1, 1, 0, 1, True),
    ('aspp 1x1 2048->256', 64, 64, 2048, 256, 1, 1, 0, 1, False),
    ('aspp 3x3 d4 2048->256', 64, 64, 2048, 256, 3, 1, 4, 4, False),
    ('aspp project 1024->256', 64, 64, 1024, 256, 1, 1, 0, 1, False),
]


def timed(fn, reps=3, rounds=4):
    v = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if r:
            v.append(e0.elapsed_time(e1) / reps)
    return float(np.median(v))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    lib = _abi.load()
    dev = torch.device('cuda:0')
    st = _abi.stream_ptr(dev)
    tot = [0.0, 0.0]
    for name, H, W, Cin, Cout, k_, s, p, d, res in SHAPES:
        if flt not in name:
            continue
        x = torch.relu(torch.randn((B, H, W, Cin), device=dev))
        w = torch.randn((Cout, k_ * k_ * Cin), device=dev) / np.sqrt(Cin * k_ * k_)
        b = torch.randn((Cout,), device=dev)
        Ho = (H + 2 * p - d * (k_ - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k_ - 1) - 1) // s + 1
        M = B * Ho * Wo
        K = k_ * k_ * Cin
        r32 = torch.relu(torch.randn((M, Cout), device=dev)) if res else None
        out32 = torch.empty((M, Cout), device=dev)
        xh = torch.empty((B * H * W, 2 * Cin), dtype=torch.float16, device=dev)
        _abi.check(lib.emp_hl32_from_f32(_abi.ptr(x), _abi.ptr(xh), B * H * W, Cin, Cin, Cin, st), 'hl32')
        rh = None
        if res:
            rh = torch.empty((M, 2 * Cout), dtype=torch.float16, device=dev)
            _abi.check(lib.emp_hl32_from_f32(_abi.ptr(r32), _abi.ptr(rh), M, Cout, Cout, Cout, st), 'hl32')
        outh = torch.empty((M, 2 * Cout), dtype=torch.float16, device=dev)
        img = torch.empty((2 * Cout * K,), dtype=torch.float16, device=dev)
        _abi.check(lib.emp_x3p_pack_weights(_abi.ptr(w), _abi.ptr(img), Cout, K, st), 'pack')
        flops = 2.0 * M * Cout * K

        def new():
            _abi.check(lib.emp_conv2d_hl32_f16x3(_abi.ptr(xh), B, H, W, Cin, Cin, _abi.ptr(img), _abi.ptr(b), None, _abi.ptr(rh) if res else None,
                                                 Cout, 1, _abi.ptr(outh), Cout, 1, Cout, k_, k_, s, p, d, 1, st), 'x3p')

        scratch = torch.empty((16 << 20,), device=dev) if os.environ.get('X3P_KSPLIT') else None

        def new_split():
            _abi.check(lib.emp_conv2d_hl32_f16x3_ksplit(_abi.ptr(xh), B, H, W, Cin, Cin, _abi.ptr(img), _abi.ptr(b), None, _abi.ptr(rh) if res else None,
                                                        Cout, 1, _abi.ptr(outh), Cout, 1, None, 0, 0, Cout, k_, k_, s, p, d, 1, _abi.ptr(scratch),
                                                        scratch.numel() * 4, st), 'x3p ksplit')
        if scratch is not None:
            new = new_split      # noqa: F811  (X3P_KSPLIT=1: the split-K entry; the launcher's rule decides whether a shape splits)

        def old():
            _abi.check(lib.emp_conv2d_nhwc_f16x3(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, _abi.ptr(r32) if res else None,
                                                 Cout, _abi.ptr(out32), Cout, Cout, k_, k_, s, p, d, 1, 1, 0, st), 'x3')
        tn, to = timed(new), timed(old)
        tot[0] += tn
        tot[1] += to
        wgs = -(-M // 256) * (Cout // 256)
        print(f"{name:26s} {flops / 1e9:8.1f} GF | new {tn:7.3f} ms {3 * flops / tn / 1e9:6.0f} TF fp16  {wgs:5d} WG | round 5 {to:7.3f} ms "
              f"{3 * flops / to / 1e9:6.0f} TF fp16 | x{to / tn:.2f}", flush=True)
    print(f"sum: new {tot[0]:.3f} ms, round 5 (fp32 weights through the C ABI) {tot[1]:.3f} ms")


if __name__ == '__main__':
    main()
