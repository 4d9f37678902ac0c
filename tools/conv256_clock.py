"""In-kernel clock of the dominant kernel (conv_igemm256_kernel) on its ASPP 3x3 launch: a DIAGNOSTIC build of the library
(-DEMP_CLOCK_STAMP: wave 0 of every workgroup stamps s_memtime and s_memrealtime around its K loop; the product library
executes no stamp) is launched back to back on random data for >= 2 s, then the quotient
    clock = d(s_memtime) / d(s_memrealtime) x 100 MHz        (median over the workgroups of the last launch)
is printed next to the launch's TFLOP/s (MI355X_MICROARCH.md, DVFS give-back item 6).

    python tools/conv256_clock.py build        # here or on the GPU box: lib/diag/libempanada_hip_clock.so
    python tools/conv256_clock.py run [batch]  # on the GPU box (gpurun)"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'empanada-napari_amd')
DIAG = os.path.join(PKG, 'lib', 'diag', 'libempanada_hip_clock.so')


def build():
    sys.path.insert(0, PKG)
    import build as b
    os.makedirs(os.path.dirname(DIAG), exist_ok=True)
    b.build_all()          # the product objects (every file but the stamped one is linked as it is)
    hipcc = b._hipcc()
    obj = os.path.join(os.path.dirname(DIAG), 'conv_igemm256_clock.o')
    subprocess.check_call([hipcc] + b.COMMON + ['-DEMP_CLOCK_STAMP', '-c', os.path.join(b.CSRC, 'conv_igemm256.hip'), '-o', obj])
    objs = [obj if s == 'conv_igemm256.hip' else os.path.join(b.OBJ, s.replace('.hip', '.o')) for s in b.SOURCES]
    subprocess.check_call([hipcc, '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', DIAG] + objs)
    print(DIAG)


def run(B):
    import torch
    lib = C.CDLL(DIAG)
    dev = torch.device('cuda:0')
    vp = C.c_void_p
    lib.emp_conv2d_nhwc_f16.restype = C.c_int
    lib.emp_conv2d_nhwc_f16.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.emp_diag_clock_stamps.restype = C.c_int
    lib.emp_diag_clock_stamps.argtypes = [vp, C.c_int]
    ptr = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for name, H, W, Cin, Cout, k, pad, dil in (('aspp 3x3 d4 2048->512 (merged decoders)', 64, 64, 2048, 512, 3, 4, 4),
                                               ('layer4 conv2 512->512 3x3 d2', 64, 64, 512, 512, 3, 2, 2),
                                               ('layer4 conv1 2048->512 1x1', 64, 64, 2048, 512, 1, 0, 1)):
        x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
        w = (torch.randn((Cout, k * k, Cin), device=dev) / np.sqrt(Cin * k * k)).to(torch.float16)
        b = torch.randn((Cout,), device=dev)
        out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        flops = 2.0 * B * H * W * Cout * Cin * k * k
        n_wg = (B * H * W // 256) * (Cout // 256)

        def launch():
            rc = lib.emp_conv2d_nhwc_f16(ptr(x), B, H, W, Cin, Cin, ptr(w), ptr(b), None, None, 0, ptr(out), Cout, Cout, k, k, 1,
                                         pad, dil, 1, 3 + 16 * 4, stream)
            assert rc == 0, rc
        launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.2:       # >= 2 s of back-to-back launches before the launch that is read
            for _ in range(20):
                launch()
            n += 20
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        st = np.zeros((min(n_wg, 8192), 2), dtype=np.uint64)
        assert lib.emp_diag_clock_stamps(st.ctypes.data_as(vp), len(st)) == 0
        ok = st[:, 1] > 0
        ghz = st[ok, 0].astype(np.float64) / st[ok, 1].astype(np.float64) * 0.1
        loop_us = st[ok, 1].astype(np.float64) / 100.0
        print(f'{name:42s} batch {B}: {ms * 1e3:8.1f} us per launch = {flops / ms / 1e9:7.1f} TFLOP/s; in-kernel clock median '
              f'{np.median(ghz):.3f} GHz (p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}; {ok.sum()} workgroups, '
              f'K loop median {np.median(loop_us):.1f} us)', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32)
