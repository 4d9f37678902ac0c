"""conv32 (fp32 reference mode) per shape.  Round 4 ran it as an A/B of the kernel's 64 x 64 tile against a 128 x 128 / 128 x 64
form with b128 fragment reads, four accumulators per wave and register-prefetched global loads (EMP_CONV32_TILE=64 / 128, an
experiment build): profiles/r04_conv32_tiles.txt -- the larger tile was slower on 8 of 10 shapes and was not kept (finding
43).  Against the library as it is, the three columns measure the same kernel.  python tools/conv32_bench.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [
    # name, N, H, W, Cin, Cout, k, stride, dil
    ('stage1 1x1 168->168 b4', 4, 256, 256, 176, 168, 1, 1, 1),
    ('stage2 1x1 392->392 b4', 4, 128, 128, 400, 392, 1, 1, 1),
    ('stage3 1x1 784->784 b4', 4, 64, 64, 784, 784, 1, 1, 1),
    ('stage4 1x1 1624->1624 b4', 4, 32, 32, 1632, 1624, 1, 1, 1),
    ('resnet l1 conv3 64->256 b4', 4, 256, 256, 64, 256, 1, 1, 1),
    ('resnet l3 conv2 3x3 256 b4', 4, 64, 64, 256, 256, 3, 1, 1),
    ('aspp 3x3 2048->256 d6 b4', 4, 64, 64, 2048, 256, 3, 1, 6),
    ('aspp 3x3 2048->256 d6 b1', 1, 64, 64, 2048, 256, 3, 1, 6),
    ('fuse pw 320->256 b4', 4, 256, 256, 320, 256, 1, 1, 1),
    ('resnet l4 conv3 512->2048 b8', 8, 64, 64, 512, 2048, 1, 1, 1),
]


def child():
    import time
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from empanada_napari_amd import _abi
    lib = _abi.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    out = {}
    for name, N, H, W, Cin, Cout, k, stride, dil in SHAPES:
        pad = dil * (k // 2)
        x = torch.randn((N, H, W, Cin), device=dev)
        w = torch.randn((Cout, k * k, Cin), device=dev) * 0.02
        b = torch.zeros((Cout,), device=dev)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        o = torch.empty((N, Ho, Wo, Cout), device=dev)

        def run():
            _abi.check(lib.emp_conv2d_nhwc_f32(_abi.ptr(x), N, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0, _abi.ptr(o), Cout,
                                               Cout, k, k, stride, pad, dil, 1, _abi.stream_ptr(dev)), 'conv32')
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        reps = 10
        t = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        fl = 2.0 * N * Ho * Wo * Cout * Cin * k * k
        out[name] = {'us': round(dt * 1e6, 1), 'TFLOPs': round(fl / dt / 1e12, 1), 'sum': float(o.double().sum())}
    print(json.dumps(out))


if __name__ == '__main__':
    if os.environ.get('CONV32_CHILD'):
        child()
        sys.exit(0)
    res = {}
    for tile in ('64', '128', '0'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CONV32_CHILD='1', EMP_CONV32_TILE=tile),
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tile] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    lines = []
    for name, *_ in SHAPES:
        a, b, c = res['64'][name], res['128'][name], res['0'][name]
        same = a['sum'] == b['sum']
        lines.append(f"{name:34s} 64x64 {a['us']:9.1f} us {a['TFLOPs']:6.1f} TF | 128xBN {b['us']:9.1f} us {b['TFLOPs']:6.1f} TF | auto {c['us']:9.1f} us | bit-identical sums: {same}")
    txt = '\n'.join(lines)
    print(txt)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    open(os.path.join(ROOT, 'gpurun_out', 'conv32_bench.txt'), 'w').write(txt + '\n')
