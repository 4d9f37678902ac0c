"""can the MFMA-bound half of the step (layer3/4, ASPP) and the HBM / VALU-bound half (stem, layer1/2,
separable blocks, heads) run side by side on CU-masked streams (hipExtStreamCreateWithCUMask)?  The library's OWN launches
(C ABI, explicit stream argument) on streams that own CUs [lo, hi) of every XCD (bit i of the mask = CU i // 8 of XCD i % 8,
tools/microbench/cu_mask_probe.hip checks that against HW_REG_XCC_ID; every XCD keeps CUs of both streams):

  (a) the long-K / write-bound launches of the dominant kernel alone on 256 / 224 / 192 / 160 / 128 CUs
  (b) a layer1 block's convs, a separable block (fuse + head), alone on 32 / 64 / 96 / 128 / 256 CUs
  (c) (a) and (b) side by side on complementary masks
  (d) the whole forward: two engines of 16 tiles each on two streams (unmasked and masked) against one engine of 32

python tools/cu_partition.py [--out gpurun_out/cu_partition.txt]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi, synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab  # noqa: E402
from empanada_napari_amd.preprocess import normalize_params  # noqa: E402

NXCD, NCU = 8, 256
_hip = None
LINES = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    LINES.append(s)


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL('libamdhip64.so')
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    return _hip


def masked_stream(lo, hi, dev):
    """torch stream over a HIP stream that owns CUs [lo, hi) of every XCD"""
    words = (C.c_uint32 * (NCU // 32))()
    for i in range(NCU):
        if lo <= i // NXCD < hi:
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip().hipExtStreamCreateWithCUMask(C.byref(s), NCU // 32, words)
    assert rc == 0 and s.value, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(s.value, device=dev)


def timeit(stream, fn, reps=4, rounds=3):
    ts = []
    with torch.cuda.stream(stream):
        for r in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn(stream)
            e1.record(stream)
            stream.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


class Conv:
    def __init__(self, name, B, H, W, Cin, Cout, k, s, p, d, variant, dev, relu_sparse=True):
        self.name, self.variant = name, variant
        x = torch.randn((B, H, W, Cin), device=dev)
        if relu_sparse:
            x = torch.relu(x)          # post-ReLU statistics (the network's maps): fewer toggling bits than random data
        self.x = x.to(torch.float16)
        self.w = (torch.randn((Cout, k * k, Cin), device=dev) / np.sqrt(Cin * k * k)).to(torch.float16)
        self.b = torch.randn((Cout,), device=dev)
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
        self.out = torch.empty((B, Ho, Wo, Cout), device=dev, dtype=torch.float16)
        self.args = (B, H, W, Cin, Cout, k, s, p, d)
        self.flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        self.bytes = (self.x.numel() + self.out.numel() + self.w.numel()) * 2
        self.lib = _abi.load()

    def __call__(self, stream):
        B, H, W, Cin, Cout, k, s, p, d = self.args
        _abi.check(self.lib.emp_conv2d_nhwc_f16(_abi.ptr(self.x), B, H, W, Cin, Cin, _abi.ptr(self.w), _abi.ptr(self.b), None,
                                                None, 0, _abi.ptr(self.out), Cout, Cout, k, k, s, p, d, 1, self.variant,
                                                C.c_void_p(stream.cuda_stream)), 'conv')


class Sep:
    def __init__(self, name, B, H, W, Cc, Cout, hc, dev):
        self.name = name
        lib = self.lib = _abi.load()
        st = _abi.stream_ptr(dev)
        self.x = torch.relu(torch.randn((B, H, W, Cc), device=dev)).to(torch.float16)
        self.dw = (torch.randn((25, Cc), device=dev) * 0.2).to(torch.float16)
        pw = (torch.randn((Cout, Cc), device=dev) / np.sqrt(Cc)).to(torch.float16)
        self.pwp = torch.empty_like(pw)
        _abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pw), Cc, Cc, Cout, _abi.ptr(self.pwp), st), 'pack')
        self.b = torch.randn((Cout,), device=dev)
        self.out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
        self.hw = torch.randn((max(hc, 1), Cout), device=dev)
        self.hb = torch.randn((max(hc, 1),), device=dev)
        self.ho = torch.empty((B, max(hc, 1), H, W), device=dev)
        self.args = (B, H, W, Cc, Cout, hc)
        self.flops = 2.0 * B * H * W * Cc * (25 + Cout)
        self.bytes = (self.x.numel() + (0 if hc else self.out.numel())) * 2
        torch.cuda.synchronize()

    def __call__(self, stream):
        B, H, W, Cc, Cout, hc = self.args
        _abi.check(self.lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(self.x), B, H, W, Cc, Cc, _abi.ptr(self.dw), _abi.ptr(self.pwp),
                                                    _abi.ptr(self.b), Cout, 1, None if hc else _abi.ptr(self.out), Cout,
                                                    _abi.ptr(self.hw) if hc else None, _abi.ptr(self.hb) if hc else None, hc,
                                                    _abi.ptr(self.ho) if hc else None, C.c_void_p(stream.cuda_stream)), 'sep')


def side_by_side(sa, fa, sb, fb, ta, tb, budget_ms=40.0):
    """both streams kept busy for about the same wall time; -> (ms per launch of a, of b) while the other runs"""
    ra, rb = max(2, int(budget_ms / ta)), max(2, int(budget_ms / tb))
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ea[0].record(sa)
    eb[0].record(sb)
    ia = ib = 0
    while ia < ra or ib < rb:          # interleaved enqueue so that neither queue runs dry
        if ia < ra:
            fa(sa)
            ia += 1
        for _ in range(max(1, rb // ra)):
            if ib < rb:
                fb(sb)
                ib += 1
    ea[1].record(sa)
    eb[1].record(sb)
    torch.cuda.synchronize()
    return ea[0].elapsed_time(ea[1]) / ra, eb[0].elapsed_time(eb[1]) / rb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'cu_partition.txt'))
    ap.add_argument('--skip-model', action='store_true')
    ap.add_argument('--only-model', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    B = 32
    say(f'# CU-partition probe, batch {B} x 1024^2 shapes, {torch.cuda.get_device_name(0)}')
    if a.only_model:
        full = masked_stream(0, 32, dev)
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        return whole_forward(a, dev, full, s1, s2, {}, {})
    mfma = [Conv('merged ASPP 3x3 d4 2048->512', B, 64, 64, 2048, 512, 3, 1, 4, 4, 64, dev),
            Conv('layer4 conv2 3x3 d2 512->512', B, 64, 64, 512, 512, 3, 1, 2, 2, 64, dev),
            Conv('layer4.x.conv3 512->2048 (write-bound)', B, 64, 64, 512, 2048, 1, 1, 0, 1, 64, dev),
            Conv('layer3.x.conv1 1024->256', B, 64, 64, 1024, 256, 1, 1, 0, 1, 64, dev)]
    hbm = [Conv('layer1 conv1 256->64 1x1', B, 256, 256, 256, 64, 1, 1, 0, 1, 0, dev),
           Conv('layer1 conv2 64->64 3x3', B, 256, 256, 64, 64, 3, 1, 1, 1, 0, dev),
           Conv('layer1 conv3 64->256 1x1', B, 256, 256, 64, 256, 1, 1, 0, 1, 0, dev),
           Conv('layer2 conv3 128->512 1x1', B, 128, 128, 128, 512, 1, 1, 0, 1, 0, dev),
           Sep('sepconv5 fuse 320->256 @256^2', B, 256, 256, 320, 256, 0, dev),
           Sep('sepconv5 head 256->256->1 @256^2', B, 256, 256, 256, 256, 1, dev)]
    full = masked_stream(0, 32, dev)
    base = {}
    say('\n(a) MFMA-side launches alone on CUs [lo, 32) of every XCD: ms (TFLOP/s)')
    cuts_a = [0, 4, 8, 12, 16]
    say(f"{'launch':42s} " + ' '.join(f'{(32 - lo) * 8:>14d}' for lo in cuts_a))
    sa = {lo: masked_stream(lo, 32, dev) for lo in cuts_a}
    for k in mfma:
        row = []
        for lo in cuts_a:
            t = timeit(sa[lo], k)
            base[(k.name, 'a', lo)] = t
            row.append(f'{t:6.3f} ({k.flops / t / 1e9:5.0f})')
        say(f'{k.name:42s} ' + ' '.join(f'{r:>14s}' for r in row))
    say('\n(b) HBM / VALU-side launches alone on CUs [0, hi) of every XCD: ms (algorithmic GB/s)')
    cuts_b = [4, 8, 12, 16, 32]
    say(f"{'launch':42s} " + ' '.join(f'{hi * 8:>14d}' for hi in cuts_b))
    sb = {hi: masked_stream(0, hi, dev) for hi in cuts_b}
    for k in hbm:
        row = []
        for hi in cuts_b:
            t = timeit(sb[hi], k)
            base[(k.name, 'b', hi)] = t
            row.append(f'{t:6.3f} ({k.bytes / t / 1e6:5.0f})')
        say(f'{k.name:42s} ' + ' '.join(f'{r:>14s}' for r in row))
    say('\n(c) side by side on complementary masks: ms per launch while the other stream runs (alone on the same mask; alone on 256 CUs)')
    say('    serial = t_a(256) + t_b(256) for one launch of each; pair = what the same two launches cost side by side = max over the'
        ' streams of (its ms per launch), when both are kept busy')
    for ka in (mfma[0], mfma[2]):
        for kb in (hbm[2], hbm[4], hbm[5]):
            for c in (4, 8, 12, 16):
                ta, tb = base[(ka.name, 'a', c)], base[(kb.name, 'b', c)]
                pa, pb = side_by_side(sa[c], ka, sb[c], kb, ta, tb)
                ta0, tb0 = base[(ka.name, 'a', 0)], base[(kb.name, 'b', 32)]
                # work-conserving figure of merit: time to do one unit of each = with the partition, the streams advance at
                # 1/pa and 1/pb units per ms; a step needs n_a units of a and n_b of b in the ratio of their serial times
                say(f'  {ka.name[:28]:28s} on {(32 - c) * 8:3d} || {kb.name[:30]:30s} on {c * 8:3d}: a {pa:6.3f} ({ta:6.3f}; {ta0:6.3f})  '
                    f'b {pb:6.3f} ({tb:6.3f}; {tb0:6.3f})  slowdown a x{pa / ta0:4.2f} b x{pb / tb0:4.2f}  '
                    f'-> 1/(1/x_a + 1/x_b) = {1.0 / (ta0 / pa + tb0 / pb):4.2f} of serial time for a balanced mix')
    # unmasked pair: two ordinary streams
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    for ka in (mfma[0], mfma[2]):
        for kb in (hbm[2], hbm[4]):
            ta0, tb0 = base[(ka.name, 'a', 0)], base[(kb.name, 'b', 32)]
            pa, pb = side_by_side(s1, ka, s2, kb, ta0, tb0)
            say(f'  {ka.name[:28]:28s} unmasked || {kb.name[:30]:30s} unmasked: a {pa:6.3f} ({ta0:6.3f})  b {pb:6.3f} ({tb0:6.3f})  '
                f'-> {1.0 / (ta0 / pa + tb0 / pb):4.2f} of serial time')
    if a.skip_model:
        return finish(a)
    del mfma, hbm
    whole_forward(a, dev, full, s1, s2, sa, sb)


def whole_forward(a, dev, full, s1, s2, sa, sb):
    # ---- (d) the whole forward ----
    say('\n(d) whole forward (PanopticDeepLabPR / resnet50, 1024^2 uint8 tiles): one engine of 32 vs two engines of 16 on two streams')
    torch.cuda.empty_cache()
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    tiles = torch.from_numpy(synth.em_tiles(32, 1024, seed=1234))[:, None].to(dev)
    m32 = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    m32.reserve(32, 1024, 1024)

    def run(pairs, steps):
        """pairs: [(model, stream, tiles)]; one 'step' = every pair's forward once, enqueued round-robin"""
        outs = [[torch.empty((t.shape[0], 1, 1024, 1024), dtype=torch.float32, device=dev),
                 torch.empty((t.shape[0], 1, 256, 256), dtype=torch.float32, device=dev),
                 torch.empty((t.shape[0], 2, 256, 256), dtype=torch.float32, device=dev)] for _, _, t in pairs]
        for w in range(2):
            for (m, s, t), o in zip(pairs, outs):
                with torch.cuda.stream(s):
                    m(t, 2, interpolate_ins=False, sub=float(sub), mul=float(mul), out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for (m, s, t), o in zip(pairs, outs):
                with torch.cuda.stream(s):
                    m(t, 2, interpolate_ins=False, sub=float(sub), mul=float(mul), out=o)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps, outs

    ms32, o32 = run([(m32, full, tiles)], 8)
    say(f'  one engine, batch 32, one stream (all CUs):                 {ms32:7.3f} ms per 32 tiles = {32e3 / ms32:7.1f} tiles/s (forward only)')
    ref = [t.clone() for t in o32[0]]
    del m32, o32
    torch.cuda.empty_cache()
    ma = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    mb = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    ma.reserve(16, 1024, 1024)
    mb.reserve(16, 1024, 1024)
    ta_, tb_ = tiles[:16], tiles[16:]
    ms, o = run([(ma, full, ta_), (mb, full, tb_)], 8)
    same = all(torch.equal(torch.cat([o[0][i], o[1][i]]), ref[i]) for i in range(3))
    say(f'  two engines of 16, ONE stream (serial):                      {ms:7.3f} ms per 32 tiles = {32e3 / ms:7.1f} tiles/s   outputs == batch-32 run: {same}')
    ms, o = run([(ma, s1, ta_), (mb, s2, tb_)], 8)
    same = all(torch.equal(torch.cat([o[0][i], o[1][i]]), ref[i]) for i in range(3))
    say(f'  two engines of 16, two unmasked streams:                     {ms:7.3f} ms per 32 tiles = {32e3 / ms:7.1f} tiles/s   outputs == batch-32 run: {same}')
    for c in (16, 12, 8):
        if c not in sa:
            continue
        ms, o = run([(ma, sa[c], ta_), (mb, sb[c], tb_)], 8)
        same = all(torch.equal(torch.cat([o[0][i], o[1][i]]), ref[i]) for i in range(3))
        say(f'  two engines of 16, masked streams {(32 - c) * 8:3d} + {c * 8:3d} CUs (whole forward each): {ms:7.3f} ms per 32 tiles = {32e3 / ms:7.1f} tiles/s   outputs == batch-32 run: {same}')
    # (e) the same two engines on two unmasked streams with a CONTROLLED offset: the second stream starts d ms after the
    # first and both then run their forwards back to back; which kernels meet depends on d (encoder of one half-batch next
    # to the decoder of the other, ...).  A flat curve = no schedule of whole half-batch forwards beats the average above.
    say('\n(e) two engines of 16 on two unmasked streams, the second delayed by d ms (spin kernel), 6 forwards each:')
    outs = [[torch.empty((16, 1, 1024, 1024), dtype=torch.float32, device=dev),
             torch.empty((16, 1, 256, 256), dtype=torch.float32, device=dev),
             torch.empty((16, 2, 256, 256), dtype=torch.float32, device=dev)] for _ in range(2)]
    clk = 100e6      # torch.cuda._sleep counts cycles of a ~100 MHz-class counter on ROCm? calibrate instead
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    per_ms = 10_000_000 / e0.elapsed_time(e1)
    for d in (0.0, 1.5, 3.0, 4.5, 6.0, 7.5, 9.0, 10.5, 12.0):
        n = 6
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s2):
            if d > 0:
                torch.cuda._sleep(int(d * per_ms))
        for _ in range(n):
            with torch.cuda.stream(s1):
                ma(ta_, 2, interpolate_ins=False, sub=float(sub), mul=float(mul), out=outs[0])
            with torch.cuda.stream(s2):
                mb(tb_, 2, interpolate_ins=False, sub=float(sub), mul=float(mul), out=outs[1])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        say(f'  d = {d:5.1f} ms: {ms:7.2f} ms for {n} x 32 tiles -> {(ms - d) / n:7.3f} ms per 32 tiles net of the delay ({32e3 * n / (ms - d):7.1f} tiles/s)')
    # (f) four engines of 8 tiles on four unmasked streams: more launches in flight, smaller tails
    del ma, mb
    torch.cuda.empty_cache()
    ms4 = [HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16')) for _ in range(4)]
    for m in ms4:
        m.reserve(8, 1024, 1024)
    st4 = [torch.cuda.Stream(dev) for _ in range(4)]
    ms, o = run([(m, full, tiles[8 * i:8 * i + 8]) for i, m in enumerate(ms4)], 8)
    say(f'\n(f) four engines of 8, ONE stream (serial):                    {ms:7.3f} ms per 32 tiles = {32e3 / ms:7.1f} tiles/s')
    ms, o = run([(m, st4[i], tiles[8 * i:8 * i + 8]) for i, m in enumerate(ms4)], 8)
    same = all(torch.equal(torch.cat([o[k][i] for k in range(4)]), ref[i]) for i in range(3))
    say(f'    four engines of 8, four unmasked streams:                  {ms:7.3f} ms per 32 tiles = {32e3 / ms:7.1f} tiles/s   outputs == batch-32 run: {same}')
    finish(a)


def finish(a):
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, 'w').write('\n'.join(LINES) + '\n')


if __name__ == '__main__':
    main()
