"""Race screens for the kernels whose pipelines keep LDS-DMA in flight across barriers (counted vmcnt waits, raw
s_barrier): at full-chip sizes, many repetitions must give bit-identical outputs, and those must agree with a
differently structured kernel computing the same op.  A schedule-dependent hazard shows up as rare wrong tiles
that come and go between runs (cdna_hip_programming.md, 8-phase template notes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv256_repeatable_and_equal_to_128_tile():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cin, Cout, k, d = 8, 64, 64, 512, 512, 3, 2            # 128 pixel tiles x 2 cout tiles, 144 K-tiles
    g = torch.Generator().manual_seed(0)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.float16).to(dev())
    w = (torch.randn((Cout, k * k, Cin), generator=g) / np.sqrt(Cin * k * k)).to(torch.float16).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    res = torch.randn((N, H, W, Cout), generator=g).to(torch.float16).to(dev())

    def run(variant):
        out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), N, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, _abi.ptr(res),
                                           Cout, _abi.ptr(out), Cout, Cout, k, k, 1, d, d, 1, variant,
                                           _abi.stream_ptr(dev())), 'conv')
        return out

    first = run(64)
    for _ in range(20):
        assert torch.equal(run(64), first)
    # the 128x128 kernel walks K in the same order (256-channel groups, tap-major, ascending 32-channel MFMA steps) and
    # applies the same fp32 epilogue: the choice of tile -- which depends on the batch size -- does not change a bit
    assert torch.equal(first, run(16 + 3))


def test_sepconv_repeatable_and_equal_to_unfused():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, Cout = 8, 256, 256, 320, 256                         # 16 tiles per workgroup
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, H, W, Cc), generator=g).to(torch.float16).to(dev())
    dw = (torch.randn((25, Cc), generator=g) * 0.2).to(torch.float16).to(dev())
    pw = (torch.randn((Cout, Cc), generator=g) / np.sqrt(Cc)).to(torch.float16).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    pwp = torch.empty_like(pw)
    st = _abi.stream_ptr(dev())
    _abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pw), Cc, Cc, Cout, _abi.ptr(pwp), st), 'pack')

    def fused():
        out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(x), N, H, W, Cc, Cc, _abi.ptr(dw), _abi.ptr(pwp), _abi.ptr(b), Cout,
                                               1, _abi.ptr(out), Cout, None, None, 0, None, st), 'sepconv')
        return out

    mid = torch.empty((N, H, W, Cc), dtype=torch.float16, device=dev())
    ref = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_dwconv_nhwc_f16(_abi.ptr(x), N, H, W, Cc, Cc, _abi.ptr(dw), 5, _abi.ptr(mid), Cc, st), 'dw')
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(mid), N, H, W, Cc, Cc, _abi.ptr(pw), _abi.ptr(b), None, None, 0,
                                       _abi.ptr(ref), Cout, Cout, 1, 1, 1, 0, 1, 1, 0, st), 'pw')
    for _ in range(20):
        assert torch.equal(fused(), ref)


def test_sepconv_precise_repeatable():
    """the block with the exact depthwise half (sepconv_precise.hip: halo DMA from the depthwise waves, compiler-tracked tap / weight loads,
    raw barriers) at 16 tiles per workgroup: 20 launches bit-identical, image 3 equal to the fp64 reference (race screen)"""
    from gpu_common import dev
    from empanada_napari_amd import _abi
    import torch.nn.functional as F
    lib = _abi.load()
    N, H, W, Cc, Cout = 8, 256, 256, 320, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, H, W, Cc), generator=g).to(torch.float16).to(dev())
    dw = (torch.randn((25, Cc), generator=g) * 0.2).to(dev())
    pw = (torch.randn((Cout, Cc), generator=g) / np.sqrt(Cc)).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    pwp = torch.empty((Cout, Cc), dtype=torch.float16, device=dev())
    dwp = torch.empty_like(dw)
    st = _abi.stream_ptr(dev())
    _abi.check(lib.emp_sepconvp_pack_pw(_abi.ptr(pw), Cc, Cc, Cout, _abi.ptr(pwp), st), 'pack_pw')
    _abi.check(lib.emp_sepconvp_pack_dw(_abi.ptr(dw), 5, Cc, _abi.ptr(dwp), st), 'pack_dw')

    def fused():
        out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconvp_nhwc_f16(_abi.ptr(x), N, H, W, Cc, Cc, 5, _abi.ptr(dwp), _abi.ptr(pwp), _abi.ptr(b), Cout,
                                             1, _abi.ptr(out), Cout, None, None, 0, None, st), 'sepconvp')
        return out

    first = fused()
    for _ in range(20):
        assert torch.equal(fused(), first)
    xin = x[3:4].double().cpu().permute(0, 3, 1, 2)
    d = F.conv2d(xin, dw.double().cpu().t().reshape(Cc, 1, 5, 5), padding=2, groups=Cc)
    ref = torch.relu(F.conv2d(d, pw.to(torch.float16).double().cpu()[:, :, None, None], b.double().cpu()))
    err = (first[3:4].double().cpu().permute(0, 3, 1, 2) - ref).abs()
    tol = 2.0 ** (torch.floor(torch.log2(ref.abs().clamp_min(2.0 ** -14))) - 11) + 2e-6 * float(ref.abs().max())
    assert bool((err <= tol).all()), float(err.max())


def test_stem_pool_repeatable():
    from gpu_common import dev
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    x = torch.from_numpy(synth.em_tiles(4, 512, seed=9))[:, None].to(dev())
    model(x, 2, False, sub=146.8, mul=0.0307)
    first = model.tap('p1').clone()
    for _ in range(5):
        model(x, 2, False, sub=146.8, mul=0.0307)
        assert torch.equal(model.tap('p1'), first)


def test_conv3x3_c64_repeatable_and_equal_to_implicit_gemm():
    """the register-weight 3x3 kernel at the bench's layer1 size: ~16 tiles per persistent workgroup through the ring of
    three halo slots; 20 repetitions bit-identical and equal to the implicit-GEMM kernel (same K order).  [A first
    version let the 24th, dummy LDS-DMA of a tile land zeros on top of piece 22 -- caught by exactly this kind of run.]"""
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W = 8, 256, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, H, W, 64), generator=g).to(torch.float16).to(dev())
    w = (torch.randn((64, 9, 64), generator=g) / 24.0).to(torch.float16).to(dev())
    b = torch.randn((64,), generator=g).to(dev())

    def run(variant):
        out = torch.empty((N, H, W, 64), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), N, H, W, 64, 64, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                           _abi.ptr(out), 64, 64, 3, 3, 1, 1, 1, 1, variant, _abi.stream_ptr(dev())), 'conv')
        return out

    first = run(96)
    for _ in range(20):
        assert torch.equal(run(96), first)
    assert torch.equal(first, run(32 + 3))          # the 128x64 implicit-GEMM tile
    assert torch.equal(first, run(0))               # auto picks the register-weight kernel at this size
    # the 128-channel variant (layer2 conv2): 16 tiles per workgroup, output transposed through the consumed halo slot
    N2, H2 = 16, 128
    x2 = torch.randn((N2, H2, H2, 128), generator=g).to(torch.float16).to(dev())
    w2 = (torch.randn((128, 9, 128), generator=g) / 34.0).to(torch.float16).to(dev())
    b2 = torch.randn((128,), generator=g).to(dev())

    def run2(variant):
        out = torch.empty((N2, H2, H2, 128), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x2), N2, H2, H2, 128, 128, _abi.ptr(w2), _abi.ptr(b2), None, None, 0,
                                           _abi.ptr(out), 128, 128, 3, 3, 1, 1, 1, 1, variant, _abi.stream_ptr(dev())), 'conv')
        return out

    first2 = run2(96)
    for _ in range(20):
        assert torch.equal(run2(96), first2)
    assert torch.equal(first2, run2(16 + 3))


@pytest.mark.parametrize('shape', [
    # N, H, W, Cin, Cout, k, dil, res        (full-chip sizes: every CU holds one or two workgroups)
    (8, 64, 64, 512, 2048, 1, 1, True),      # layer4 conv3: short K, wide N, residual
    (8, 128, 128, 512, 128, 1, 1, False),    # layer2 conv1: the 256 x 128 half tile
    (8, 64, 64, 256, 256, 3, 1, False),      # 72 K-tiles, tap walk
    (1, 64, 64, 2048, 256, 3, 4, False),     # ASPP 3x3 at batch 1: 576 K-tiles through the deep ring
])
def test_half_tile_and_deep_ring_tiles_repeatable_and_equal_to_128_tile(shape):
    """conv_igemm_h256_kernel (tile code 5) and conv_igemm_s64_kernel (7) keep LDS-DMA in flight across raw barriers with
    counted vmcnt waits, like the 256x256 kernel: repeated runs must agree bit for bit with each other and with the
    two-stage 128x128 tile."""
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cin, Cout, k, d, use_res = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.float16).to(dev())
    w = (torch.randn((Cout, k * k, Cin), generator=g) / np.sqrt(Cin * k * k)).to(torch.float16).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    res = torch.randn((N, H, W, Cout), generator=g).to(torch.float16).to(dev()) if use_res else None
    pad = d * (k - 1) // 2

    def run(variant):
        out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), N, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None,
                                           _abi.ptr(res) if use_res else None, Cout if use_res else 0, _abi.ptr(out), Cout,
                                           Cout, k, k, 1, pad, d, 1, variant, _abi.stream_ptr(dev())), 'conv')
        return out

    ref = run(16 + 3)
    for tile in (5, 7):
        for rep in range(10):
            assert torch.equal(run(16 * tile + 3), ref), f'tile code {tile}, repetition {rep}'


def test_back_to_back_fusion_repeatable_at_batch_32():
    """The fused conv3 -> next conv1 launch (conv_igemm256_kernel<0, true>) reuses the LDS ring for the finished tile and a
    second GEMM behind one barrier: at the bench size (32 x 1024^2, one workgroup on every CU for 32 tiles each) repeated
    forwards must reproduce layer1's maps bit for bit."""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=2), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    base = synth.em_tiles(4, 1024, seed=9)
    x = torch.from_numpy(normalize(np.concatenate([base] * 8), 0.57571, 0.12765))[:, None].cuda()
    first = None
    for rep in range(6):
        model(x, 2, False)
        cur = [model.tap(t).clone() for t in ('encoder.layer1.1.c1', 'encoder.layer1.2.c1', 'encoder.layer1.1')]
        if first is None:
            first = cur
            # the batch holds 8 copies of 4 tiles: copies must agree with each other (tile-position independence)
            for t in cur:
                assert torch.equal(t[:4], t[4:8]) and torch.equal(t[:4], t[28:32])
        else:
            for a, b in zip(first, cur):
                assert torch.equal(a, b), f'repetition {rep}'
