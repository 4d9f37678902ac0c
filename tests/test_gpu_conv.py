"""Implicit-GEMM MFMA conv (csrc/conv_igemm.hip) vs a torch fp32 reference of
the same op on the same fp16-rounded operands.  Tolerance: the kernel
accumulates in fp32 and rounds the result to fp16 once -> |err| <= 2^-10 |y| +
small accumulation-order noise; asserted as atol 2e-3*scale, rtol 2e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, relu, res, bias_n
    (2, 16, 16, 64, 64, 1, 1, 0, 1, True, False, False),
    (1, 16, 24, 64, 256, 1, 1, 0, 1, True, True, False),
    (2, 16, 16, 128, 128, 3, 1, 1, 1, True, False, False),
    (1, 20, 12, 128, 128, 3, 2, 1, 1, True, False, False),     # stride 2, M tail
    (1, 16, 16, 256, 512, 1, 2, 0, 1, False, False, False),    # downsample
    (1, 12, 12, 512, 256, 3, 1, 6, 6, True, False, False),     # ASPP rate 6 (mostly padding)
    (2, 8, 8, 2048, 256, 3, 1, 2, 2, True, False, False),      # ASPP rate 2, K = 18432
    (2, 8, 8, 1024, 256, 1, 1, 0, 1, True, False, True),       # projection with per-image bias
    (1, 24, 24, 256, 32, 1, 1, 0, 1, True, False, False),      # low-level project (Cout 32)
    (1, 24, 24, 256, 16, 1, 1, 0, 1, True, False, False),      # Cout 16
    (1, 16, 16, 288, 256, 1, 1, 0, 1, True, False, False),     # Cin padded 288 -> 320
    (1, 1, 300, 257, 256, 1, 1, 0, 1, True, False, False),     # PointRend MLP shape (Cin 257)
    (3, 9, 7, 64, 64, 3, 1, 1, 1, False, True, False),         # odd sizes + residual, no relu
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 16 + 3, 32 + 1, 32 + 2, 32 + 3, 48 + 1, 48 + 2, 48 + 3, 112,
                                     (1 << 8) + 3, (2 << 8) + 19, (4 << 8) + 35])   # bits 8+: K-walk group size; 112: deep-ring 64x64
@pytest.mark.parametrize('case', CASES)
def test_conv_matches_fp32_reference(case, variant):
    from gpu_common import conv_hip, conv_ref, dev
    N, H, W, Cin, Cout, k, stride, pad, dil, relu, use_res, use_bn = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = (torch.randn((N, H, W, Cin), generator=g)).to(torch.float16).to(dev())
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / np.sqrt(Cin * k * k))
    b = torch.randn((Cout,), generator=g) * 0.1
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(torch.float16).to(dev()) if use_res else None
    bn = torch.randn((N, Cout), generator=g) * 0.2 if use_bn else None
    y = conv_hip(x, w, b, bn, res, stride, pad, dil, relu, variant).float().cpu()
    ref = conv_ref(x, w, b, bn, res, stride, pad, dil, relu)
    err = (y - ref).abs()
    tol = 2e-3 + 2e-3 * ref.abs()
    assert torch.all(err <= tol), f'max err {err.max():.4e} at ref {ref.flatten()[err.argmax()]:.4f}'


def test_conv_writes_only_its_channel_slice():
    """torch.cat elimination: the conv writes channels [coff, coff+Cout) of a wider buffer."""
    from gpu_common import conv_hip, conv_ref, dev
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1, 8, 8, 64), generator=g).to(torch.float16).to(dev())
    w = torch.randn((32, 64, 1, 1), generator=g) * 0.1
    out = conv_hip(x, w, None, None, None, 1, 0, 1, True, 0, out_ld=320, out_coff=256).float().cpu()
    ref = conv_ref(x, w, None, None, None, 1, 0, 1, True)
    assert torch.all(out[..., :256] == 7.0) and torch.all(out[..., 288:] == 7.0)
    assert torch.allclose(out[..., 256:288], ref, atol=2e-3, rtol=2e-3)


def test_conv_rejects_bad_arguments():
    from empanada_napari_amd import _abi
    from gpu_common import dev
    lib = _abi.load()
    x = torch.zeros((1, 4, 4, 48), dtype=torch.float16, device=dev())
    rc = lib.emp_conv2d_nhwc_f16(_abi.ptr(x), 1, 4, 4, 48, 48, _abi.ptr(x), None, None, None, 0, _abi.ptr(x), 8, 8,
                                 1, 1, 1, 0, 1, 0, 0, None)
    assert rc == -1 and b'multiple of 64' in lib.emp_last_error()


CASES_256 = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, relu, res, bias_n   (Cout % 256 == 0: the 256x256 tile, variant 64)
    (1, 16, 24, 128, 256, 1, 1, 0, 1, True, True, False),       # 4 K-tiles: the shortest K the ring accepts
    (1, 16, 16, 256, 512, 1, 2, 0, 1, False, False, False),     # strided 1x1, two cout tiles
    (1, 12, 12, 512, 256, 3, 1, 6, 6, True, False, False),      # ASPP rate 6, group-major K walk (16 slabs)
    (2, 8, 8, 2048, 256, 3, 1, 2, 2, True, False, False),       # K = 18432
    (2, 8, 8, 1024, 256, 1, 1, 0, 1, True, False, True),        # per-image bias
    (1, 1, 300, 288, 256, 1, 1, 0, 1, True, False, False),      # M tail (300 = 256 + 44), Cin padded to 320
    (3, 24, 24, 128, 256, 3, 1, 1, 1, False, True, False),      # 7 pixel tiles, residual, no relu
    (1, 40, 40, 512, 1024, 1, 1, 0, 1, True, True, False),      # 4 cout tiles
]


@pytest.mark.parametrize('case', CASES_256)
def test_conv256_matches_fp32_reference(case):
    from gpu_common import conv_hip, conv_ref, dev
    N, H, W, Cin, Cout, k, stride, pad, dil, relu, use_res, use_bn = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = (torch.randn((N, H, W, Cin), generator=g)).to(torch.float16).to(dev())
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / np.sqrt(Cin * k * k))
    b = torch.randn((Cout,), generator=g) * 0.1
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(torch.float16).to(dev()) if use_res else None
    bn = torch.randn((N, Cout), generator=g) * 0.2 if use_bn else None
    ref = conv_ref(x, w, b, bn, res, stride, pad, dil, relu)
    for rep in range(3):      # the pipeline spans barriers with DMA in flight: repeat to catch timing-dependent races
        y = conv_hip(x, w, b, bn, res, stride, pad, dil, relu, 64).float().cpu()
        err = (y - ref).abs()
        tol = 2e-3 + 2e-3 * ref.abs()
        assert torch.all(err <= tol), f'rep {rep}: max err {err.max():.4e} at ref {ref.flatten()[err.argmax()]:.4f}'


def test_conv256_rejects_other_cout():
    from gpu_common import conv_hip, dev
    from empanada_napari_amd._abi import EmpError
    x = torch.zeros((1, 8, 8, 64), dtype=torch.float16, device=dev())
    w = torch.zeros((64, 64, 1, 1))
    with pytest.raises(EmpError):
        conv_hip(x, w, None, None, None, 1, 0, 1, True, 64)


CASES_DUAL = [
    # N, H, W, Cin, H2, W2, Cin2, stride2, Cout, variant
    (1, 24, 40, 64, 24, 40, 64, 1, 256, 0),          # layer1.0: conv3 (64) + shortcut (64), same resolution
    (1, 24, 40, 64, 24, 40, 64, 1, 256, 64),         # the same on the 256x256 tile (2 + 2 K-tiles: the shortest ring)
    (2, 12, 20, 128, 24, 40, 256, 2, 512, 0),        # layer2.0: the shortcut samples every second pixel
    (2, 12, 20, 128, 23, 39, 256, 2, 512, 64),       # odd source size ((H-1)*2 < H2)
    (1, 8, 8, 512, 8, 8, 1024, 1, 2048, 0),          # layer4.0 (output stride 16): auto -> 256x256 tile, 48 K-tiles
    (1, 9, 7, 64, 9, 7, 128, 1, 64, 0),              # Cout 64 -> the 128x64 tile; M tail
    (1, 16, 16, 192, 31, 31, 64, 2, 128, 16 + 1),    # register-staged variant of the 128x128 tile
]


@pytest.mark.parametrize('case', CASES_DUAL)
def test_conv1x1_dual_source_matches_fp32_reference(case):
    """out = relu(x . W1 + x2[::s, ::s] . W2 + b): the bottleneck's conv3 with the projection shortcut K-concatenated"""
    import ctypes as C
    from gpu_common import dev
    from empanada_napari_amd import _abi
    N, H, W, Cin, H2, W2, Cin2, s2, Cout, variant = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.float16)
    x2 = torch.randn((N, H2, W2, Cin2), generator=g).to(torch.float16)
    w = (torch.randn((Cout, Cin + Cin2), generator=g) / np.sqrt(Cin + Cin2)).to(torch.float16)
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = x.float() @ w[:, :Cin].float().T + x2[:, ::s2, ::s2][:, :H, :W].float() @ w[:, Cin:].float().T + b
    ref = torch.relu(ref)
    xd, x2d, wd, bd = x.to(dev()), x2.to(dev()), w.to(dev()).contiguous(), b.float().to(dev())
    lib = _abi.load()
    for rep in range(3):
        out = torch.full((N, H, W, Cout), 7.0, dtype=torch.float16, device=dev())
        _abi.check(lib.emp_conv1x1_dual_nhwc_f16(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(x2d), H2, W2, Cin2, Cin2, s2,
                                                 _abi.ptr(wd), _abi.ptr(bd), _abi.ptr(out), Cout, Cout, 1, variant,
                                                 _abi.stream_ptr(dev())), 'emp_conv1x1_dual_nhwc_f16')
        torch.cuda.synchronize()
        err = (out.float().cpu() - ref).abs()
        tol = 2e-3 + 2e-3 * ref.abs()
        assert torch.all(err <= tol), f'rep {rep}: max err {err.max():.4e}'


def test_conv1x1_dual_source_rejects_bad_geometry():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    from empanada_napari_amd._abi import EmpError
    lib = _abi.load()
    x = torch.zeros((1, 8, 8, 64), dtype=torch.float16, device=dev())
    x2 = torch.zeros((1, 8, 8, 64), dtype=torch.float16, device=dev())
    w = torch.zeros((64, 128), dtype=torch.float16, device=dev())
    out = torch.zeros((1, 8, 8, 64), dtype=torch.float16, device=dev())
    with pytest.raises(EmpError, match='does not cover'):        # stride 2 over an 8x8 source cannot feed 8x8 outputs
        _abi.check(lib.emp_conv1x1_dual_nhwc_f16(_abi.ptr(x), 1, 8, 8, 64, 64, _abi.ptr(x2), 8, 8, 64, 64, 2, _abi.ptr(w), None,
                                                 _abi.ptr(out), 64, 64, 1, 0, _abi.stream_ptr(dev())), 'dual')


@pytest.mark.parametrize('shape', [(1, 8, 16), (1, 16, 32), (2, 20, 37), (3, 64, 64), (1, 9, 250), (5, 40, 48)])
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('C', [64, 128])
def test_conv3x3_c64_register_weights_equals_implicit_gemm(shape, relu, C):
    """variant 96: the 64 -> 64 channel 3x3 kernel with register-resident weights and an LDS halo tile; same K order as
    the implicit-GEMM kernel, so the outputs are identical (image borders, tile tails, several tiles per workgroup)"""
    from gpu_common import conv_hip, conv_ref, dev
    N, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn((N, H, W, C), generator=g).to(torch.float16).to(dev())
    w = torch.randn((C, C, 3, 3), generator=g) / (3.0 * np.sqrt(C))
    b = torch.randn((C,), generator=g) * 0.1
    base = conv_hip(x, w, b, None, None, 1, 1, 1, relu, 16 + 3)
    ref = conv_ref(x, w, b, None, None, 1, 1, 1, relu)
    for rep in range(3):
        y = conv_hip(x, w, b, None, None, 1, 1, 1, relu, 96)
        err = (y.float().cpu() - ref).abs()
        assert torch.all(err <= 2e-3 + 2e-3 * ref.abs()), f'rep {rep}: max err {err.max():.4e}'
        assert torch.equal(y, base), f'rep {rep}: differs from the implicit-GEMM kernel ({(y != base).sum().item()} values)'


@pytest.mark.parametrize('case', CASES_256 + [CASES[4], CASES[5], CASES[6], CASES[10]])
def test_conv_tile_variants_are_bit_identical(case):
    """Every tile walks K in the same order (256-channel groups, tap-major, ascending 32-channel MFMA steps from a zero
    accumulator) and applies the same fp32 epilogue: 128x128 (16), 64x64 (48), 256x256 (64), the half tile (80) and the
    deep-ring 64x64 tile of the batch-1 path (112) must agree bit for bit -- which tile runs depends on the batch, and a
    batch of N has to equal N batch-1 calls."""
    from gpu_common import conv_hip, dev
    N, H, W, Cin, Cout, k, stride, pad, dil, relu, use_res, use_bn = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = (torch.randn((N, H, W, Cin), generator=g)).to(torch.float16).to(dev())
    w = torch.randn((Cout, Cin, k, k), generator=g) * (1.0 / np.sqrt(Cin * k * k))
    b = torch.randn((Cout,), generator=g) * 0.1
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn((N, Ho, Wo, Cout), generator=g).to(torch.float16).to(dev()) if use_res else None
    bn = torch.randn((N, Cout), generator=g) * 0.2 if use_bn else None
    tiles = [16, 48, 112] + ([64] if Cout % 256 == 0 else []) + ([80] if Cout % 128 == 0 else [])
    ref = None
    for rep in range(2):
        for t in tiles:
            y = conv_hip(x, w, b, bn, res, stride, pad, dil, relu, t + 3)
            if ref is None:
                ref = y.clone()
            assert torch.equal(ref, y), f'tile code {t} (rep {rep}) differs in {(ref != y).float().mean().item():.2e} of the elements'
