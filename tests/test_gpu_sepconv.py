"""Fused separable conv (csrc/sepconv.hip: depthwise 5x5 -> pointwise -> bias/act [-> 1x1 head]) vs
 (a) the unfused HIP pair dwconv + implicit-GEMM conv on the same operands: same summation order,
     asserted BIT-EXACT;
 (b) a torch fp32 reference of the same op with the depthwise result rounded to fp16 (the precision the
     engine keeps between the two convs): |err| <= 2e-3 + 2e-3*|ref| (one fp16 rounding of the output
     plus accumulation-order noise);
 (c) head mode: fp32 planes vs head_w . relu(y_fp32) + head_b, where y is never rounded to fp16:
     tolerance 1e-3 + 1e-3*|ref|."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, C, in_ld, Cout, act
    (2, 16, 32, 128, 128, 128, 1),
    (1, 24, 40, 320, 320, 256, 1),       # W % 16 = 8: ragged tile column
    (3, 13, 21, 256, 256, 256, 1),       # odd sizes, ragged rows and columns
    (1, 8, 16, 128, 192, 256, 0),        # channel slice of a wider buffer, no activation
    (2, 40, 48, 192, 192, 128, 2),       # SiLU
    (1, 72, 272, 256, 256, 256, 1),      # more tiles than workgroups -> several tiles per workgroup
]


def _operands(case, seed=0):
    N, H, W, Cc, in_ld, Cout, act = case
    g = torch.Generator().manual_seed(seed + hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, in_ld), generator=g).to(torch.float16)
    dw = (torch.randn((Cc, 5, 5), generator=g) * 0.2).to(torch.float16)
    pw = (torch.randn((Cout, Cc), generator=g) / np.sqrt(Cc)).to(torch.float16)
    b = torch.randn((Cout,), generator=g) * 0.1
    return x, dw, pw, b


def _apply_act(y, act):
    if act == 1:
        return torch.relu(y)
    if act == 2:
        return y * torch.sigmoid(y)
    return y


def _ref(x, dw, pw, b, Cc, act, round_dw=True):
    xin = x[..., :Cc].float().permute(0, 3, 1, 2)
    d = F.conv2d(xin, dw.float()[:, None], padding=2, groups=Cc)
    if round_dw:
        d = d.to(torch.float16).float()
    y = F.conv2d(d, pw.float()[:, :, None, None], b)
    return _apply_act(y, act)            # (N,Cout,H,W) fp32


def _fused(x, dw, pw, b, case, head=None):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, in_ld, Cout, act = case
    xd = x.to(dev())
    dwd = dw.reshape(Cc, 25).t().contiguous().to(dev())          # (25, C)
    pwu = pw.contiguous().to(dev())
    pwd = torch.empty_like(pwu)
    _abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pwu), Cc, Cc, Cout, _abi.ptr(pwd), _abi.stream_ptr(dev())), 'pack')
    bd = b.float().to(dev())
    if head is None:
        out = torch.full((N, H, W, Cout), 7.0, dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), _abi.ptr(pwd),
                                               _abi.ptr(bd), Cout, act, _abi.ptr(out), Cout, None, None, 0, None,
                                               _abi.stream_ptr(dev())), 'sepconv')
        torch.cuda.synchronize()
        return out
    hw, hb = head
    hc = hw.shape[0]
    hout = torch.full((N, hc, H, W), 7.0, dtype=torch.float32, device=dev())
    hwd, hbd = hw.float().contiguous().to(dev()), hb.float().to(dev())
    _abi.check(lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), _abi.ptr(pwd),
                                           _abi.ptr(bd), Cout, act, None, 0, _abi.ptr(hwd), _abi.ptr(hbd), hc,
                                           _abi.ptr(hout), _abi.stream_ptr(dev())), 'sepconv head')
    torch.cuda.synchronize()
    return hout


def _unfused(x, dw, pw, b, case):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, in_ld, Cout, act = case
    xd = x.to(dev())
    dwd = dw.reshape(Cc, 25).t().contiguous().to(dev())
    mid = torch.empty((N, H, W, Cc), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_dwconv_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), 5, _abi.ptr(mid), Cc,
                                       _abi.stream_ptr(dev())), 'dwconv')
    out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
    pwd = pw.contiguous().to(dev())
    bd = b.float().to(dev())
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(mid), N, H, W, Cc, Cc, _abi.ptr(pwd), _abi.ptr(bd), None, None, 0,
                                       _abi.ptr(out), Cout, Cout, 1, 1, 1, 0, 1, act, 0, _abi.stream_ptr(dev())),
               'conv')
    torch.cuda.synchronize()
    return out, mid


@pytest.mark.parametrize('case', CASES)
def test_fused_matches_fp32_reference(case):
    x, dw, pw, b = _operands(case)
    y = _fused(x, dw, pw, b, case).float().cpu().permute(0, 3, 1, 2)
    ref = _ref(x, dw, pw, b, case[3], case[6])
    err = (y - ref).abs()
    tol = 2e-3 + 2e-3 * ref.abs()
    assert torch.all(err <= tol), f'max err {err.max():.4e} at ref {ref.flatten()[err.argmax()]:.4f}'


@pytest.mark.parametrize('case', CASES)
def test_fused_equals_unfused_pair_bit_exact(case):
    x, dw, pw, b = _operands(case, seed=1)
    y = _fused(x, dw, pw, b, case)
    u, _ = _unfused(x, dw, pw, b, case)
    assert torch.equal(y, u), f'{(y.float() - u.float()).abs().max().item():.3e} max difference'


def test_dwconv_matches_fp32_reference():
    case = (2, 19, 37, 128, 128, 128, 0)
    x, dw, pw, b = _operands(case, seed=2)
    _, mid = _unfused(x, dw, pw, b, case)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), dw.float()[:, None], padding=2, groups=128).permute(0, 2, 3, 1)
    err = (mid.float().cpu() - ref).abs()
    assert torch.all(err <= 1e-3 + 1e-3 * ref.abs()), err.max()


@pytest.mark.parametrize('hc', [1, 2])
@pytest.mark.parametrize('case', [CASES[4], CASES[2], CASES[0], CASES[5]])   # heads: C <= 256 (LDS budget)
def test_fused_head(case, hc):
    x, dw, pw, b = _operands(case, seed=3)
    Cout = case[5]
    g = torch.Generator().manual_seed(hc)
    hw = torch.randn((hc, Cout), generator=g) / np.sqrt(Cout)
    hb = torch.randn((hc,), generator=g)
    out = _fused(x, dw, pw, b, case, head=(hw, hb)).cpu()
    y = _ref(x, dw, pw, b, case[3], case[6])
    ref = F.conv2d(y, hw[:, :, None, None], hb)
    err = (out - ref).abs()
    assert torch.all(err <= 1e-3 + 1e-3 * ref.abs()), f'max err {err.max():.4e}'


def test_unsupported_shape_is_rejected():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    x = torch.zeros((1, 8, 16, 64), dtype=torch.float16, device=dev())
    w = torch.zeros((25, 64), dtype=torch.float16, device=dev())
    p = torch.zeros((64, 64), dtype=torch.float16, device=dev())
    o = torch.zeros((1, 8, 16, 64), dtype=torch.float16, device=dev())
    rc = lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(x), 1, 8, 16, 64, 64, _abi.ptr(w), _abi.ptr(p), None, 64, 1,
                                     _abi.ptr(o), 64, None, None, 0, None, _abi.stream_ptr(dev()))
    assert rc != 0 and b'unsupported' in lib.emp_last_error()


CASES3 = [
    # N, H, W, C, in_ld, Cout, act   (depthwise 3x3: the BiFPN node's separable conv, SiLU after the folded BN)
    (2, 16, 32, 128, 128, 128, 2),
    (3, 13, 21, 128, 128, 128, 2),       # ragged rows and columns
    (1, 64, 128, 128, 128, 128, 2),      # one tile per workgroup and more
    (1, 8, 16, 128, 192, 256, 0),        # channel slice, Cout 256
    (2, 40, 48, 256, 256, 256, 1),
]


@pytest.mark.parametrize('case', CASES3)
def test_fused_3x3_equals_unfused_pair_and_reference(case):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, in_ld, Cout, act = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, in_ld), generator=g).to(torch.float16)
    dw = (torch.randn((Cc, 3, 3), generator=g) * 0.3).to(torch.float16)
    pw = (torch.randn((Cout, Cc), generator=g) / np.sqrt(Cc)).to(torch.float16)
    b = torch.randn((Cout,), generator=g) * 0.1
    xd = x.to(dev())
    dwd = dw.reshape(Cc, 9).t().contiguous().to(dev())           # (9, C)
    pwu = pw.contiguous().to(dev())
    pwd = torch.empty_like(pwu)
    _abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pwu), Cc, Cc, Cout, _abi.ptr(pwd), _abi.stream_ptr(dev())), 'pack')
    bd = b.float().to(dev())
    mid = torch.empty((N, H, W, Cc), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_dwconv_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), 3, _abi.ptr(mid), Cc,
                                       _abi.stream_ptr(dev())), 'dwconv')
    u = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(mid), N, H, W, Cc, Cc, _abi.ptr(pwu), _abi.ptr(bd), None, None, 0,
                                       _abi.ptr(u), Cout, Cout, 1, 1, 1, 0, 1, act, 0, _abi.stream_ptr(dev())), 'conv')
    for rep in range(3):
        y = torch.full((N, H, W, Cout), 7.0, dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconv3x3_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), _abi.ptr(pwd), _abi.ptr(bd),
                                               Cout, act, _abi.ptr(y), Cout, _abi.stream_ptr(dev())), 'sepconv3')
        torch.cuda.synchronize()
        assert torch.equal(y, u), f'rep {rep}: {(y.float() - u.float()).abs().max().item():.3e} max difference to dwconv + conv'
    xin = x[..., :Cc].float().permute(0, 3, 1, 2)
    d = F.conv2d(xin, dw.float()[:, None], padding=1, groups=Cc).to(torch.float16).float()
    ref = _apply_act(F.conv2d(d, pw.float()[:, :, None, None], b), act)
    err = (y.float().cpu().permute(0, 3, 1, 2) - ref).abs()
    assert torch.all(err <= 2e-3 + 2e-3 * ref.abs()), err.max()
