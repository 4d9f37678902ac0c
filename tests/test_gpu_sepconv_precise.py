"""Fused separable conv with an exact depthwise half (csrc/sepconv_precise.hip: depthwise KxK -> pointwise -> bias/act
[-> 1x1 head]; the engine runs it for the blocks the centre heat-map depends on, pdl_net.hip precise_layer): fp32 depthwise
taps on the vector pipe, the depthwise result carried to the matrix pipe as an fp16 hi + lo pair (21 bits), fp16 pointwise
weights, fp32 accumulation; input and output maps fp16.  Checked against

 (a) an fp64 torch reference of the same op on the same fp16 input, the SAME fp32 taps, the pointwise weights rounded to
     fp16 and nothing rounded in between: the fp16 output must be the correctly rounded value up to summation-order
     noise (|err| <= half an fp16 ulp of the reference + 2e-6 * scale), the fp32 head output within 4e-6 * scale;
 (b) the unfused HIP pair dwconv + conv (fp16 taps / intermediate): the fused block must be CLOSER to the all-fp32 truth
     than the pair -- the reason it exists (profiles/r02_error_budget.csv, DESIGN finding 24);
 (c) itself: repeated launches and a batch vs its images one by one are bit-identical (fixed summation order)."""
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
    (1, 16, 16, 512, 512, 256, 1),       # 8 channel chunks (the taps are no longer held in LDS: C <= 512)
]

CASES3 = [
    # depthwise 3x3: the BiFPN node's separable conv, SiLU after the folded BN
    (2, 16, 32, 128, 128, 128, 2),
    (3, 13, 21, 128, 128, 128, 2),       # ragged rows and columns
    (1, 64, 128, 128, 128, 128, 2),      # one tile per workgroup and more
    (1, 8, 16, 128, 192, 256, 0),        # channel slice, Cout 256
    (2, 40, 48, 256, 256, 256, 1),
    (4, 8, 8, 128, 128, 128, 2),         # a P7-sized map: fewer tiles than workgroups
]


def _operands(case, ks=5, seed=0):
    N, H, W, Cc, in_ld, Cout, act = case
    g = torch.Generator().manual_seed(seed + hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, in_ld), generator=g).to(torch.float16)
    dw = torch.randn((Cc, ks, ks), generator=g) * (0.2 if ks == 5 else 0.3)          # fp32: NOT representable in fp16
    pw = torch.randn((Cout, Cc), generator=g) / np.sqrt(Cc)
    b = torch.randn((Cout,), generator=g) * 0.1
    return x, dw, pw, b


def _apply_act(y, act):
    if act == 1:
        return torch.relu(y)
    if act == 2:
        return y * torch.sigmoid(y)
    return y


def _ref64(x, dw, pw, b, Cc, act, ks=5, pw16=True):
    """fp64 reference; pw16: pointwise weights rounded to fp16 as the kernel holds them (False: the all-fp32 truth;
    'split': as the fp16 hi + lo pair of the weight-split form)"""
    xin = x[..., :Cc].double().permute(0, 3, 1, 2)
    d = F.conv2d(xin, dw.double()[:, None], padding=ks // 2, groups=Cc)
    if pw16 == 'split':
        hi = pw.to(torch.float16)
        w = hi.double() + (pw - hi.float()).to(torch.float16).double()
    else:
        w = pw.to(torch.float16).double() if pw16 else pw.double()
    return _apply_act(F.conv2d(d, w[:, :, None, None], b.double()), act)     # (N,Cout,H,W) fp64


def _fused(x, dw, pw, b, case, head=None, ks=5, ws=False):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    if ws:      # pointwise weights as an fp16 hi + lo pair (Cout == 128): same calls, the _ws_ entry points
        class _L:
            emp_sepconvp_pack_dw = lib.emp_sepconvp_pack_dw
            emp_sepconvp_pack_pw = lib.emp_sepconvp_ws_pack_pw
            emp_sepconvp_nhwc_f16 = lib.emp_sepconvp_ws_nhwc_f16
        lib = _L
    N, H, W, Cc, in_ld, Cout, act = case
    xd = x.to(dev())
    dwu = dw.reshape(Cc, ks * ks).t().contiguous().float().to(dev())          # (ks*ks, C) fp32
    dwd = torch.empty_like(dwu)                                               # chunk-major [C/64][ks*ks][64]
    _abi.check(lib.emp_sepconvp_pack_dw(_abi.ptr(dwu), ks, Cc, _abi.ptr(dwd), _abi.stream_ptr(dev())), 'pack_dw')
    pwu = pw.contiguous().float().to(dev())
    pwd = torch.empty(((2 if ws else 1) * Cout, Cc), dtype=torch.float16, device=dev())          # fragment order (ws: hi, then lo)
    _abi.check(lib.emp_sepconvp_pack_pw(_abi.ptr(pwu), Cc, Cc, Cout, _abi.ptr(pwd), _abi.stream_ptr(dev())), 'pack')
    bd = b.float().to(dev())
    if head is None:
        out = torch.full((N, H, W, Cout), 7.0, dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconvp_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, ks, _abi.ptr(dwd), _abi.ptr(pwd),
                                             _abi.ptr(bd), Cout, act, _abi.ptr(out), Cout, None, None, 0, None,
                                             _abi.stream_ptr(dev())), 'sepconvp')
        torch.cuda.synchronize()
        return out
    hw, hb = head
    hc = hw.shape[0]
    hout = torch.full((N, hc, H, W), 7.0, dtype=torch.float32, device=dev())
    hwd, hbd = hw.float().contiguous().to(dev()), hb.float().to(dev())
    _abi.check(lib.emp_sepconvp_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, ks, _abi.ptr(dwd), _abi.ptr(pwd),
                                         _abi.ptr(bd), Cout, act, None, 0, _abi.ptr(hwd), _abi.ptr(hbd), hc,
                                         _abi.ptr(hout), _abi.stream_ptr(dev())), 'sepconvp head')
    torch.cuda.synchronize()
    return hout


def _unfused(x, dw, pw, b, case, ks=5):
    """the fp16 pair the fused block replaces: taps, intermediate map and pointwise weights rounded to fp16"""
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, in_ld, Cout, act = case
    xd = x.to(dev())
    dwd = dw.reshape(Cc, ks * ks).t().contiguous().to(torch.float16).to(dev())
    mid = torch.empty((N, H, W, Cc), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_dwconv_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, _abi.ptr(dwd), ks, _abi.ptr(mid), Cc,
                                       _abi.stream_ptr(dev())), 'dwconv')
    out = torch.empty((N, H, W, Cout), dtype=torch.float16, device=dev())
    pwd = pw.contiguous().to(torch.float16).to(dev())
    bd = b.float().to(dev())
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(mid), N, H, W, Cc, Cc, _abi.ptr(pwd), _abi.ptr(bd), None, None, 0,
                                       _abi.ptr(out), Cout, Cout, 1, 1, 1, 0, 1, act, 0, _abi.stream_ptr(dev())),
               'conv')
    torch.cuda.synchronize()
    return out, mid


def _half_ulp(ref):
    """half an fp16 ulp of |ref| (normal range; 2^-25 below it)"""
    e = torch.floor(torch.log2(ref.abs().clamp_min(2.0 ** -14)))
    return 2.0 ** (e - 11)


def _check_fp16_output(y, ref64, what):
    y = y.double().cpu().permute(0, 3, 1, 2)
    scale = float(ref64.abs().max())
    err = (y - ref64).abs()
    tol = _half_ulp(ref64) + 2e-6 * scale
    bad = err > tol
    assert not bool(bad.any()), (f'{what}: {int(bad.sum())} elements beyond half an fp16 ulp; max err {float(err.max()):.3e} '
                                 f'at ref {float(ref64.flatten()[err.argmax()]):.4f}')


@pytest.mark.parametrize('case', CASES)
def test_fused_is_the_correctly_rounded_fp32_result(case):
    x, dw, pw, b = _operands(case)
    _check_fp16_output(_fused(x, dw, pw, b, case), _ref64(x, dw, pw, b, case[3], case[6]), '5x5')


@pytest.mark.parametrize('case', CASES3)
def test_fused_3x3_is_the_correctly_rounded_fp32_result(case):
    x, dw, pw, b = _operands(case, ks=3)
    _check_fp16_output(_fused(x, dw, pw, b, case, ks=3), _ref64(x, dw, pw, b, case[3], case[6], ks=3), '3x3')


@pytest.mark.parametrize('ks,case', [(5, CASES[1]), (5, CASES[5]), (3, CASES3[2]), (3, CASES3[4])])
def test_fused_beats_the_fp16_pair(ks, case):
    x, dw, pw, b = _operands(case, ks=ks, seed=1)
    ref = _ref64(x, dw, pw, b, case[3], case[6], ks=ks, pw16=False)
    y = _fused(x, dw, pw, b, case, ks=ks).double().cpu().permute(0, 3, 1, 2)
    u = _unfused(x, dw, pw, b, case, ks=ks)[0].double().cpu().permute(0, 3, 1, 2)
    rms_f = float((y - ref).pow(2).mean().sqrt())
    rms_u = float((u - ref).pow(2).mean().sqrt())
    print(f'rms error vs fp64: fused {rms_f:.3e}, dwconv + conv {rms_u:.3e}')
    assert rms_f < 0.85 * rms_u


@pytest.mark.parametrize('hc', [1, 2])
@pytest.mark.parametrize('case', [CASES[4], CASES[2], CASES[0], CASES[5], CASES[1]])
def test_fused_head(case, hc):
    x, dw, pw, b = _operands(case, seed=3)
    Cout = case[5]
    g = torch.Generator().manual_seed(hc)
    hw = torch.randn((hc, Cout), generator=g) / np.sqrt(Cout)
    hb = torch.randn((hc,), generator=g)
    out = _fused(x, dw, pw, b, case, head=(hw, hb)).double().cpu()
    y = _ref64(x, dw, pw, b, case[3], case[6])
    ref = F.conv2d(y, hw.double()[:, :, None, None], hb.double())
    # error scale: the head sums Cout terms of magnitude |y| * |hw|
    scale = float((y.abs().amax(1, keepdim=True) * hw.abs().sum(1).max()).max())
    err = (out - ref).abs()
    assert float(err.max()) <= 4e-6 * scale, f'max err {float(err.max()):.4e} (scale {scale:.2f})'


@pytest.mark.parametrize('ks,case', [(5, CASES[2]), (5, CASES[5]), (3, CASES3[1])])
def test_repeatable_and_batch_invariant(ks, case):
    x, dw, pw, b = _operands(case, ks=ks, seed=4)
    y0 = _fused(x, dw, pw, b, case, ks=ks)
    for _ in range(3):
        assert torch.equal(_fused(x, dw, pw, b, case, ks=ks), y0)
    N = case[0]
    for i in range(N):
        one = (1,) + tuple(case[1:])
        assert torch.equal(_fused(x[i:i + 1].contiguous(), dw, pw, b, one, ks=ks)[0], y0[i]), f'image {i} alone differs'
    if ks == 5:
        g = torch.Generator().manual_seed(9)
        hw, hb = torch.randn((2, case[5]), generator=g) / 16, torch.randn((2,), generator=g)
        h0 = _fused(x, dw, pw, b, case, head=(hw, hb))
        for _ in range(3):
            assert torch.equal(_fused(x, dw, pw, b, case, head=(hw, hb)), h0)


def test_unsupported_shape_is_rejected():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    x = torch.zeros((1, 8, 16, 64), dtype=torch.float16, device=dev())
    w = torch.zeros((25, 64), dtype=torch.float32, device=dev())
    p = torch.zeros((64, 64), dtype=torch.float16, device=dev())
    o = torch.zeros((1, 8, 16, 64), dtype=torch.float16, device=dev())
    rc = lib.emp_sepconvp_nhwc_f16(_abi.ptr(x), 1, 8, 16, 64, 64, 5, _abi.ptr(w), _abi.ptr(p), None, 64, 1,
                                   _abi.ptr(o), 64, None, None, 0, None, _abi.stream_ptr(dev()))
    assert rc != 0 and b'unsupported' in lib.emp_last_error()


WS_CASES = [c for c in CASES if c[5] == 128]
WS_CASES3 = [c for c in CASES3 if c[5] == 128]


@pytest.mark.parametrize('ks,case', [(5, c) for c in WS_CASES] + [(3, c) for c in WS_CASES3])
def test_weight_split_form_is_the_correctly_rounded_result_with_hi_lo_weights(ks, case):
    """Round 4 (emp_sepconvp_ws_*, Cout == 128: the BiFPN network's node / fusion / centre-head blocks): the pointwise
    weights enter as an fp16 hi + lo pair, a third MFMA per product.  Against the fp64 reference with the SAME pair the
    fp16 output is correctly rounded; against the all-fp32 truth it is closer than the fp16-weight form; launches repeat
    bit for bit."""
    x, dw, pw, b = _operands(case, ks)
    y = _fused(x, dw, pw, b, case, ks=ks, ws=True)
    _check_fp16_output(y, _ref64(x, dw, pw, b, case[3], case[6], ks, pw16='split'), f'{ks}x{ks} ws')
    truth = _ref64(x, dw, pw, b, case[3], case[6], ks, pw16=False)
    e_ws = (y.double().cpu().permute(0, 3, 1, 2) - truth).pow(2).mean().sqrt()
    e_16 = (_fused(x, dw, pw, b, case, ks=ks).double().cpu().permute(0, 3, 1, 2) - truth).pow(2).mean().sqrt()
    assert float(e_ws) < float(e_16), (float(e_ws), float(e_16))
    assert torch.equal(y, _fused(x, dw, pw, b, case, ks=ks, ws=True))


def test_weight_split_head_mode():
    """the centre head of the BiFPN network: 5x5 block (128 -> 128, hi + lo weights) + fp32 1x1 head in one launch"""
    case = (2, 24, 40, 128, 128, 128, 1)
    x, dw, pw, b = _operands(case)
    g = torch.Generator().manual_seed(5)
    hw, hb = torch.randn((2, 128), generator=g) / np.sqrt(128), torch.randn((2,), generator=g) * 0.1
    got = _fused(x, dw, pw, b, case, head=(hw, hb), ws=True).double().cpu()
    y = _ref64(x, dw, pw, b, 128, 1, 5, pw16='split')
    ref = F.conv2d(y, hw.double()[:, :, None, None], hb.double())
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 4e-6 * scale


def test_weight_split_refuses_256_couts():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    w = torch.zeros((256, 128), dtype=torch.float32, device=dev())
    p = torch.zeros((512, 128), dtype=torch.float16, device=dev())
    assert lib.emp_sepconvp_ws_pack_pw(_abi.ptr(w), 128, 128, 256, _abi.ptr(p), _abi.stream_ptr(dev())) != 0
