"""Fused separable conv with the depthwise conv on the matrix pipe (csrc/sepconv_mma.hip): block-diagonal fp16 tap
fragments, A fragments straight from the swizzled halo tile, fp32 accumulation.
 lo = 1: the depthwise result reaches the pointwise conv as an fp16 hi + lo pair -- against the fp64 reference with fp16
         taps, fp16 pointwise weights and NOTHING rounded in between: half an fp16 ulp (+ summation-order noise);
 lo = 0: the depthwise result is rounded to fp16 (the precision of sepconv.hip) -- against the fp64 reference with that
         rounding: 2e-3 + 2e-3 |ref| (a different fp32 summation order moves a few depthwise values across a rounding
         boundary);
 head mode, ragged tiles, C up to 512, 3x3 taps, repeatability and batch invariance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_gpu_sepconv_precise as tp

pytestmark = pytest.mark.gpu


def _fused(x, dw, pw, b, case, lo, head=None, ks=5):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cc, in_ld, Cout, act = case
    xd = x.to(dev())
    dwu = dw.reshape(Cc, ks * ks).t().contiguous().float().to(dev())          # (ks*ks, C) fp32
    npair = (ks * ks + 1) // 2
    dwd = torch.empty((Cc // 16) * npair * 512, dtype=torch.float16, device=dev())
    _abi.check(lib.emp_sepconvm_pack_dw(_abi.ptr(dwu), ks, Cc, _abi.ptr(dwd), _abi.stream_ptr(dev())), 'pack_dw')
    pwu = pw.contiguous().float().to(dev())
    pwd = torch.empty((Cout, Cc), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_sepconvp_pack_pw(_abi.ptr(pwu), Cc, Cc, Cout, _abi.ptr(pwd), _abi.stream_ptr(dev())), 'pack_pw')
    bd = b.float().to(dev())
    if head is None:
        out = torch.full((N, H, W, Cout), 7.0, dtype=torch.float16, device=dev())
        _abi.check(lib.emp_sepconvm_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, ks, lo, _abi.ptr(dwd), _abi.ptr(pwd),
                                             _abi.ptr(bd), Cout, act, _abi.ptr(out), Cout, None, None, 0, None,
                                             _abi.stream_ptr(dev())), 'sepconvm')
        torch.cuda.synchronize()
        return out
    hw, hb = head
    hc = hw.shape[0]
    hout = torch.full((N, hc, H, W), 7.0, dtype=torch.float32, device=dev())
    hwd, hbd = hw.float().contiguous().to(dev()), hb.float().to(dev())
    _abi.check(lib.emp_sepconvm_nhwc_f16(_abi.ptr(xd), N, H, W, Cc, in_ld, ks, lo, _abi.ptr(dwd), _abi.ptr(pwd),
                                         _abi.ptr(bd), Cout, act, None, 0, _abi.ptr(hwd), _abi.ptr(hbd), hc,
                                         _abi.ptr(hout), _abi.stream_ptr(dev())), 'sepconvm head')
    torch.cuda.synchronize()
    return hout


def _ref64(x, dw, pw, b, Cc, act, ks, round_dw):
    xin = x[..., :Cc].double().permute(0, 3, 1, 2)
    d = F.conv2d(xin, dw.to(torch.float16).double()[:, None], padding=ks // 2, groups=Cc)
    if round_dw:
        d = d.to(torch.float16).double()
    return tp._apply_act(F.conv2d(d, pw.to(torch.float16).double()[:, :, None, None], b.double()), act)


@pytest.mark.parametrize('case', tp.CASES)
def test_lo_is_the_correctly_rounded_result(case):
    x, dw, pw, b = tp._operands(case)
    tp._check_fp16_output(_fused(x, dw, pw, b, case, 1), _ref64(x, dw, pw, b, case[3], case[6], 5, False), 'mma 5x5 lo')


@pytest.mark.parametrize('case', tp.CASES3)
def test_lo_3x3(case):
    x, dw, pw, b = tp._operands(case, ks=3)
    tp._check_fp16_output(_fused(x, dw, pw, b, case, 1, ks=3), _ref64(x, dw, pw, b, case[3], case[6], 3, False), 'mma 3x3 lo')


@pytest.mark.parametrize('ks,case', [(5, tp.CASES[0]), (5, tp.CASES[1]), (5, tp.CASES[2]), (5, tp.CASES[5]), (3, tp.CASES3[1]), (3, tp.CASES3[4])])
def test_fp16_depthwise_result(ks, case):
    x, dw, pw, b = tp._operands(case, ks=ks, seed=1)
    y = _fused(x, dw, pw, b, case, 0, ks=ks).double().cpu().permute(0, 3, 1, 2)
    ref = _ref64(x, dw, pw, b, case[3], case[6], ks, True)
    err = (y - ref).abs()
    assert bool((err <= 2e-3 + 2e-3 * ref.abs()).all()), float(err.max())


@pytest.mark.parametrize('lo', [0, 1])
@pytest.mark.parametrize('hc', [1, 2])
@pytest.mark.parametrize('case', [tp.CASES[4], tp.CASES[2], tp.CASES[0], tp.CASES[5]])
def test_head(case, hc, lo):
    x, dw, pw, b = tp._operands(case, seed=3)
    Cout = case[5]
    g = torch.Generator().manual_seed(hc)
    hw = torch.randn((hc, Cout), generator=g) / np.sqrt(Cout)
    hb = torch.randn((hc,), generator=g)
    out = _fused(x, dw, pw, b, case, lo, head=(hw, hb)).double().cpu()
    y = _ref64(x, dw, pw, b, case[3], case[6], 5, not lo)
    ref = F.conv2d(y, hw.double()[:, :, None, None], hb.double())
    scale = float((y.abs().amax(1, keepdim=True) * hw.abs().sum(1).max()).max())
    err = (out - ref).abs()
    assert float(err.max()) <= (4e-6 if lo else 1e-3) * scale, f'max err {float(err.max()):.4e} (scale {scale:.2f})'


@pytest.mark.parametrize('lo', [0, 1])
def test_repeatable_and_batch_invariant(lo):
    case = tp.CASES[2]
    x, dw, pw, b = tp._operands(case, seed=4)
    y0 = _fused(x, dw, pw, b, case, lo)
    for _ in range(3):
        assert torch.equal(_fused(x, dw, pw, b, case, lo), y0)
    for i in range(case[0]):
        one = (1,) + tuple(case[1:])
        assert torch.equal(_fused(x[i:i + 1].contiguous(), dw, pw, b, one, lo)[0], y0[i])
