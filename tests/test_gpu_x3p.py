"""Round 6: the fp16x3 convolution on the dominant kernel's machinery (csrc/conv16x3p.hip) and the `hl32` plane format it reads,
op level through the C ABI (emp_hl32_*, emp_x3p_pack_weights, emp_conv2d_hl32_f16x3), and -- ADVICE r05 -- every variant of round
5's fp16x3 kernels that emp_conv2d_nhwc_f16x3 cannot reach (long-K split-role kernel, its LDS-DMA weight image, the two-buffer
kernel, the K-concatenated second source, the fused head, act 0 / 1 / 2 on the same symbol) through emp_conv2d_nhwc_f16x3_ex.
The reference computes these convolutions in fp32 (empanada/inference/engines.py:248-255; models/encoders/resnet.py:109-129,
models/decoders/aspp.py:51-103, models/heads.py:12-15): every case is held to an fp64 convolution of the same fp32 operands
at 4 x the bound of the exact fp32 kernel, and the new kernel additionally BIT FOR BIT to round 5's kernels (same operand
split, same K order, same three products per step)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from empanada_napari_amd import _abi
    return _abi, _abi.load()


def _ref(x, w, b, stride, pad, dil, act, res=None, bias_n=None, x2=None, w2=None, stride2=1):
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride, pad, dil)
    if x2 is not None:
        ref = ref + F.conv2d(x2.permute(0, 3, 1, 2).double(), w2.double(), None, stride2)
    if bias_n is not None:
        ref = ref + bias_n.double()[:, :, None, None]
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    return ref


def _to_hl32(x, ld=None):
    """(rows..., C) fp32 cuda tensor -> hl32 buffer (rows, 2 * ld) fp16 through the library"""
    from gpu_common import dev
    _abi, lib = _lib()
    C = x.shape[-1]
    ld = ld or C
    rows = x.numel() // C
    x = x.contiguous()
    out = torch.zeros((rows, 2 * ld), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_hl32_from_f32(_abi.ptr(x), _abi.ptr(out), rows, C, C, ld, _abi.stream_ptr(dev())), 'hl32_from_f32')
    return out


def _from_hl32(h, C, ld=None):
    from gpu_common import dev
    _abi, lib = _lib()
    ld = ld or C
    rows = h.shape[0]
    out = torch.zeros((rows, C), dtype=torch.float32, device=dev())
    _abi.check(lib.emp_hl32_to_f32(_abi.ptr(h), _abi.ptr(out), rows, C, ld, C, _abi.stream_ptr(dev())), 'hl32_to_f32')
    return out


def test_hl32_is_the_kernels_operand_split():
    """hi = fp16(x), lo = fp16(x - hi) per element, laid out per 32-channel block as [32 hi | 32 lo]; hi + lo is x to 2^-21"""
    from gpu_common import dev
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((37, 96), generator=g) * 3.0).to(dev())
    h = _to_hl32(x, ld=128)                     # rows wider than C: the tail of the row is not written
    assert h.shape == (37, 256)
    blocks = h[:, :192].reshape(37, 3, 2, 32)
    hi, lo = blocks[:, :, 0].reshape(37, 96).float(), blocks[:, :, 1].reshape(37, 96).float()
    want_hi = x.half().float()
    assert torch.equal(hi, want_hi)
    assert torch.equal(lo, (x - want_hi).half().float())
    assert float(h[:, 192:].abs().max()) == 0.0
    back = _from_hl32(h, 96, ld=128)
    assert torch.equal(back, hi + lo)
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())


X3P_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, res ('' | 'f32' | 'hl32'), bias_n, out_fmt
    (2, 16, 16, 256, 256, 1, 1, 0, 1, 1, '', False, 0),          # pointwise, 8 K steps, M = 512
    (1, 20, 20, 64, 256, 3, 1, 2, 2, 1, 'hl32', False, 1),       # dilated 3x3, 18 K steps, ragged M (400), hl32 residual + output
    (2, 18, 14, 128, 512, 3, 2, 1, 1, 0, '', False, 1),          # stride 2, two cout tiles, no activation
    (3, 16, 16, 128, 256, 1, 1, 0, 1, 2, 'f32', True, 0),        # K = 128: the minimum (4 steps); SiLU, fp32 residual, per-image bias
    (1, 24, 40, 96, 256, 3, 1, 6, 6, 1, '', True, 1),            # dilation 6 (ASPP), Cin = 3 blocks, M = 960 (ragged), per-image bias
    (1, 32, 32, 512, 768, 1, 1, 0, 1, 1, 'hl32', False, 0),      # three cout tiles, M = 1024, K = 512
]


@pytest.mark.parametrize('case', X3P_CASES)
def test_conv16x3p_equals_fp64_and_round5_kernels_bit_for_bit(case):
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout, k, stride, pad, dil, act, res, use_bn, out_fmt = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g) if res else None
    bn = torch.randn((N, Cout), generator=g) if use_bn else None
    xd, bd = x.to(dev()), b.to(dev())
    wd = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dev())
    rd = r.to(dev()) if res else None
    bnd = bn.to(dev()) if use_bn else None
    K = k * k * Cin
    img = torch.zeros((2 * Cout * K,), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_x3p_pack_weights(_abi.ptr(wd), _abi.ptr(img), Cout, K, _abi.stream_ptr(dev())), 'x3p_pack')
    xh = _to_hl32(xd)
    rh = _to_hl32(rd) if res == 'hl32' else rd
    # the residual the kernel sees: hi + lo (22 bits) in the hl32 case
    r_seen = _from_hl32(rh, Cout).reshape(N, Ho, Wo, Cout).cpu() if res == 'hl32' else r
    M = N * Ho * Wo
    if out_fmt:
        out = torch.full((M, 2 * (Cout + 32)), 7.0, dtype=torch.float16, device=dev())      # a channel slice of a wider hl32 row
        old = Cout + 32
    else:
        out = torch.full((M, Cout + 8), 7.0, device=dev())
        old = Cout + 8
    _abi.check(lib.emp_conv2d_hl32_f16x3(_abi.ptr(xh), N, H, W, Cin, Cin, _abi.ptr(img), _abi.ptr(bd), _abi.ptr(bnd) if use_bn else None,
                                         _abi.ptr(rh) if res else None, Cout, 1 if res == 'hl32' else 0, _abi.ptr(out), old, out_fmt, Cout,
                                         k, k, stride, pad, dil, act, _abi.stream_ptr(dev())), 'conv16x3p')
    torch.cuda.synchronize()
    if out_fmt:
        got = _from_hl32(out, Cout, ld=old)
        assert torch.all(out[:, 2 * Cout:] == 7.0), 'wrote outside its channel slice'
    else:
        got = out[:, :Cout]
        assert torch.all(out[:, Cout:] == 7.0), 'wrote outside its channel slice'
    got = got.reshape(N, Ho, Wo, Cout).cpu()
    ref = _ref(x, w, b, stride, pad, dil, act, r_seen, bn).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    assert err < 8e-6 * scale * np.sqrt(K / 64.0 + 1.0) + (2.0 ** -21 * scale if out_fmt else 0.0), err
    # round 5's kernels on the same fp32 operands: the same split, K order and products -> the same bits (an hl32 residual is
    # not the same operand: those cases compare to the fp64 reference only; an hl32 output is the fp32 output, split)
    if res != 'hl32':
        o5 = torch.zeros((M, Cout), device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16x3(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wd), _abi.ptr(bd), _abi.ptr(bnd) if use_bn else None,
                                             _abi.ptr(rd) if res else None, Cout, _abi.ptr(o5), Cout, Cout, k, k, stride, pad, dil, act, 1, 0,
                                             _abi.stream_ptr(dev())), 'conv16x3')
        torch.cuda.synchronize()
        o5 = o5.reshape(N, Ho, Wo, Cout).cpu()
        if out_fmt:
            o5 = o5.half().float() + (o5 - o5.half().float()).half().float()
        assert torch.equal(got, o5), f'{int((got != o5).sum())} of {got.numel()} values differ from the round-5 kernel'


X3P_KSPLIT_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, res, bias_n, out_fmt, split2 | (tiles, K steps) -> S x steps per split
    (1, 16, 16, 2048, 512, 3, 1, 4, 4, 1, '', False, 0, 256),     # the merged ASPP branch: 1 x 2, 576 -> 8 x 72, two fp32 destinations
    (2, 16, 16, 512, 256, 3, 1, 2, 2, 1, 'hl32', False, 1, 0),    # 2 x 1, 144 -> 8 x 18, hl32 residual + hl32 output through the finish pass
    (1, 20, 20, 1024, 256, 1, 1, 0, 1, 2, 'f32', True, 0, 0),     # pointwise (linear walk, any split boundary): 2 x 1, 32 -> 4 x 8; ragged M
    (1, 24, 24, 96, 256, 3, 1, 1, 1, 0, '', True, 1, 0),          # 3 x 1, 27 steps (tap-major: whole taps) -> 3 x 9
]


@pytest.mark.parametrize('case', X3P_KSPLIT_CASES)
def test_conv16x3p_split_k_op_level(case):
    """Round 6 (late): the plane kernel with a K range per workgroup (conv16x3p_kernel<0, 0, 0, KSPLIT>: raw fp32 partial sums to a
    scratch; x3p_finish_kernel: ascending sum + bias / bias_n / residual / activation, fp32 or hl32 rows, optional second destination)
    through emp_conv2d_hl32_f16x3_ksplit: against fp64, and against the unsplit launch within fp32 summation order."""
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout, k, stride, pad, dil, act, res, use_bn, out_fmt, split2 = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g) if res else None
    bn = torch.randn((N, Cout), generator=g) if use_bn else None
    xd, bd = x.to(dev()), b.to(dev())
    wd = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dev())
    rd = r.to(dev()) if res else None
    bnd = bn.to(dev()) if use_bn else None
    K = k * k * Cin
    img = torch.zeros((2 * Cout * K,), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_x3p_pack_weights(_abi.ptr(wd), _abi.ptr(img), Cout, K, _abi.stream_ptr(dev())), 'x3p_pack')
    xh = _to_hl32(xd)
    rh = _to_hl32(rd) if res == 'hl32' else rd
    r_seen = _from_hl32(rh, Cout).reshape(N, Ho, Wo, Cout).cpu() if res == 'hl32' else r
    M = N * Ho * Wo
    scratch = torch.zeros((16 << 20,), device=dev())      # 64 MiB
    c0 = split2 or Cout

    def run(split):
        if out_fmt:
            o = torch.full((M, 2 * (c0 + 32)), 7.0, dtype=torch.float16, device=dev())
            o2 = torch.full((M, 2 * (Cout - c0 + 32)), 7.0, dtype=torch.float16, device=dev())
            ld, ld2 = c0 + 32, Cout - c0 + 32
        else:
            o = torch.full((M, c0 + 8), 7.0, device=dev())
            o2 = torch.full((M, Cout - c0 + 8), 7.0, device=dev())
            ld, ld2 = c0 + 8, Cout - c0 + 8
        _abi.check(lib.emp_conv2d_hl32_f16x3_ksplit(
            _abi.ptr(xh), N, H, W, Cin, Cin, _abi.ptr(img), _abi.ptr(bd), _abi.ptr(bnd) if use_bn else None, _abi.ptr(rh) if res else None, Cout,
            1 if res == 'hl32' else 0, _abi.ptr(o), ld, out_fmt, _abi.ptr(o2) if split2 else None, ld2, split2, Cout, k, k, stride, pad, dil, act,
            _abi.ptr(scratch), scratch.numel() * 4 if split else 0, _abi.stream_ptr(dev())), 'conv16x3p ksplit')
        torch.cuda.synchronize()
        parts = []
        for t, c, l in ((o, c0, ld), (o2, Cout - c0, ld2)):
            if not c:
                continue
            if out_fmt:
                assert torch.all(t[:, 2 * c:] == 7.0), 'wrote outside its channel slice'
                parts.append(_from_hl32(t, c, ld=l))
            else:
                assert torch.all(t[:, c:] == 7.0), 'wrote outside its channel slice'
                parts.append(t[:, :c])
        return torch.cat(parts, 1).reshape(N, Ho, Wo, Cout).cpu()

    got, unsplit = run(True), run(False)      # (scratch_bytes = 0: the rule cannot split -> the plain launch)
    ref = _ref(x, w, b, stride, pad, dil, act, r_seen, bn).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) < 8e-6 * scale * np.sqrt(K / 64.0 + 1.0) + (2.0 ** -21 * scale if out_fmt else 0.0)
    d = float((got - unsplit).abs().max())
    assert 0.0 < d < 4e-6 * scale, d      # another summation order of the same terms (> 0: the split launch really ran)


def test_conv16x3p_refuses_what_it_cannot_take():
    from gpu_common import dev
    _abi, lib = _lib()
    x = torch.zeros((1, 8, 8, 2 * 64), dtype=torch.float16, device=dev())
    img = torch.zeros((2 * 256 * 64,), dtype=torch.float16, device=dev())
    o = torch.zeros((64, 256), device=dev())
    s = _abi.stream_ptr(dev())
    # K = 64: two K steps (< 4)
    assert lib.emp_conv2d_hl32_f16x3(_abi.ptr(x), 1, 8, 8, 64, 64, _abi.ptr(img), None, None, None, 0, 0, _abi.ptr(o), 256, 0, 256, 1, 1, 1, 0, 1, 0, s) != 0
    # Cout not a multiple of 256
    assert lib.emp_conv2d_hl32_f16x3(_abi.ptr(x), 1, 8, 8, 64, 64, _abi.ptr(img), None, None, None, 0, 0, _abi.ptr(o), 128, 0, 128, 3, 3, 1, 1, 1, 0, s) != 0
    assert lib.emp_x3p_pack_weights(_abi.ptr(o), _abi.ptr(img), 100, 64, s) != 0


# ---- the long-K variants of round 5's kernels, op level ----
EX_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, res, bias_n  (all K >= 1024: the split-role / two-buffer kernels)
    (1, 20, 20, 128, 200, 3, 1, 1, 1, 0, False, False),      # K = 1152, ragged M (400) and Cout (200), no activation
    (1, 20, 20, 128, 200, 3, 1, 1, 1, 1, True, False),       # ReLU + residual
    (2, 12, 12, 128, 256, 3, 1, 2, 2, 2, False, True),       # SiLU + per-image bias, dilation
    (1, 17, 19, 1056, 136, 1, 1, 0, 1, 1, True, True),       # K = 1056 = 33 steps (odd), Cin % 32 == 0, M = 323
    (1, 16, 16, 1040, 128, 1, 1, 0, 1, 1, False, False),     # Cin % 32 != 0: the two-buffer four-wave kernel (K tail of 16)
    (1, 24, 24, 2048, 64, 1, 1, 0, 1, 1, False, False),      # BN = 64 tile, long K
]


@pytest.mark.parametrize('wmode', [0, 1, 2])
@pytest.mark.parametrize('case', EX_CASES)
def test_round5_long_k_kernels_op_level(case, wmode):
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout, k, stride, pad, dil, act, res, use_bn = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g) if res else None
    bn = torch.randn((N, Cout), generator=g) if use_bn else None
    ref = _ref(x, w, b, stride, pad, dil, act, r, bn)
    xd, bd = x.to(dev()), b.to(dev())
    wd = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dev())
    rd = r.to(dev()) if res else None
    bnd = bn.to(dev()) if use_bn else None
    out = torch.full((N, Ho, Wo, Cout + 8), 7.0, device=dev())
    _abi.check(lib.emp_conv2d_nhwc_f16x3_ex(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wd), _abi.ptr(bd), _abi.ptr(bnd) if use_bn else None,
                                            _abi.ptr(rd) if res else None, Cout, _abi.ptr(out), Cout + 8, 0, Cout, k, k, stride, pad, dil, act,
                                            wmode, None, 0, 0, 0, 0, 1, None, None, 0, None, _abi.stream_ptr(dev())), 'conv16x3_ex')
    torch.cuda.synchronize()
    got = out[..., :Cout].cpu().permute(0, 3, 1, 2).double()
    assert torch.all(out[..., Cout:] == 7.0), 'wrote outside its channel slice'
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err < 8e-6 * scale * np.sqrt(Cin * k * k / 64.0 + 1.0), err


KSPLIT_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, res, bias_n, out_fmt | (pixel tiles x cout tiles, steps) -> S x steps per split
    (1, 16, 16, 256, 256, 3, 1, 1, 1, 1, False, False, 0),     # 2 x 2, 72 -> 4 x 18
    (1, 20, 20, 160, 200, 3, 1, 2, 2, 0, True, True, 0),       # 4 x 2, 45 -> 2 x 23 | 22: a split starts in the middle of a tap; ragged M, Cout
    (2, 12, 12, 2048, 256, 3, 1, 4, 4, 1, False, True, 0),     # the ASPP branch: 3 x 2, 576 -> 8 x 72, per-image bias
    (1, 24, 24, 1024, 512, 1, 1, 0, 1, 2, True, False, 0),     # 1x1, 5 x 4, 32 (the shortest K the rule splits) -> 2 x 16, SiLU + residual
    (1, 16, 16, 512, 256, 3, 1, 1, 1, 1, False, False, 1),     # hl32 output through the finish pass
]


@pytest.mark.parametrize('wmode', [0, 1, 2])
@pytest.mark.parametrize('case', KSPLIT_CASES)
def test_split_k_op_level(case, wmode):
    """Round 6 (late): long-K launches of few workgroups run S workgroups per tile over K / S each (conv16x3s_kernel<KSPLIT>,
    raw fp32 partial sums, ksplit_finish32_kernel adds them in ascending order, then bias / bias_n / residual / activation) --
    wmode | 4.  Against fp64 within the unsplit kernels' bound, and against the unsplit launch within fp32 summation order."""
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout, k, stride, pad, dil, act, res, use_bn, ofmt = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g) if res else None
    bn = torch.randn((N, Cout), generator=g) if use_bn else None
    ref = _ref(x, w, b, stride, pad, dil, act, r, bn)
    xd, bd = x.to(dev()), b.to(dev())
    wd = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(dev())
    rd = r.to(dev()) if res else None
    bnd = bn.to(dev()) if use_bn else None
    ld = Cout if ofmt else Cout + 8
    outs = []
    for split in (4, 0):
        out = torch.full((N, Ho, Wo, ld), 7.0, device=dev())
        _abi.check(lib.emp_conv2d_nhwc_f16x3_ex(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wd), _abi.ptr(bd), _abi.ptr(bnd) if use_bn else None,
                                                _abi.ptr(rd) if res else None, Cout, _abi.ptr(out), ld, ofmt, Cout, k, k, stride, pad, dil, act,
                                                wmode | split, None, 0, 0, 0, 0, 1, None, None, 0, None, _abi.stream_ptr(dev())), 'conv16x3_ex')
        torch.cuda.synchronize()
        outs.append(out)
    if ofmt:
        f = [torch.zeros((N * Ho * Wo, Cout), device=dev()) for _ in outs]
        for o, t in zip(outs, f):
            _abi.check(lib.emp_hl32_to_f32(_abi.ptr(o), _abi.ptr(t), N * Ho * Wo, Cout, Cout, Cout, _abi.stream_ptr(dev())), 'hl32_to_f32')
        torch.cuda.synchronize()
        outs = [t.reshape(N, Ho, Wo, Cout) for t in f]
    else:
        assert torch.all(outs[0][..., Cout:] == 7.0), 'wrote outside its channel slice'
    got, unsplit = (o[..., :Cout].cpu().permute(0, 3, 1, 2).double() for o in outs)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 8e-6 * scale * np.sqrt(Cin * k * k / 64.0 + 1.0)
    d = float((got - unsplit).abs().max())
    assert d < 4e-6 * scale, d
    assert d > 0.0, 'the split launch ran unsplit'      # (a different summation order shows in the last bits)


@pytest.mark.parametrize('wmode', [0, 1])
@pytest.mark.parametrize('geom', [(2, 16, 16, 256, 1024, 512, 2), (1, 10, 14, 64, 256, 64, 1), (1, 12, 12, 512, 2048, 1024, 1)])
def test_round5_second_source_op_level(geom, wmode):
    """conv3 + projection shortcut as one K-concatenated convolution (Conv32::in2): relu(W3 . c2 + Wd . x[::s, ::s] + b);
    the third geometry (K = 1536) takes the two-buffer instantiation"""
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout, Cin2, s2 = geom
    g = torch.Generator().manual_seed(11 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g)
    x2 = torch.randn((N, H * s2, W * s2, Cin2), generator=g)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / np.sqrt(Cin)
    w2 = torch.randn((Cout, Cin2, 1, 1), generator=g) / np.sqrt(Cin2)
    b = torch.randn((Cout,), generator=g)
    ref = _ref(x, w, b, 1, 0, 1, 1, None, None, x2, w2, s2)
    wcat = torch.cat([w.reshape(Cout, Cin), w2.reshape(Cout, Cin2)], 1).contiguous().to(dev())
    xd, x2d, bd = x.to(dev()), x2.to(dev()), b.to(dev())
    out = torch.zeros((N, H, W, Cout), device=dev())
    _abi.check(lib.emp_conv2d_nhwc_f16x3_ex(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wcat), _abi.ptr(bd), None, None, 0, _abi.ptr(out), Cout, 0,
                                            Cout, 1, 1, 1, 0, 1, 1, wmode, _abi.ptr(x2d), H * s2, W * s2, Cin2, Cin2, s2, None, None, 0, None,
                                            _abi.stream_ptr(dev())), 'conv16x3_ex in2')
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 8e-6 * scale * np.sqrt((Cin + Cin2) / 64.0 + 1.0)


@pytest.mark.parametrize('head_c,Cout', [(1, 256), (2, 256), (4, 128), (1, 64)])
def test_round5_fused_head_op_level(head_c, Cout):
    """the head's 1x1 inside the pointwise conv's epilogue (heads.py:14): head_w . relu(conv(x) + b) + head_b without the
    Cout-wide map; partial sums per cout tile added in ascending order"""
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin = 2, 18, 22, 256
    g = torch.Generator().manual_seed(5 + head_c)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, 1, 1), generator=g) / np.sqrt(Cin)
    b = torch.randn((Cout,), generator=g)
    hw = torch.randn((head_c, Cout), generator=g) / np.sqrt(Cout)
    hb = torch.randn((head_c,), generator=g)
    mid = _ref(x, w, b, 1, 0, 1, 1)
    ref = (torch.einsum('kc,ncyx->nkyx', hw.double(), mid) + hb.double()[None, :, None, None]).reshape(N, head_c, H * W)
    xd, bd, hwd, hbd = x.to(dev()), b.to(dev()), hw.contiguous().to(dev()), hb.to(dev())
    wd = w.reshape(Cout, Cin).contiguous().to(dev())
    ho = torch.zeros((N, head_c, H * W), device=dev())
    _abi.check(lib.emp_conv2d_nhwc_f16x3_ex(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wd), _abi.ptr(bd), None, None, 0, None, Cout, 0, Cout,
                                            1, 1, 1, 0, 1, 1, 1, None, 0, 0, 0, 0, 1, _abi.ptr(hwd), _abi.ptr(hbd), head_c, _abi.ptr(ho),
                                            _abi.stream_ptr(dev())), 'conv16x3_ex head')
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert float((ho.cpu().double() - ref).abs().max()) < 2e-5 * scale


def test_round5_hl32_output_is_the_split_of_the_fp32_output():
    """the old kernels' hl32 epilogue (the boundary INTO the plane region: layer3.0.conv1, conv3 + shortcut)"""
    from gpu_common import dev
    _abi, lib = _lib()
    N, H, W, Cin, Cout = 1, 20, 12, 512, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn((N, H, W, Cin), generator=g).to(dev())
    w = (torch.randn((Cout, Cin), generator=g) / np.sqrt(Cin)).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    a = torch.zeros((N * H * W, Cout), device=dev())
    h = torch.zeros((N * H * W, 2 * Cout), dtype=torch.float16, device=dev())
    for out, fmt in ((a, 0), (h, 1)):
        _abi.check(lib.emp_conv2d_nhwc_f16x3_ex(_abi.ptr(x), N, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0, _abi.ptr(out), Cout, fmt,
                                                Cout, 1, 1, 1, 0, 1, 1, 1, None, 0, 0, 0, 0, 1, None, None, 0, None, _abi.stream_ptr(dev())), 'ex')
    torch.cuda.synchronize()
    assert torch.equal(h, _to_hl32(a))


# ---- the plane region inside the network (pdl_net.hip run32): layer3 / layer4 / ASPP as hl32 maps ----
@pytest.mark.parametrize('arch,ncls,size', [('pdl', 1, 384), ('bifpn', 1, 256), ('bifpn', 4, 384)])
def test_plane_region_forward_vs_oracle_and_vs_round5_path(arch, ncls, size, monkeypatch):
    """EMP_X3_PLANES_MIN_TILES=1 forces the region on at test sizes (by default it starts at 128 pixel tiles of layer3: 8 tiles of
    1024^2).  Heads within 1e-3 (max norm) of the oracle's fp32 forward; against the same network with EMP_X3_PLANES=0 (round 5's
    kernels on fp32 maps) the heads differ by fp32 rounding only (the residual and the pooled branch see hi + lo = 22 bits), and the
    layer4 output -- read back as hl32 -- is that path's map: the region really runs on the planes."""
    import os
    from gpu_common import dev
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    _abi, lib = _lib()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = dict(weights.MITONET_PDL_CFG if arch == 'pdl' else weights.MITONET_MINI_CFG, num_classes=ncls)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0 if arch == 'pdl' else 3), cfg)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
        w, b = P[name]
        P[name] = (w, b + np.float32(shift))
    img = synth.em_tiles(2, size, seed=17)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    monkeypatch.setenv('EMP_X3_PLANES_MIN_TILES', '1')
    on = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    monkeypatch.setenv('EMP_X3_PLANES', '0')
    off = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    o_on = {k: v.cpu().numpy() for k, v in on(x.cuda(), 2, False).items()}
    o_off = {k: v.cpu().numpy() for k, v in off(x.cuda(), 2, False).items()}
    ref = pdl_model.model_forward(P, x, cfg, 2, False)
    for k in ('ctr_hmp', 'offsets'):
        scale = max(1.0, float(ref[k].pow(2).mean().sqrt()))
        assert float(np.abs(o_on[k] - ref[k].numpy()).max()) / scale < 1e-3, k
        assert float(np.abs(o_on[k] - o_off[k]).max()) / scale < 5e-5, k
    # the layer4 output of the forced network is an hl32 map equal to the other path's fp32 map up to its 22-bit split
    s4 = size // (16 if arch == 'pdl' else 32)
    raw_on = on.tap_raw('encoder.layer4.2', (2 * s4 * s4, 2048))          # 4 bytes per element either way
    raw_off = off.tap_raw('encoder.layer4.2', (2 * s4 * s4, 2048))
    as_f32 = torch.zeros_like(raw_off)
    _abi.check(lib.emp_hl32_to_f32(_abi.ptr(raw_on), _abi.ptr(as_f32), 2 * s4 * s4, 2048, 2048, 2048, _abi.stream_ptr(dev())), 'hl32_to_f32')
    torch.cuda.synchronize()
    scale = float(raw_off.abs().max())
    assert float((as_f32 - raw_off).abs().max()) < 1e-4 * scale
    assert float((raw_on - raw_off).abs().max()) > 1e-2 * scale          # ... and NOT an fp32 map


def test_merged_aspp_equals_the_two_launches(monkeypatch):
    """both decoders' ASPP branch i as ONE conv16x3p launch with two destinations (weights stacked along Cout at finalize):
    the same K order per cout tile -> the heads are bit-identical to the per-decoder launches (EMP_X3_MERGE_ASPP=0)"""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    x = torch.from_numpy(normalize(synth.em_tiles(2, 256, seed=4), 0.57571, 0.12765))[:, None].cuda()
    monkeypatch.setenv('EMP_X3_PLANES_MIN_TILES', '1')
    a = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    monkeypatch.setenv('EMP_X3_MERGE_ASPP', '0')
    b = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    oa, ob = a(x, 2, False), b(x, 2, False)
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    ca = a.tap_raw('instance_decoder.aspp.cat', (2 * 16 * 16, 1024))
    cb = b.tap_raw('instance_decoder.aspp.cat', (2 * 16 * 16, 1024))
    assert torch.equal(ca, cb) and float(ca.abs().max()) > 0


def test_merged_low_level_projections_equal_the_two_launches(monkeypatch):
    """the two decoders' low-level projections (decoders/panoptic_deeplab.py:68-80: a 1x1 conv of the same encoder map each) as ONE
    launch with two destinations (conv16x3.hip store4): bit-identical heads"""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=2), cfg)
    x = torch.from_numpy(normalize(synth.em_tiles(2, 192, seed=9), 0.57571, 0.12765))[:, None].cuda()
    a = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    monkeypatch.setenv('EMP_X3_MERGE_PROJ', '0')
    b = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    oa, ob = a(x, 2, False), b(x, 2, False)
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k


def test_fused_fp32_stem_equals_the_two_launches(monkeypatch):
    """fp16x3 mode: conv1 + bn1 + relu + maxpool (encoders/resnet.py:164-168,219-222) as ONE launch on the matrix pipe with an fp32 tile
    and output (stem.hip stem_pool32_kernel) against the VALU stem + max-pool launches (EMP_X3_FUSE_STEM=0): the pooled map within
    the split product's 2^-21, on a raw uint8 tile with ragged borders (factor_pad fused) and on a float tile"""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize, normalize_params
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=5), cfg)
    img = synth.em_tiles(2, 256, seed=6)[:, :200, :232]
    a = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    monkeypatch.setenv('EMP_X3_FUSE_STEM', '0')
    b = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    raw = torch.from_numpy(np.ascontiguousarray(img))[:, None].cuda()
    flt = torch.from_numpy(normalize(np.ascontiguousarray(img), 0.57571, 0.12765))[:, None].cuda()
    for x, kw in ((raw, dict(sub=float(sub), mul=float(mul))), (flt, {})):
        oa = a(x, 2, False, pad_to=(208, 240), **kw)
        pa = a.tap_raw('p1', (2, 52, 60, 64)).clone()
        ob = b(x, 2, False, pad_to=(208, 240), **kw)
        pb = b.tap_raw('p1', (2, 52, 60, 64))
        scale = float(pb.abs().max())
        assert scale > 0.1 and float((pa - pb).abs().max()) < 4e-6 * scale
        for k in ('ctr_hmp', 'offsets'):
            s = max(1.0, float(ob[k].abs().max()))
            assert float((oa[k] - ob[k]).abs().max()) < 5e-5 * s, k


@pytest.mark.parametrize('env', [{'EMP_X3_FUSE_DS': '0'}, {'EMP_X3_FUSE_HEAD': '0', 'EMP_X3_FUSE_SEP': '0'}, {'EMP_X3_FUSE_SEP': '0'},
                                 {'EMP_X3_MERGE_ASPP': '0', 'EMP_X3_MERGE_PROJ': '0'}, {'EMP_X3_FUSE_STEM': '0', 'EMP_X3P_KGROUP': '64'},
                                 {'EMP_X3_SPEC': '0', 'EMP_X3_PLANES': '0'}, {'EMP_X3_WIMG': '0', 'EMP_X3_PLANES': '0'},
                                 {'EMP_X3_PLANES': '0'}, {'EMP_X3_PLANES': '0', 'EMP_X3_KSPLIT': '0'},      # (small batches: split-K on round 5's kernels / not)
                                 {'EMP_X3_PLANES_MIN_TILES': '100000'}, {'EMP_X3_PLANES_MIN_TILES': '100000', 'EMP_X3_SMALL_ASPP': '0'}])      # (below the threshold: ASPP merged + K-split on the plane kernel / not)
def test_every_ab_switch_of_the_mode_stays_within_the_gate(env, monkeypatch):
    """the A/B switches of the fp16x3 mode select other kernels / launch groupings for the same arithmetic:
    with the plane region forced on at test size, every combination keeps the heads within 1e-3 (max norm) of the oracle's fp32
    forward and within fp32 rounding of the default configuration"""
    import os
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
        w, b = P[name]
        P[name] = (w, b + np.float32(shift))
    x = torch.from_numpy(normalize(synth.em_tiles(2, 256, seed=13), 0.57571, 0.12765))[:, None]
    monkeypatch.setenv('EMP_X3_PLANES_MIN_TILES', '1')
    base = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16x3')
    ob = {k: v.cpu() for k, v in base(x.cuda(), 2, False).items()}
    oa = {k: v.cpu() for k, v in alt(x.cuda(), 2, False).items()}
    ref = pdl_model.pdl_forward(P, x, cfg, 2, False)
    for k in ('ctr_hmp', 'offsets'):
        scale = max(1.0, float(ref[k].pow(2).mean().sqrt()))
        assert float((oa[k] - ref[k]).abs().max()) / scale < 1e-3, (env, k)
        assert float((oa[k] - ob[k]).abs().max()) / scale < 5e-5, (env, k)
