"""Round 6: the depthwise-separable block of the fp16x3 mode as ONE launch (csrc/sepconv_x3.hip) -- fp32 NHWC
maps in and out, the depthwise half in fp32 on the vector pipe, its result split into an fp16 pair in LDS, three fp16 MFMAs per
pointwise product -- against an fp64 evaluation of the reference's block (empanada/models/blocks.py:15-33: depthwise KxK, no bias
-> pointwise 1x1 -> folded BatchNorm -> ReLU / SiLU; heads.py:12-15 adds the final 1x1)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, C, Cout, ks, act, head_c
    (2, 24, 40, 288, 256, 5, 1, 0),      # the Panoptic-DeepLab fuse block (256 + 32 channels = 9 chunks), ragged tiles
    (1, 16, 32, 256, 256, 5, 1, 1),      # a head: the 256-channel map is never stored
    (2, 19, 21, 256, 256, 5, 1, 2),      # offsets head, odd sizes
    (1, 32, 32, 256, 128, 5, 1, 0),      # BiFPN decoder fusion (2F -> F)
    (3, 16, 16, 128, 128, 3, 2, 0),      # BiFPN node: 3x3, SiLU
    (1, 24, 24, 128, 128, 5, 1, 1),      # BiFPN head
    (1, 8, 16, 64, 128, 5, 0, 0),        # one tile, two chunks, no activation
]


@pytest.mark.parametrize('case', CASES)
def test_sepconv_x3_equals_fp64(case):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, C, Cout, ks, act, hc = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, C), generator=g)
    dw = torch.randn((C, 1, ks, ks), generator=g) / ks
    pw = torch.randn((Cout, C), generator=g) / np.sqrt(C)
    b = torch.randn((Cout,), generator=g)
    hw = torch.randn((max(hc, 1), Cout), generator=g) / np.sqrt(Cout)
    hb = torch.randn((max(hc, 1),), generator=g)
    mid = F.conv2d(x.permute(0, 3, 1, 2).double(), dw.double(), None, 1, ks // 2, 1, C)
    y = F.conv2d(mid, pw.double()[:, :, None, None], b.double())
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = y * torch.sigmoid(y)
    st = _abi.stream_ptr(dev())
    xd = x.to(dev())
    dwd = dw.reshape(C, ks * ks).t().contiguous().to(dev())      # (k*k, C)
    pwd, bd = pw.contiguous().to(dev()), b.to(dev())
    dwp = torch.zeros((ks * ks * C,), device=dev())
    pwp = torch.zeros((2 * C * Cout,), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_sepconv_x3_pack(_abi.ptr(dwd), _abi.ptr(pwd), ks, C, Cout, _abi.ptr(dwp), _abi.ptr(pwp), st), 'pack')
    if hc:
        want = torch.einsum('kc,ncyx->nkyx', hw[:hc].double(), y) + hb[:hc].double()[None, :, None, None]
        hwd, hbd = hw[:hc].contiguous().to(dev()), hb[:hc].contiguous().to(dev())
        out = torch.zeros((N, hc, H * W), device=dev())
        _abi.check(lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(xd), N, H, W, C, C, _abi.ptr(dwp), _abi.ptr(pwp), _abi.ptr(bd), Cout, act, None, 0,
                                               _abi.ptr(hwd), _abi.ptr(hbd), hc, _abi.ptr(out), ks, st), 'sepconv_x3 head')
        torch.cuda.synchronize()
        got = out.reshape(N, hc, H, W).cpu().double()
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) < 2e-5 * scale
    else:
        out = torch.full((N, H, W, Cout + 4), 7.0, device=dev())      # a channel slice of a wider buffer
        _abi.check(lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(xd), N, H, W, C, C, _abi.ptr(dwp), _abi.ptr(pwp), _abi.ptr(bd), Cout, act, _abi.ptr(out),
                                               Cout + 4, None, None, 0, None, ks, st), 'sepconv_x3')
        torch.cuda.synchronize()
        assert torch.all(out[..., Cout:] == 7.0), 'wrote outside its channel slice'
        got = out[..., :Cout].cpu().permute(0, 3, 1, 2).double()
        scale = float(y.abs().max())
        err = float((got - y).abs().max())
        assert err < 8e-6 * scale * np.sqrt(C / 64.0 + 1.0) + 2e-6 * scale, err
        # ... and it is not the fp16 product: the depthwise result rounded to fp16 once would leave ~3e-4 of the scale
        y16 = F.conv2d(mid.half().double(), pw.half().double()[:, :, None, None], b.double())
        if act == 0:
            assert float((y16 - y).abs().max()) > 20 * err


def test_sepconv_x3_batch_equals_single_images():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, C, Cout, ks = 3, 16, 32, 128, 128, 5
    g = torch.Generator().manual_seed(2)
    x = torch.randn((N, H, W, C), generator=g).to(dev())
    dw = (torch.randn((ks * ks, C), generator=g) / ks).to(dev())
    pw = (torch.randn((Cout, C), generator=g) / np.sqrt(C)).to(dev())
    b = torch.randn((Cout,), generator=g).to(dev())
    st = _abi.stream_ptr(dev())
    dwp = torch.zeros((ks * ks * C,), device=dev())
    pwp = torch.zeros((2 * C * Cout,), dtype=torch.float16, device=dev())
    _abi.check(lib.emp_sepconv_x3_pack(_abi.ptr(dw), _abi.ptr(pw), ks, C, Cout, _abi.ptr(dwp), _abi.ptr(pwp), st), 'pack')
    full = torch.zeros((N, H, W, Cout), device=dev())
    _abi.check(lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(x), N, H, W, C, C, _abi.ptr(dwp), _abi.ptr(pwp), _abi.ptr(b), Cout, 1, _abi.ptr(full), Cout,
                                           None, None, 0, None, ks, st), 'batch')
    for i in range(N):
        one = torch.zeros((1, H, W, Cout), device=dev())
        xi = x[i:i + 1].contiguous()
        _abi.check(lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(xi), 1, H, W, C, C, _abi.ptr(dwp), _abi.ptr(pwp), _abi.ptr(b), Cout, 1, _abi.ptr(one), Cout,
                                               None, None, 0, None, ks, st), 'one')
        torch.cuda.synchronize()
        assert torch.equal(one[0], full[i])


def test_sepconv_x3_refuses_unsupported_shapes():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    z = torch.zeros((4096,), device=dev())
    st = _abi.stream_ptr(dev())
    # C not a multiple of 32; Cout 192; head with a 3x3 block
    assert lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(z), 1, 8, 16, 48, 48, _abi.ptr(z), _abi.ptr(z), None, 128, 1, _abi.ptr(z), 128, None, None, 0, None, 5, st) != 0
    assert lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(z), 1, 8, 16, 64, 64, _abi.ptr(z), _abi.ptr(z), None, 192, 1, _abi.ptr(z), 192, None, None, 0, None, 5, st) != 0
    assert lib.emp_sepconv_x3_nhwc_f32(_abi.ptr(z), 1, 8, 16, 64, 64, _abi.ptr(z), _abi.ptr(z), None, 128, 1, None, 0, _abi.ptr(z), _abi.ptr(z), 1, _abi.ptr(z), 3, st) != 0
