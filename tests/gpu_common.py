"""Helpers shared by the -m gpu tests (HIP path vs oracle on identical inputs)."""
import ctypes as C

import numpy as np
import torch

from empanada_napari_amd import _abi


def dev():
    return torch.device('cuda:0')


def conv_hip(x_nhwc, w_oihw, bias=None, bias_n=None, res=None, stride=1, pad=0, dil=1, relu=False, variant=0,
             out_ld=None, out_coff=0):
    """x (N,H,W,Cin) fp16 cuda; w (Cout,Cin,KH,KW) fp32 -> out (N,Ho,Wo,Cout) fp16 through the C ABI."""
    lib = _abi.load()
    N, H, W, Cin = x_nhwc.shape
    Cout, _, KH, KW = w_oihw.shape
    cin_pad = (Cin + 63) // 64 * 64
    xin = torch.zeros((N, H, W, cin_pad), dtype=torch.float16, device=dev())
    xin[..., :Cin] = x_nhwc
    wp = torch.zeros((Cout, KH * KW, cin_pad), dtype=torch.float16, device=dev())
    wp[..., :Cin] = w_oihw.permute(0, 2, 3, 1).reshape(Cout, KH * KW, Cin).to(torch.float16)
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    ld = out_ld or Cout
    out = torch.full((N, Ho, Wo, ld), 7.0, dtype=torch.float16, device=dev())
    b = None if bias is None else bias.float().contiguous().to(dev())
    bn = None if bias_n is None else bias_n.float().contiguous().to(dev())
    r = None if res is None else res.contiguous()
    optr = C.c_void_p(out.data_ptr() + 2 * out_coff)
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(xin), N, H, W, cin_pad, cin_pad, _abi.ptr(wp), _abi.ptr(b),
                                       _abi.ptr(bn), _abi.ptr(r), 0 if r is None else r.shape[-1], optr, ld, Cout,
                                       KH, KW, stride, pad, dil, int(relu), variant, _abi.stream_ptr(dev())),
               'emp_conv2d_nhwc_f16')
    torch.cuda.synchronize()
    return out


def conv_ref(x_nhwc, w_oihw, bias=None, bias_n=None, res=None, stride=1, pad=0, dil=1, relu=False):
    """fp32 CPU reference on the same fp16-rounded operands."""
    import torch.nn.functional as F
    x = x_nhwc.float().cpu().permute(0, 3, 1, 2)
    w = w_oihw.to(torch.float16).float().cpu()
    y = F.conv2d(x, w, None if bias is None else bias.float().cpu(), stride, pad, dil)
    if bias_n is not None:
        y = y + bias_n.float().cpu()[:, :, None, None]
    if res is not None:
        y = y + res.float().cpu().permute(0, 3, 1, 2)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1).contiguous()
