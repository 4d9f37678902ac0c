"""Three back-to-back launches of the fp16x3 convolution on one shape (for rocprofv3 --pmc passes):
python tools/one_conv16x3.py H W Cin Cout k dil B"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import _abi  # noqa: E402

H, W, Cin, Cout, k, d, B = [int(a) for a in sys.argv[1:8]]
lib = _abi.load()
dev = torch.device('cuda:0')
p = d * (k // 2)
x = torch.relu(torch.randn((B, H, W, Cin), device=dev))
w = torch.randn((Cout, k * k, Cin), device=dev) / np.sqrt(Cin * k * k)
b = torch.randn((Cout,), device=dev)
out = torch.empty((B, H, W, Cout), device=dev)
for _ in range(3):
    _abi.check(lib.emp_conv2d_nhwc_f16x3(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0, _abi.ptr(out), Cout, Cout,
                                         k, k, 1, p, d, 1, 1, 0, _abi.stream_ptr(dev)), 'x3')
torch.cuda.synchronize()
