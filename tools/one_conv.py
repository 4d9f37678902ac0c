"""Run one conv shape a few times (for rocprofv3 --pmc): python tools/one_conv.py H W Cin Cout k dil variant batch"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import _abi
lib = _abi.load()
H, W, Cin, Cout, k, d, v, B = [int(a) for a in sys.argv[1:9]]
p = d * (k - 1) // 2
dev = torch.device('cuda:0')
x = torch.randn((B, H, W, Cin), device=dev).to(torch.float16)
w = (torch.randn((Cout, k * k, Cin), device=dev) / np.sqrt(Cin * k * k)).to(torch.float16)
b = torch.randn((Cout,), device=dev)
out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
for _ in range(3):
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                       _abi.ptr(out), Cout, Cout, k, k, 1, p, d, 1, v, _abi.stream_ptr(dev)), 'conv')
torch.cuda.synchronize()
