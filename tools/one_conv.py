"""Run one conv shape a few times (for rocprofv3 --pmc): python tools/one_conv.py H W Cin Cout k dil variant batch [packed]
(packed = 1: the 256x256 tile with its packed weight image, emp_conv256_pack_weights + variant bit 20)"""
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
if len(sys.argv) > 9 and int(sys.argv[9]):
    wp = torch.empty_like(w)
    _abi.check(lib.emp_conv256_pack_weights(_abi.ptr(w), _abi.ptr(wp), Cout, k * k, Cin, 0, _abi.stream_ptr(dev)), 'pack')
    w, v = wp, v | (1 << 20)
for _ in range(3):
    _abi.check(lib.emp_conv2d_nhwc_f16(_abi.ptr(x), B, H, W, Cin, Cin, _abi.ptr(w), _abi.ptr(b), None, None, 0,
                                       _abi.ptr(out), Cout, Cout, k, k, 1, p, d, 1, v, _abi.stream_ptr(dev)), 'conv')
torch.cuda.synchronize()
