"""Run the fused separable conv a few times (for rocprofv3 --pmc): python tools/one_sepconv.py H W C Cout head_c batch"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import _abi
lib = _abi.load()
H, W, Cc, Cout, hc, B = [int(a) for a in sys.argv[1:7]]
dev = torch.device('cuda:0')
st = _abi.stream_ptr(dev)
x = torch.randn((B, H, W, Cc), device=dev).to(torch.float16)
dw = (torch.randn((25, Cc), device=dev) * 0.2).to(torch.float16)
pw = (torch.randn((Cout, Cc), device=dev) / np.sqrt(Cc)).to(torch.float16)
pwp = torch.empty_like(pw)
_abi.check(lib.emp_sepconv5x5_pack_pw(_abi.ptr(pw), Cc, Cc, Cout, _abi.ptr(pwp), st), 'pack')
b = torch.randn((Cout,), device=dev)
out = torch.empty((B, H, W, Cout), device=dev, dtype=torch.float16)
hw = torch.randn((max(hc, 1), Cout), device=dev)
hb = torch.randn((max(hc, 1),), device=dev)
ho = torch.empty((B, max(hc, 1), H, W), device=dev)
for _ in range(3):
    _abi.check(lib.emp_sepconv5x5_nhwc_f16(_abi.ptr(x), B, H, W, Cc, Cc, _abi.ptr(dw), _abi.ptr(pwp), _abi.ptr(b), Cout, 1,
                                           None if hc else _abi.ptr(out), Cout, _abi.ptr(hw) if hc else None,
                                           _abi.ptr(hb) if hc else None, hc, _abi.ptr(ho), st), 'fused')
torch.cuda.synchronize()

