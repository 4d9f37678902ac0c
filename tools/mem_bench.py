"""Achievable HBM rates on this box (torch copy / fill) next to the short-K conv shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import _abi
lib = _abi.load()
dev = torch.device('cuda:0')

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

n = 512 * 1024 * 1024  # fp16 elements = 1 GiB
a = torch.empty(n, dtype=torch.float16, device=dev)
b = torch.empty(n, dtype=torch.float16, device=dev)
t = timeit(lambda: b.copy_(a)); print(f'copy 1GiB->1GiB : {t:.3f} ms  {2*n*2/t/1e9:.2f} TB/s (r+w)')
t = timeit(lambda: b.zero_()); print(f'fill 1GiB       : {t:.3f} ms  {n*2/t/1e9:.2f} TB/s (w)')
t = timeit(lambda: a.sum()); print(f'reduce 1GiB     : {t:.3f} ms  {n*2/t/1e9:.2f} TB/s (r)')
c = torch.empty(n // 4, dtype=torch.float16, device=dev)
t = timeit(lambda: torch.add(a[:n // 4], c, out=b[:n // 4])); print(f'add 0.25GiB x2->1: {t:.3f} ms  {3*(n//4)*2/t/1e9:.2f} TB/s')
