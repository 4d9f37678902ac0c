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
# write rate against the share of reads in the stream (finding 10 in DESIGN.md): pure fill, broadcast copies that read
# 1/8, 1/4, 1/2 as many bytes as they write, plain copy (1:1), add (2:1)
for k in (8, 4, 2):
    src = a[:n // k]
    t = timeit(lambda: b.view(k, n // k).copy_(src.unsqueeze(0).expand(k, n // k)))
    print(f'broadcast copy read 1/{k} : {t:.3f} ms  write {n*2/t/1e9:.2f} TB/s, read {n*2/k/t/1e9:.2f} TB/s')
