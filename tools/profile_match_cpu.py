"""Host matcher / tracker stage timing on a dumped run-list stack (tools/profile_match.py with EMP_DUMP_RUNS=<file>):
python tools/profile_match_cpu.py gpurun_out/match_runs.npz [ranks]   -- no GPU needed (the matcher is host code)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import sparse  # noqa: E402

d = np.load(sys.argv[1])
off, S, n = int(d['off']), int(d['S']), d['n']
runs = d['runs'].astype(np.int64)
ends = np.cumsum(n)
stack = [runs[e - k:e] for e, k in zip(ends, n)]
D = len(stack)
for rep in range(3):
    sm = sparse.StackMatcher(1, 10000, 0.25, 0.25, match=True)
    t0 = time.perf_counter()
    for r in stack:
        sm.push_runs(r, S, off)
    t1 = time.perf_counter()
    sm.prepare()
    t2 = time.perf_counter()
    sm.run_range(0, D - 1, +1)
    t3 = time.perf_counter()
    sm.begin_backward()
    sm.run_range(0, D - 1, -1)
    t4 = time.perf_counter()
    inst = sm.track_range('xy', (D, S, S), 0, D - 1, 0)
    t5 = time.perf_counter()
    print(f'push {1e3*(t1-t0):6.1f}  pairs {1e3*(t2-t1):6.1f}  forward {1e3*(t3-t2):6.1f}  backward {1e3*(t4-t3):6.1f}  '
          f'track+export {1e3*(t5-t4):6.1f} ms   ({D} slices, {len(inst)} tracks)', flush=True)
