"""The assignment step of the slice matcher on label maps whose objects touch a few neighbours (discs on a jittered grid,
a fraction of them split in two from slice to slice -> competing overlaps in every step), at growing slice sizes: the
dense solve of the whole IoU matrix (what the reference hands to scipy, EMP_SM_FULL_LSA=1) against the default (the same
algorithm on the matrix's non-zero entries, csrc/matcher.hip lsa_maximize_sparse: identical assignment).
    python tools/lsa_scaling.py [sizes...]      (host code only: no GPU needed)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import sparse  # noqa: E402


def slice_runs(S, z, rng, pitch=52):
    """(n,3) {start, length, component id} raster-ordered runs of one slice: discs on a jittered grid, radius breathing with z,
    every 5th disc cut in two halves (two components) on odd slices"""
    lab = np.zeros((S, S), dtype=np.int32)
    yy, xx = np.mgrid[0:pitch, 0:pitch]
    k = 1
    for gy in range(0, S - pitch, pitch):
        for gx in range(0, S - pitch, pitch):
            cy, cx = pitch / 2 + 3 * np.sin(0.3 * z + gy), pitch / 2 + 3 * np.cos(0.2 * z + gx)
            r = 14 + 5 * np.sin(0.25 * z + 0.01 * (gy + gx))
            m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
            blk = lab[gy:gy + pitch, gx:gx + pitch]
            if (k % 5 == 0) and (z % 2 == 1):
                blk[m & (xx < cx - 1)] = k
                blk[m & (xx > cx + 1)] = k + 1
                k += 2
            else:
                blk[m] = k
                k += 1
    flat = lab.ravel()
    edge = np.flatnonzero(np.diff(flat, prepend=0, append=0) != 0)
    starts, ends = edge[:-1], edge[1:]
    keep = flat[starts] > 0
    starts, ends = starts[keep], ends[keep]
    return np.stack([starts, ends - starts, flat[starts]], 1).astype(np.int64), k - 1


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]
    for S in sizes:
        rng = np.random.default_rng(S)
        stack, nobj = [], 0
        for z in range(12):
            r, k = slice_runs(S, z, rng)
            stack.append(r)
            nobj = max(nobj, k)
        out = {}
        for mode, name in (('1', 'whole matrix'), (None, 'sparse solver')):
            if mode is None:
                os.environ.pop('EMP_SM_FULL_LSA', None)
            else:
                os.environ['EMP_SM_FULL_LSA'] = mode
            best = 1e9
            for rep in range(2):
                sm = sparse.StackMatcher(1, 100000, 0.25, 0.25, match=True)
                for r in stack:
                    sm.push_runs(r, S, 0)
                sm.prepare()
                t0 = time.perf_counter()
                sm.run_range(0, len(stack) - 1, +1)
                best = min(best, time.perf_counter() - t0)
            sm.begin_backward()
            sm.run_range(0, len(stack) - 1, -1)
            inst = sm.track_range('xy', (len(stack), S, S), 0, len(stack) - 1, 0)
            out[name] = inst
            print(f'{S:5d}^2 slices, {nobj:5d} objects: forward pass {1e3 * best / (len(stack) - 1):8.3f} ms per slice  [{name}; '
                  f'steps sparse / dense: {sm.solver_stats()}]', flush=True)
        a, b = out.values()
        same = list(a) == list(b) and all(np.array_equal(a[k]['starts'], b[k]['starts']) and np.array_equal(a[k]['runs'], b[k]['runs']) for k in a)
        print(f'        identical trackers: {same}')


if __name__ == '__main__':
    main()
