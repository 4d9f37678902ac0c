"""The library's assignment solver (csrc/matcher.hip lsa_maximize, exported as emp_lsa_maximize) against
scipy.optimize.linear_sum_assignment(maximize=True) -- the call of the reference's matcher (empanada/inference/matcher.py:218)
-- INCLUDING which optimum is returned for tied matrices: label maps must be bit-identical, and with competing equal IoUs
the choice decides which slice object inherits which label.  Matrices: continuous random, small integers (heavy ties),
sparse IoU-like (mostly zeros), constant, rectangular both ways, degenerate shapes; plus the ADVICE r02 tie cases (one
target split into two equal halves, nt > nm and nt < nm) through the stack matcher with the solver block vs scipy on the
FULL IoU matrix."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

from empanada_napari_amd import _abi
from empanada_napari_amd import sparse as ps


def _ours(m):
    lib = _abi.load()
    m = np.ascontiguousarray(m, dtype=np.float64)
    nr, nc = m.shape
    k = min(nr, nc)
    rows, cols = np.empty(k, np.int64), np.empty(k, np.int64)
    _abi.check(lib.emp_lsa_maximize(m.ctypes.data_as(C.c_void_p), nr, nc, rows.ctypes.data_as(C.c_void_p),
                                    cols.ctypes.data_as(C.c_void_p)), 'emp_lsa_maximize')
    return rows, cols


def _check(m):
    r, c = linear_sum_assignment(m, maximize=True)
    rr, cc = _ours(m)
    assert np.array_equal(r, rr) and np.array_equal(c, cc), f'\n{m}\nscipy {list(zip(r, c))}\nours  {list(zip(rr, cc))}'


@pytest.mark.parametrize('kind', ['uniform', 'ints', 'sparse', 'iou'])
def test_matches_scipy_including_ties(kind):
    rng = np.random.default_rng({'uniform': 1, 'ints': 2, 'sparse': 3, 'iou': 4}[kind])
    for _ in range(3000):
        nr, nc = rng.integers(1, 9, 2)
        if kind == 'uniform':
            m = rng.random((nr, nc))
        elif kind == 'ints':
            m = rng.integers(0, 3, (nr, nc)).astype(np.float64)
        elif kind == 'sparse':
            m = rng.random((nr, nc)) * (rng.random((nr, nc)) < 0.35)
        else:       # ratios of small integers: exactly tied IoUs of equal splits
            m = rng.integers(0, 4, (nr, nc)) / rng.integers(4, 7, (nr, nc)) * (rng.random((nr, nc)) < 0.6)
        _check(m)


def test_larger_and_degenerate_shapes():
    rng = np.random.default_rng(9)
    for nr, nc in ((40, 40), (25, 60), (60, 25), (1, 30), (30, 1), (64, 64)):
        _check(rng.random((nr, nc)))
        _check(rng.integers(0, 2, (nr, nc)).astype(np.float64))
        _check(np.zeros((nr, nc)))
        _check(np.full((nr, nc), 0.5))
    assert _ours(np.zeros((0, 5)))[0].size == 0 and _ours(np.zeros((4, 0)))[0].size == 0


def _square(y0, x0, h, w, W):
    starts = np.array([(y0 + r) * W + x0 for r in range(h)], dtype=np.int64)
    return {'box': (y0, x0, y0 + h, x0 + w), 'starts': starts, 'runs': np.full(h, w, dtype=np.int64)}


@pytest.mark.parametrize('case', ['split_in_two', 'two_into_one', 'three_way', 'grid'])
def test_tied_ious_match_the_reference_on_the_full_matrix(case):
    """the C++ matcher hands only the conflict block of the IoU matrix to the solver while the reference solves
    the full nt x nm matrix (matcher.py:216-218); under exact IoU ties the tie-break could depend on the matrix shape.
    Deliberately tied slices, several target / match counts: the stack matcher's labels equal the Python RLEMatcher's
    (= the reference's algorithm on the full matrix with scipy), forward and backward."""
    W = 40
    if case == 'split_in_two':          # one target, two matches with EQUAL overlap and area (nt < nm)
        a = {1001: _square(4, 4, 8, 16, W), 1002: _square(20, 2, 6, 6, W)}
        b = {1001: _square(4, 4, 8, 8, W), 1002: _square(4, 12, 8, 8, W), 1003: _square(30, 30, 4, 4, W)}
    elif case == 'two_into_one':        # two equal targets, one match that overlaps both equally (nt > nm)
        a = {1001: _square(4, 4, 8, 8, W), 1002: _square(4, 12, 8, 8, W), 1003: _square(20, 2, 6, 6, W), 1004: _square(30, 2, 4, 4, W)}
        b = {1001: _square(4, 4, 8, 16, W)}
    elif case == 'three_way':           # a chain of equal overlaps: 3 targets, 2 matches astride them, every IoU equal
        a = {1001: _square(2, 2, 6, 12, W), 1002: _square(2, 14, 6, 12, W), 1003: _square(2, 26, 6, 12, W)}
        b = {1001: _square(2, 8, 6, 12, W), 1002: _square(2, 20, 6, 12, W), 1003: _square(12, 2, 4, 4, W)}
    else:                               # a 2 x 2 block of identical squares shifted by half a square: every overlap equal
        a = {1001 + i: _square(4 + 8 * (i // 2), 4 + 8 * (i % 2), 8, 8, W) for i in range(4)}
        b = {1001 + i: _square(8 + 8 * (i // 2), 8 + 8 * (i % 2), 8, 8, W) for i in range(4)}
    stack = [a, b, a, b]
    sm = ps.StackMatcher(1, 1000, 0.25, 0.25)
    for seg in stack:
        sm.push_objects(seg)
    sm.forward()
    m = ps.RLEMatcher(1, 1000, 0.25, 0.25)
    fwd = []
    for seg in stack:
        seg = {k: dict(v) for k, v in seg.items()}
        fwd.append(seg if m.target_rle is None and not m.initialize_target(seg) else m(seg))
    for i, want in enumerate(fwd):
        got = sm.slice_objects(i)
        assert list(got) == list(want), (case, 'forward', i, list(got), list(want))
    inst = sm.backward_and_track('xy', (len(stack), 40, W))
    m.target_rle, m.assign_new = None, False
    tr = ps.InstanceTracker(1, 1000, (len(stack), 40, W), 'xy')
    for i in range(len(stack) - 1, -1, -1):
        seg = fwd[i] if m.target_rle is None and not m.initialize_target(fwd[i]) else m(fwd[i])
        tr.update(seg, i)
    tr.finish()
    assert list(inst) == list(tr.instances)
    for k in inst:
        np.testing.assert_array_equal(inst[k]['starts'], tr.instances[k]['starts'])
