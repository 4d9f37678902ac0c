"""The library's assignment solvers (csrc/matcher.hip lsa_maximize, exported as emp_lsa_maximize, and its sparse form
lsa_maximize_sparse = emp_lsa_maximize_sparse, the one the slice matcher calls) against
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


def _ours_sparse(m):
    """the matcher's solver: the same algorithm fed with the non-zero entries only (emp_lsa_maximize_sparse)"""
    lib = _abi.load()
    m = np.asarray(m, dtype=np.float64)
    nr, nc = m.shape
    er, ec = np.nonzero(m)
    ew = np.ascontiguousarray(m[er, ec], dtype=np.float64)
    er, ec = np.ascontiguousarray(er, dtype=np.int64), np.ascontiguousarray(ec, dtype=np.int64)
    k = min(nr, nc)
    rows, cols = np.empty(k, np.int64), np.empty(k, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    _abi.check(lib.emp_lsa_maximize_sparse(nr, nc, len(er), vp(er), vp(ec), vp(ew), vp(rows), vp(cols)), 'emp_lsa_maximize_sparse')
    return rows, cols


def _check(m):
    r, c = linear_sum_assignment(m, maximize=True)
    rr, cc = _ours(m)
    assert np.array_equal(r, rr) and np.array_equal(c, cc), f'\n{m}\nscipy {list(zip(r, c))}\nours  {list(zip(rr, cc))}'
    rr, cc = _ours_sparse(m)
    assert np.array_equal(r, rr) and np.array_equal(c, cc), f'\n{m}\nscipy {list(zip(r, c))}\nsparse {list(zip(rr, cc))}'


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


def test_sparse_solver_on_block_structured_matrices():
    """IoU-like matrices (small dense blocks, permuted, padded with empty rows / columns; tie-heavy and continuous values,
    both orientations) up to 300 x 400: the sparse solver returns scipy's assignment, pair for pair"""
    rng = np.random.default_rng(21)
    for it in range(600):
        k = rng.integers(3, 60)
        shapes = [(rng.integers(1, 5), rng.integers(1, 5)) for _ in range(k)]
        nr, nc = sum(a for a, _ in shapes) + rng.integers(0, 30), sum(b for _, b in shapes) + rng.integers(0, 80)
        M = np.zeros((nr, nc))
        r0 = c0 = 0
        for a, b in shapes:
            M[r0:r0 + a, c0:c0 + b] = rng.choice([0.0, 0.25, 0.5, 0.5, 1 / 3], size=(a, b)) if it % 2 else \
                rng.random((a, b)) * (rng.random((a, b)) < 0.8)
            r0, c0 = r0 + a, c0 + b
        M = M[rng.permutation(nr)][:, rng.permutation(nc)]
        if it % 5 == 0:
            M = M.T.copy()
        r, c = linear_sum_assignment(M, maximize=True)
        rr, cc = _ours_sparse(M)
        assert np.array_equal(r, rr) and np.array_equal(c, cc), (it, M.shape)


def test_sparse_solver_does_not_scale_with_the_empty_part():
    """7 600 objects with two overlaps each (a 4096^2 slice): the dense algorithm needs ~0.3 s for this matrix, the sparse
    form a few milliseconds -- asserted loosely (< 0.2 s) so that a regression to a dense scan is caught"""
    import time
    rng = np.random.default_rng(5)
    nr, nc = 7600, 8300
    er = np.repeat(np.arange(nr), 2)
    ec = np.clip(er + rng.integers(-1, 2, len(er)), 0, nc - 1)
    _, idx = np.unique(er * nc + ec, return_index=True)
    er, ec = np.ascontiguousarray(er[idx], dtype=np.int64), np.ascontiguousarray(ec[idx], dtype=np.int64)
    ew = rng.random(len(er))
    lib = _abi.load()
    rows, cols = np.empty(nr, np.int64), np.empty(nr, np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    _abi.check(lib.emp_lsa_maximize_sparse(nr, nc, len(er), vp(er), vp(ec), vp(ew), vp(rows), vp(cols)), 'sparse')
    dt = time.perf_counter() - t0
    assert dt < 0.2, dt
    assert len(set(cols.tolist())) == nr and np.array_equal(rows, np.arange(nr))
    val = {(int(a), int(b)): w for a, b, w in zip(er, ec, ew)}
    got = sum(val.get((int(a), int(b)), 0.0) for a, b in zip(rows, cols))
    # the optimum of this banded matrix from scipy on its 60 x 66 leading block is not comparable; check optimality by duality
    # on a sample instead: no single swap of two rows' columns improves the value
    for _ in range(2000):
        i, j = rng.integers(0, nr, 2)
        cur = val.get((int(i), int(cols[i])), 0.0) + val.get((int(j), int(cols[j])), 0.0)
        alt = val.get((int(i), int(cols[j])), 0.0) + val.get((int(j), int(cols[i])), 0.0)
        assert alt <= cur + 1e-12
    assert got > 0


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


def _tie_stack(seed, n_slices=6, W=48):
    """slices of grid-aligned 8 x 8 squares and 8 x 16 / 16 x 8 bars, shifted by half a square from slice to slice: many
    exactly equal IoUs (1/3, 1/7, 3/5 ...) and competing overlaps in every step"""
    rng = np.random.default_rng(seed)
    stack = []
    for z in range(n_slices):
        seg, lab, occ = {}, 1001, np.zeros((W, W), bool)
        shift = 4 * (z % 2)
        for gy in range(0, W - 16, 8):
            for gx in range(0, W - 16, 8):
                if rng.random() < 0.45:
                    h, w = ((8, 8), (8, 16), (16, 8))[rng.integers(0, 3)]
                    y0, x0 = gy + shift, gx + shift
                    if not occ[y0:y0 + h, x0:x0 + w].any():
                        occ[y0:y0 + h, x0:x0 + w] = True
                        seg[lab] = _square(y0, x0, h, w, W)
                        lab += 1
        stack.append(seg)
    return stack


@pytest.mark.parametrize('seed', range(12))
def test_tie_heavy_stacks_match_the_full_matrix_reference(seed, monkeypatch):
    """whole stacks with exactly tied IoUs at every step: the C++ matcher (which hands the solver the full matrix, as
    matcher.py:216-218 does) labels every slice like the Python RLEMatcher with scipy, forward and backward"""
    stack = _tie_stack(seed)
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
        assert list(got) == list(want), (seed, 'forward', i)
        for k in want:
            np.testing.assert_array_equal(got[k]['starts'], want[k]['starts'])
    sm.begin_backward()
    sm.run_range(0, len(stack) - 1, -1)
    m.target_rle, m.assign_new = None, False
    for i in range(len(stack) - 1, -1, -1):
        want = fwd[i] if m.target_rle is None and not m.initialize_target(fwd[i]) else m(fwd[i])
        got = sm.slice_objects(i)
        assert list(got) == list(want), (seed, 'backward', i)


def test_restricted_block_is_not_the_full_matrix_under_ties():
    """why the full matrix is the default: on tie-heavy block-structured matrices the optimum restricted to the conflict
    components (round 2's solver block) is a DIFFERENT optimal assignment than scipy's on the full matrix in a sizeable
    fraction of cases -- equal value, other pairs"""
    rng = np.random.default_rng(1)
    differ = total = 0
    for _ in range(400):
        k = rng.integers(2, 6)
        shapes = [(rng.integers(1, 4), rng.integers(1, 4)) for _ in range(k)]
        nr, nc = sum(a for a, _ in shapes), sum(b for _, b in shapes)
        M = np.zeros((nr, nc))
        r0 = c0 = 0
        for a, b in shapes:
            M[r0:r0 + a, c0:c0 + b] = rng.choice([0.0, 0.25, 0.5, 0.5, 1 / 3], size=(a, b))
            r0, c0 = r0 + a, c0 + b
        M = M[rng.permutation(nr)][:, rng.permutation(nc)]
        R, Cc = linear_sum_assignment(M, maximize=True)
        full = {(int(r), int(c)) for r, c in zip(R, Cc) if M[r, c] > 0}
        rows = [r for r in range(nr) if M[r].any()]
        cols = [c for c in range(nc) if M[:, c].any()]
        sub = M[np.ix_(rows, cols)]
        rr, cc = _ours(sub)
        part = {(rows[i], cols[j]) for i, j in zip(rr, cc) if sub[i, j] > 0}
        assert abs(sum(M[p] for p in part) - sum(M[p] for p in full)) < 1e-12        # the optimum VALUE is the same
        total += 1
        differ += part != full
    assert differ > 0.02 * total, (differ, total)
