"""Host half of the 3-D stitching path (C++ range algebra in libempanada_hip.so + the Python
matcher / tracker / consensus glue of empanada-napari_amd/sparse.py) vs the oracle and the
reference goldens.  No GPU needed: these entry points are host functions."""
import os

import numpy as np
import pytest

from empanada_napari_amd import sparse as ps
from oracle import sparse as osp

import sparse_case


def _rand_rle(rng, n, hi=4000):
    s = np.sort(rng.choice(hi, size=n, replace=False)).astype(np.int64)
    r = rng.integers(1, 8, size=n).astype(np.int64)
    e = np.minimum(s + r, np.append(s[1:], 10 ** 9))
    return s, e - s


def test_pair_intersections_match_oracle():
    rng = np.random.default_rng(0)
    objs = [_rand_rle(rng, int(rng.integers(1, 200))) for _ in range(12)]
    pairs = [(a, b) for a in range(12) for b in range(12) if a != b]
    got = ps.rle_pair_intersections(objs, pairs)
    want = [osp.rle_intersection(objs[a][0], objs[a][1], objs[b][0], objs[b][1]) for a, b in pairs]
    np.testing.assert_array_equal(got, np.array(want))
    a, b = objs[0], objs[1]
    assert ps.rle_iou(*a, *b) == osp.rle_iou(*a, *b)
    assert ps.rle_ioa(*a, *b) == osp.rle_ioa(*a, *b)


@pytest.mark.parametrize('thr', [1, 2, 3])
def test_vote_matches_oracle(thr):
    rng = np.random.default_rng(thr)
    for trial in range(20):
        lists = []
        for _ in range(int(rng.integers(3, 5))):
            s, r = _rand_rle(rng, int(rng.integers(2, 60)), hi=600)
            lists.append(np.stack([s, s + r], axis=1))
        got = ps.vote_by_ranges([l.copy() for l in lists], thr)
        want = np.asarray(osp.vote_by_ranges([l.copy() for l in lists], thr)).reshape(-1, 2)
        np.testing.assert_array_equal(np.asarray(got).reshape(-1, 2), want)


def test_vote_edge_cases():
    assert len(ps.vote_by_ranges([], 2)) == 0
    one = [np.array([[0, 5], [7, 9]])]
    assert len(ps.vote_by_ranges(one, 2)) == 0                       # fewer lists than votes (array_utils.py:635-639)
    touching = [np.array([[0, 10]]), np.array([[10, 20]])]
    assert ps.join_ranges(touching).tolist() == [[0, 20]]            # border ranges join (reference test case)
    assert ps.vote_by_ranges([np.array([[10, 20]]), np.array([[7, 26]])], 2).tolist() == [[10, 20]]


def test_reference_golden_ranges(golden_dir):
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    for t in range(6):
        lists = [g[f'rng{t}_in{j}'] for j in range(int(g[f'rng{t}_n']))]
        for thr in (1, 2, 3):
            got = np.asarray(ps.vote_by_ranges(lists, thr)).reshape(-1, 2)
            np.testing.assert_array_equal(got, g[f'rng{t}_vote{thr}'])
        a, b = lists[0], lists[1]
        assert ps.rle_intersection(a[:, 0], a[:, 1] - a[:, 0], b[:, 0], b[:, 1] - b[:, 0]) == int(g[f'rng{t}_inter'])
        ms, mr = ps.merge_rles(a[:, 0], a[:, 1] - a[:, 0], b[:, 0], b[:, 1] - b[:, 0])
        np.testing.assert_array_equal(np.stack([ms, mr], axis=1), g[f'rng{t}_merge'])


class _Impl:
    RLEMatcher = ps.RLEMatcher
    InstanceTracker = ps.InstanceTracker


def _flat(inst):
    keys = np.array([int(k) for k in inst], dtype=np.int64)
    boxes = np.array([list(inst[k]['box']) for k in inst], dtype=np.int64).reshape(len(keys), -1)
    off = np.cumsum([0] + [len(inst[k]['starts']) for k in inst]).astype(np.int64)
    cat = lambda key: np.concatenate([np.asarray(inst[k][key], dtype=np.int64) for k in inst]) if len(keys) else np.zeros(0, np.int64)
    return {'keys': keys, 'boxes': boxes, 'off': off, 'starts': cat('starts'), 'runs': cat('runs')}


@pytest.fixture(scope='module')
def trackers():
    # dense -> RLE through the oracle here (the GPU version of that step is tested in test_gpu_sparse.py)
    return sparse_case.run_axis_pipeline(_Impl, to_rle=osp.pan_seg_to_rle_seg)


def test_matcher_tracker_match_reference(golden_dir, trackers):
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    for tr in trackers:
        for k, v in _flat(tr.instances).items():
            np.testing.assert_array_equal(v, g[f'trk_{tr.axis}_{k}'], err_msg=f'{tr.axis} {k}')


@pytest.mark.parametrize('thr,ciou,bypass', [(2, 0.75, False), (1, 0.75, True), (3, 0.5, False)])
def test_consensus_matches_reference(golden_dir, trackers, thr, ciou, bypass):
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    inst = ps.merge_objects_from_trackers(trackers, thr, ciou, bypass)
    for k, v in _flat(inst).items():
        np.testing.assert_array_equal(v, g[f'cons_{thr}_{int(bypass)}_{k}'], err_msg=k)


def test_filters_and_relabel(trackers):
    import copy
    tr = copy.deepcopy(trackers[0])
    ot = osp.InstanceTracker(1, 1000, tr.shape3d, 'xy')
    ot.instances = copy.deepcopy(tr.instances)
    ps.remove_small_objects(tr, 300); osp.remove_small_objects(ot, 300)
    ps.remove_pancakes(tr, 6); osp.remove_pancakes(ot, 6)
    assert list(tr.instances) == list(ot.instances) and len(tr.instances) > 0
    a, b = ps.instance_relabel(tr), osp.instance_relabel(ot)
    assert list(a) == list(b)
    for k in a:
        np.testing.assert_array_equal(a[k]['starts'], b[k]['starts'])
        np.testing.assert_array_equal(a[k]['runs'], b[k]['runs'])


def _stack_matcher_trackers(push):
    """The sparse_case pipeline through the C++ stack matcher (csrc/matcher.hip)."""
    vol = sparse_case.synth_label_volume(sparse_case.SHAPE, 7, seed=5)
    out = []
    for axis, name in enumerate(('xy', 'xz', 'yz')):
        slices = sparse_case.axis_pan_slices(vol, axis, sparse_case.DIVISOR, seed=100 + axis)
        sm = ps.StackMatcher(1, sparse_case.DIVISOR, 0.25, 0.25)
        for pan in slices:
            push(sm, pan)
        sm.forward()
        fwd = [sm.slice_objects(i) for i in range(len(slices))]
        inst = sm.backward_and_track(name, sparse_case.SHAPE)
        out.append((name, slices, fwd, inst, sm))
    return out


def _push_objects(sm, pan):
    sm.push_objects(osp.pan_seg_to_rle_seg(pan, [1], sparse_case.DIVISOR, [1], force_connected=True)[1])


def _push_runs(sm, pan):
    """raw (start, length, label) triples in raster order, as the GPU run extractor emits them"""
    cc = osp.connected_components(np.where((pan >= sparse_case.DIVISOR) & (pan < 2 * sparse_case.DIVISOR), pan, 0))
    flat = cc.ravel()
    W = pan.shape[1]
    runs = []
    for y in range(pan.shape[0]):
        row = flat[y * W:(y + 1) * W]
        x = 0
        while x < W:
            if row[x] > 0:
                x0 = x
                while x < W and row[x] == row[x0]:
                    x += 1
                runs.append((y * W + x0, x - x0, int(row[x0])))
            else:
                x += 1
    sm.push_runs(np.array(runs, dtype=np.int64).reshape(-1, 3), W, sparse_case.DIVISOR)


@pytest.mark.parametrize('push', [_push_objects, _push_runs])
def test_cpp_stack_matcher_matches_reference(golden_dir, push):
    g = np.load(os.path.join(golden_dir, 'sparse.npz'))
    for name, slices, fwd, inst, sm in _stack_matcher_trackers(push):
        for k, v in _flat(inst).items():
            np.testing.assert_array_equal(v, g[f'trk_{name}_{k}'], err_msg=f'{name} {k}')


def test_cpp_stack_matcher_equals_python_matcher_per_slice():
    """every slice after the forward pass and after the backward pass: same labels, boxes and runs as RLEMatcher"""
    for name, slices, fwd, inst, sm in _stack_matcher_trackers(_push_objects):
        m = ps.RLEMatcher(1, sparse_case.DIVISOR, 0.25, 0.25)
        stack = []
        for i, pan in enumerate(slices):
            seg = osp.pan_seg_to_rle_seg(pan, [1], sparse_case.DIVISOR, [1], force_connected=True)[1]
            seg = seg if m.target_rle is None and not m.initialize_target(seg) else m(seg)
            stack.append(seg)
            assert list(seg) == list(fwd[i]), (name, i)
            for k in seg:
                assert tuple(seg[k]['box']) == fwd[i][k]['box']
                np.testing.assert_array_equal(seg[k]['starts'], fwd[i][k]['starts'])
                np.testing.assert_array_equal(seg[k]['runs'], fwd[i][k]['runs'])
        m.target_rle, m.assign_new = None, False
        for i in range(len(slices) - 1, -1, -1):
            seg = stack[i] if m.target_rle is None and not m.initialize_target(stack[i]) else m(stack[i])
            got = sm.slice_objects(i)
            assert list(seg) == list(got), (name, i)
            for k in seg:
                np.testing.assert_array_equal(seg[k]['starts'], got[k]['starts'])


def test_cpp_stack_matcher_empty_slices_and_semantic_class():
    sm = ps.StackMatcher(1, 1000, 0.25, 0.25)
    a = {1001: {'box': (0, 0, 2, 4), 'starts': np.array([0, 8]), 'runs': np.array([4, 4])}}
    for seg in ({}, a, {}, a):
        sm.push_objects(seg)
    sm.forward()
    assert [list(sm.slice_objects(i)) for i in range(4)] == [[], [1001], [], [1002]]      # new label after an empty slice
    inst = sm.backward_and_track('xy', (4, 2, 8))
    assert sorted(inst) == [1001, 1002] and inst[1001]['box'] == (1, 0, 0, 2, 2, 4)
    st = ps.StackMatcher(2, 1000, match=False)
    st.push_objects({2000: {'box': (0, 0, 1, 3), 'starts': np.array([1]), 'runs': np.array([2])}})
    st.push_objects({2000: {'box': (1, 0, 2, 2), 'starts': np.array([8]), 'runs': np.array([2])}})
    st.forward()
    inst = st.backward_and_track('xy', (2, 2, 8))
    assert list(inst) == [2000] and inst[2000]['starts'].tolist() == [24, 1] and inst[2000]['box'] == (0, 0, 0, 2, 2, 3)


@pytest.mark.parametrize('shape,density', [((5, 7, 128), 0.6), ((3, 4, 64), 1.0), ((4, 6, 70), 0.9), ((2, 3, 200), 0.5),
                                           ((6, 5, 1), 0.7)])
def test_cpp_yz_tracker_bitmap_equals_sort_and_encode(shape, density):
    """yz stacks: the C++ tracker turns each object around through a bitmap over its box; the reference decodes to
    voxels, sorts and re-encodes (tracker.py:84-88,111-120).  Dense / full-width masks make runs continue across
    rows of the raveled volume, widths of 64k exercise the open-run-at-row-end case, 2-D runs wrap across plane rows."""
    rng = np.random.default_rng(hash(shape) % 1000)
    mask = rng.random(shape) < density
    D, H, W = shape
    sm = ps.StackMatcher(2, 1000, match=False)
    ot = osp.InstanceTracker(2, 1000, shape, 'yz')
    for x in range(W):
        flat = np.flatnonzero(mask[:, :, x].ravel())
        seg = {}
        if len(flat):
            st, rn = osp.rle_encode(flat)
            zz, yy = np.unravel_index(flat, (D, H))
            seg = {2000: {'box': (int(zz.min()), int(yy.min()), int(zz.max()) + 1, int(yy.max()) + 1), 'starts': st, 'runs': rn}}
        sm.push_objects(seg)
        ot.update(seg, x)
    ot.finish()
    sm.forward()
    inst = sm.backward_and_track('yz', shape)
    assert list(inst) == list(ot.instances) == [2000]
    np.testing.assert_array_equal(inst[2000]['starts'], ot.instances[2000]['starts'])
    np.testing.assert_array_equal(inst[2000]['runs'], ot.instances[2000]['runs'])
    assert tuple(inst[2000]['box']) == tuple(ot.instances[2000]['box'])
    np.testing.assert_array_equal(np.sort(osp.rle_decode(inst[2000]['starts'], inst[2000]['runs'])), np.flatnonzero(mask.ravel()))


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_fill_holes_slices_equals_reference_loop(seed):
    """emp_fill_holes_slices (C++ host, threaded over slices) vs the reference's loop restated with scipy's
    binary_fill_holes (filters.py:178-210), including its overwrite of foreign labels inside a bounding box."""
    import ctypes as C
    from empanada_napari_amd import _abi
    from scipy.ndimage import binary_fill_holes
    rng = np.random.default_rng(seed)
    D, H, W = 5, 40, 48
    vol = np.zeros((D, H, W), np.uint32)
    yy, xx = np.mgrid[0:H, 0:W]
    for z in range(D):
        for lab in rng.permutation(np.arange(1001, 1009)):
            cy, cx, r = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(3, 12)
            d = np.hypot(yy - cy, xx - cx)
            vol[z][(d < r) & (d > r * rng.uniform(0.2, 0.7))] = lab      # rings: holes, overlaps, border contacts
        vol[z][rng.random((H, W)) < 0.05] = 0
    want = vol.copy()
    for z in range(D):
        m = want[z]
        boxes = {}
        for lab in np.unique(m[m > 0]):
            ys, xs = np.nonzero(m == lab)
            boxes[int(lab)] = (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1)
        for lab in sorted(boxes):
            y0, x0, y1, x1 = boxes[lab]
            m[y0:y1, x0:x1] = binary_fill_holes(m[y0:y1, x0:x1].astype(bool)).astype(m.dtype) * lab
    got = vol.copy()
    _abi.check(_abi.load().emp_fill_holes_slices(got.ctypes.data_as(C.c_void_p), D, H, W), 'emp_fill_holes_slices')
    assert (want != vol).any()
    np.testing.assert_array_equal(got, want)


def test_oracle_morphology_restatement_basics():
    """oracle erode / dilate / label_nd on hand-checkable input (filters.py:14-20,154-176)"""
    vol = np.zeros((5, 7, 7), np.uint32)
    vol[1:4, 2:5, 2:5] = 1003          # a 3x3x3 cube ...
    vol[0, 0, 0] = 1001                # ... and a corner voxel (border mode: reflect keeps it under dilation only)
    tr = osp.InstanceTracker(1, 1000, vol.shape, 'xy')
    tr.instances = osp.filters_pan_seg_to_rle_seg(vol, [1], 1000, [1])
    assert sorted(tr.instances) == [1001, 1002] and tr.instances[1002]['box'] == (1, 2, 2, 4, 5, 5)
    osp.erode(tr, vol.shape, [1], 1000, [1], 1)
    assert list(tr.instances) == [1001] and tr.instances[1001]['starts'].tolist() == [2 * 49 + 3 * 7 + 3]   # cube centre
    osp.dilate(tr, vol.shape, [1], 1000, [1], 1)
    assert int(tr.instances[1001]['runs'].sum()) == 7                                                       # a 3-D cross
    diag = np.zeros((2, 2, 2), np.int64); diag[0, 0, 0] = diag[1, 1, 1] = 5
    assert osp.label_nd(diag).max() == 1                                                                    # 26-connectivity


def test_cpp_stack_matcher_solver_only_where_needed(monkeypatch):
    """emp_sm_run: slices whose IoU matrix has at most one non-zero per row and column are assigned directly (every such
    pair is in any optimal assignment); the others need an assignment solver -- since round 3 the library's own
    (lsa_maximize_sparse: scipy's algorithm restated on the non-zero entries, tests/test_lsa.py); with EMP_SM_SCIPY=1 the
    whole dense IoU matrix goes to scipy itself, the reference's call (matcher.py:216-218).  Both kinds of slices occur in
    the reference case whose trackers are compared with the reference's goldens above; both solvers give the same
    trackers."""
    calls = []
    orig = ps.StackMatcher._solve_pending
    monkeypatch.setattr(ps.StackMatcher, '_solve_pending', lambda self: (calls.append(1), orig(self))[1])
    res = {}
    for flag in ('0', '1'):
        monkeypatch.setenv('EMP_SM_SCIPY', flag)
        calls.clear()
        n_slices, res[flag] = 0, []
        for name, slices, fwd, inst, sm in _stack_matcher_trackers(_push_objects):
            n_slices += 2 * len(slices)
            res[flag].append(inst)
        if flag == '0':
            assert len(calls) == 0, 'the default path must not come back to Python for the assignment'
        else:
            assert 0 < len(calls) < n_slices // 2, (len(calls), n_slices)
    for a, b in zip(res['0'], res['1']):
        assert list(a) == list(b)
        for k in a:
            assert a[k]['box'] == b[k]['box']
            np.testing.assert_array_equal(a[k]['starts'], b[k]['starts'])
            np.testing.assert_array_equal(a[k]['runs'], b[k]['runs'])


def test_vote_with_many_unsorted_stretches_uses_radix_sort_and_matches_oracle():
    """tracker run lists arrive slice block by slice block in descending order (hundreds of ascending stretches): the
    C++ vote then radix-sorts starts and ends; same ranges as the oracle's restatement of vote_by_ranges"""
    rng = np.random.default_rng(0)
    lists = []
    for t in range(3):
        st = np.sort(rng.choice(1_500_000, 30000, replace=False))
        ln = rng.integers(1, 20, len(st))
        keep = np.r_[True, st[1:] >= st[:-1] + ln[:-1]]
        st, ln = st[keep], ln[keep]
        idx = np.concatenate(np.array_split(np.arange(len(st)), 300)[::-1])
        lists.append(np.stack([st[idx], st[idx] + ln[idx]], axis=1))
    for thr in (1, 2, 3):
        np.testing.assert_array_equal(ps.vote_by_ranges(lists, thr), osp.vote_by_ranges([l.copy() for l in lists], thr))
