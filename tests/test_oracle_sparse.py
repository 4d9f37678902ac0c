"""Oracle of the sparse label algebra (oracle/sparse.py) vs
  * the reference's own unit tests (tests/test_array_utils.py, test_zarr_utils.py -- same cases,
    same expectations, restated here), and
  * golden vectors produced by the imported reference (tests/golden/sparse.npz)."""
import os

import numpy as np
import pytest

from oracle import sparse as osp


# ---- the reference's unit-test cases (reference tests/test_array_utils.py:7-155) ----
def test_ref_box_iou_cases():
    r = osp.box_iou_pairs(np.array([[0, 0, 20, 20]]), np.array([[5, 5, 25, 25]]))
    assert r[0] == [0] and r[1] == [0] and r[2][0] == pytest.approx(0.39, abs=0.02) and r[3][0] == 225
    r = osp.box_iou_pairs(np.array([[0, 0, 20, 20]]), np.array([[30, 0, 50, 20]]))
    assert r == ([], [], [], [])


def test_ref_intersection_from_ranges_cases():
    assert osp.intersection_from_ranges(np.array([[0, 10], [7, 20]]), np.array([True])) == 3
    assert osp.intersection_from_ranges(np.array([[0, 10], [7, 20]]), np.array([False])) == 0


def test_ref_split_range_by_votes_cases():
    a = osp.split_range_by_votes(np.array([0, 10]), np.array([2, 3, 3, 3, 1, 2, 2, 3, 3, 4]), 2)
    assert a.tolist() == [[0, 4], [5, 10]]
    b = osp.split_range_by_votes(np.array([0, 10]), np.array([2, 3, 3, 3, 2, 2, 2, 3, 3, 4]), 2)
    assert b.tolist() == [[0, 10]]


def test_ref_extend_range_case():
    r, v = osp.extend_range(np.array([1, 10]), np.array([3, 10]), np.array([2, 4, 4, 4, 4, 2, 2, 2, 2, 2]))
    assert list(r) == [1, 10] and list(v) == [2, 4, 5, 5, 5, 3, 3, 3, 3, 3]


def test_ref_rle_voting_case():
    assert osp.rle_voting(np.array([(10, 20), (7, 26)])).tolist() == [[10, 20], [23, 26]]


@pytest.mark.parametrize('ranges,expected', [
    ([(0, 10), (6, 10)], [[0, 10]]), ([(0, 10), (11, 20)], [[0, 10], [11, 20]]), ([(0, 10), (10, 20)], [[0, 20]])])
def test_ref_join_ranges_cases(ranges, expected):
    assert osp._join_ranges(np.array(ranges)).tolist() == expected


def test_ref_invert_ranges_case():
    assert osp.invert_ranges(np.array([(2, 6), (4, 12)]), 15).tolist() == [[0, 2], [6, 4], [12, 15]]


def test_ref_zarr_utils_cases():
    got = osp.chunk_ranges(np.array([[0, 20], [15, 35]]), 7, 6)
    assert got == [[0, 6], [6, 7], [7, 13], [13, 14], [14, 20], [15, 20], [20, 21], [21, 27], [27, 28], [28, 34], [34, 35]]
    assert osp.fill_func(np.array([0, 0, 0, 0, 0]), np.array([(2, 5), (10, 13), (15, 18)]), 7).tolist() == [0, 0, 7, 7, 7]


# ---- golden vectors from the imported reference ----
@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'sparse.npz'))


@pytest.mark.parametrize('t', range(6))
def test_range_primitives_match_reference(g, t):
    lists = [g[f'rng{t}_in{j}'] for j in range(int(g[f'rng{t}_n']))]
    for thr in (1, 2, 3):
        got = np.asarray(osp.vote_by_ranges([l.copy() for l in lists], thr)).reshape(-1, 2)
        np.testing.assert_array_equal(got, g[f'rng{t}_vote{thr}'])
    a, b = lists[0], lists[1]
    ra, rb = a[:, 1] - a[:, 0], b[:, 1] - b[:, 0]
    assert osp.rle_intersection(a[:, 0], ra, b[:, 0], rb) == int(g[f'rng{t}_inter'])
    assert osp.rle_iou(a[:, 0], ra, b[:, 0], rb) == float(g[f'rng{t}_iou'])
    ms, mr = osp.merge_rles(a[:, 0].copy(), ra.copy(), b[:, 0].copy(), rb.copy())
    np.testing.assert_array_equal(np.stack([ms, mr], axis=1), g[f'rng{t}_merge'])
    inv = osp.invert_ranges(osp.join_ranges([a.copy(), b.copy()]), 500)
    np.testing.assert_array_equal(inv, g[f'rng{t}_invert'])


def _flatten(inst):
    keys = np.array([int(k) for k in inst], dtype=np.int64)
    boxes = np.array([list(inst[k]['box']) for k in inst], dtype=np.int64).reshape(len(keys), -1)
    off = np.cumsum([0] + [len(inst[k]['starts']) for k in inst]).astype(np.int64)
    cat = lambda key: np.concatenate([np.asarray(inst[k][key], dtype=np.int64) for k in inst]) if len(keys) else np.zeros(0, np.int64)
    return {'keys': keys, 'boxes': boxes, 'off': off, 'starts': cat('starts'), 'runs': cat('runs')}


def _check(g, prefix, inst):
    f = _flatten(inst)
    for k, v in f.items():
        np.testing.assert_array_equal(v, g[f'{prefix}_{k}'], err_msg=f'{prefix}_{k}')


@pytest.fixture(scope='module')
def trackers(g):
    """Same pipeline as oracle/gen_golden.py::gen_sparse, with the oracle's matcher / tracker."""
    import importlib.util, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gen_golden_helpers', os.path.join(root, 'tests', 'sparse_case.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.run_axis_pipeline(osp)


def test_matcher_tracker_match_reference(g, trackers):
    for tr in trackers:
        _check(g, f'trk_{tr.axis}', tr.instances)
        assert len(tr.instances) > 0


@pytest.mark.parametrize('thr,ciou,bypass', [(2, 0.75, False), (1, 0.75, True), (3, 0.5, False)])
def test_consensus_matches_reference(g, trackers, thr, ciou, bypass):
    inst = osp.merge_objects_from_trackers(trackers, thr, ciou, bypass)
    _check(g, f'cons_{thr}_{int(bypass)}', inst)


def test_consensus_is_nontrivial(g):
    assert len(g['cons_2_0_keys']) >= 3 and len(g['cons_1_1_keys']) >= 3 and len(g['cons_3_0_keys']) >= 1


def test_semantic_consensus_matches_reference(g, trackers):
    shape = tuple(int(v) for v in g['volume_shape'])
    sem = []
    for tr in trackers:
        t2 = osp.InstanceTracker(2, 1000, shape, tr.axis)
        allr = osp.join_ranges([np.stack([a['starts'], a['starts'] + a['runs']], axis=1) for a in tr.instances.values()])
        t2.instances = {2000: {'box': (0, 0, 0) + shape, 'starts': allr[:, 0], 'runs': allr[:, 1] - allr[:, 0]}}
        sem.append(t2)
    _check(g, 'semcons', osp.merge_semantic_from_trackers(sem, 2))


def test_dense_rle_roundtrip_and_cc():
    """PARITY UNPINNED piece: self-consistency of connected components / RLE conversion."""
    rng = np.random.default_rng(3)
    pan = np.zeros((40, 50), np.int64)
    pan[5:15, 5:20] = 1001
    pan[10:30, 25:30] = 1001          # second piece with the same id -> split by force_connected
    pan[29:35, 30:40] = 1001          # touches the second piece diagonally at (29,30)-(29,29): 8-connected
    pan[0:3, 40:50] = 1002
    seg = osp.pan_seg_to_rle_seg(pan, [1], 1000, [1], force_connected=True)
    assert sorted(seg[1].keys()) == [1001, 1002, 1003]     # raster order of first pixel: (0,40), (5,5), (10,25)
    back = osp.rle_seg_to_pan_seg(seg, pan.shape)
    assert np.array_equal(back > 0, pan > 0)
    seg2 = osp.pan_seg_to_rle_seg(pan, [1], 1000, [1], force_connected=False)
    assert np.array_equal(osp.rle_seg_to_pan_seg(seg2, pan.shape), pan.astype(np.uint32))
    for a in seg[1].values():
        ys, xs = np.unravel_index(osp.rle_decode(a['starts'], a['runs']), pan.shape)
        assert a['box'] == (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1)
