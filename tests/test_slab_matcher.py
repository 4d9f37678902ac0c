"""Slab-wise matching (multigpu.SlabMatcher, csrc/matcher.hip emp_sm_prepare / export_state / import_state) against the
sequential passes over the whole stack (sparse.StackMatcher, itself pinned by the reference goldens in
tests/test_host_sparse.py): every rank matches its own slab between two ghost slices, the forward state ripples down
the ranks, the backward state back up, every rank tracks its own slices and the partial trackers are concatenated --
bit-identical trackers for xy / xz / yz stacks, several classes (two thing classes and a semantic one), fragmented
objects (IoA merges, competing overlaps -> assignment solver), empty slices on slab boundaries, one-slice slabs.

The ranks are threads here and the messages go through in-process mailboxes (the gloo path itself is covered by
tests/test_multigpu_cpu.py through the public API); no GPU is touched: the matcher is the host half of the library."""
import queue
import threading

import numpy as np
import pytest

import sparse_case
from empanada_napari_amd import multigpu
from empanada_napari_amd import sparse as ps
from oracle import sparse as osp

DIV = 1000
LABELS, THINGS = [1, 2, 3], [1, 2]


def _stack(shape, axis, seed, empty=()):
    """per-slice entries {class: instance dict} of a synthetic three-class stack along ``axis``"""
    vols = [sparse_case.synth_label_volume(shape, n, seed=seed + 7 * c) for c, n in ((1, 22), (2, 9))]
    rng = np.random.default_rng(seed)
    out = []
    for i in range(shape[axis]):
        pan = np.zeros([s for a, s in enumerate(shape) if a != axis], dtype=np.int64)
        for c, vol in zip((1, 2), vols):
            sl = np.take(vol, i, axis=axis).copy()
            sl[rng.random(sl.shape) < 0.06] = 0                 # fragments: false splits, IoA merges, competing overlaps
            free = pan == 0
            pan[free & (sl > 0)] = c * DIV + sl[free & (sl > 0)]
        sem = (np.take(vols[0], i, axis=axis) == 0) & (rng.random(pan.shape) < 0.2) & (pan == 0)
        pan[sem] = 3 * DIV                                       # a semantic (stuff) class: tracked, never matched
        if i in empty:
            pan[:] = 0
        out.append(osp.pan_seg_to_rle_seg(pan, LABELS, DIV, THINGS, force_connected=True))
    return out


def _sequential(entries, axis_name, shape):
    out = {}
    for c in LABELS:
        sm = ps.StackMatcher(c, DIV, 0.25, 0.25, match=c in THINGS)
        for e in entries:
            sm.push_objects(e[c])
        sm.forward()
        out[c] = sm.backward_and_track(axis_name, shape)
    return out


class _Mailboxes:
    """send_object_list / recv_object_list between threads: one queue per (src, dst)"""

    def __init__(self):
        self.q = {}
        self.lock = threading.Lock()
        self.rank = threading.local()

    def box(self, src, dst):
        with self.lock:
            return self.q.setdefault((src, dst), queue.Queue())

    def send(self, obj, dst, group):
        self.box(self.rank.value, dst).put(obj)

    def recv(self, src, group):
        return self.box(src, self.rank.value).get(timeout=60)


def _as_runs(entries):
    """the entries in the GPU extractor's form: per class ((n,3) {start, length, component id} in raster order, id offset)
    -- the matcher then keeps the raster order and tracks a slice in one pass (csrc/matcher.hip track_raster)"""
    out = []
    for e in entries:
        d = {}
        for c, inst in e.items():
            if c not in THINGS:          # a semantic class is one object labelled c * DIV: stays an instance dict
                d[c] = inst
                continue
            rows = [np.stack([a['starts'], a['runs'], np.full(len(a['starts']), lab - c * DIV)], 1) for lab, a in inst.items()]
            r = np.concatenate(rows) if rows else np.zeros((0, 3), np.int64)
            d[c] = (np.ascontiguousarray(r[np.argsort(r[:, 0], kind='stable')].astype(np.int64)), c * DIV)
        out.append(d)
    return out


def _distributed(entries, axis_name, shape, world, monkeypatch, group_size=5, as_runs=False):
    n = len(entries)
    if as_runs:
        entries = _as_runs(entries)
    width = [s for a, s in enumerate(shape) if a != ps.InstanceTracker.AXES[axis_name]][1]
    bounds = [b for b in multigpu.slab_bounds(n, world) if b[1] > b[0]]
    aw = len(bounds)
    mail = _Mailboxes()
    monkeypatch.setattr(multigpu, '_send_obj', mail.send)
    monkeypatch.setattr(multigpu, '_recv_obj', mail.recv)
    parts, errs = [None] * aw, []

    def rank_main(r):
        try:
            mail.rank.value = r
            lo, hi = bounds[r]
            sm = multigpu.SlabMatcher(LABELS, THINGS, DIV, 0.25, 0.25, width, head=as_runs and r == 0)
            for i0 in range(lo, hi, group_size):                 # pushed group by group, as the GPU extractor delivers
                sm.push(entries[i0:min(hi, i0 + group_size)])
            parts[r] = sm.finish(r, aw, lo, axis_name, shape, None)
        except Exception as e:       # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(aw)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    assert all(p is not None for p in parts), 'a rank did not finish (deadlock?)'
    return multigpu.merge_partial_trackers(parts, axis_name)


def _assert_same(got, want):
    assert list(got) == list(want)
    for c in want:
        assert [int(k) for k in got[c]] == [int(k) for k in want[c]], f'class {c}: label order differs'
        for k in want[c]:
            assert tuple(int(v) for v in got[c][k]['box']) == tuple(int(v) for v in want[c][k]['box']), (c, k)
            np.testing.assert_array_equal(got[c][k]['starts'], want[c][k]['starts'], err_msg=f'{c} {k} starts')
            np.testing.assert_array_equal(got[c][k]['runs'], want[c][k]['runs'], err_msg=f'{c} {k} runs')


SHAPE = (26, 40, 44)


@pytest.mark.parametrize('axis_name', ['xy', 'xz', 'yz'])
@pytest.mark.parametrize('world', [2, 3, 5])
def test_slab_matching_equals_sequential_passes(axis_name, world, monkeypatch):
    axis = ps.InstanceTracker.AXES[axis_name]
    entries = _stack(SHAPE, axis, seed=11 + axis)
    want = _sequential(entries, axis_name, SHAPE)
    assert sum(len(v) for v in want.values()) > 20
    _assert_same(_distributed(entries, axis_name, SHAPE, world, monkeypatch), want)


@pytest.mark.parametrize('axis_name', ['xy', 'xz', 'yz'])
@pytest.mark.parametrize('world', [1, 3])
def test_extractor_form_entries_and_head_slab_chain(axis_name, world, monkeypatch):
    """the same stacks pushed the way the GPU extractor delivers them (raster-ordered run triples): the one-pass tracker and
    the first slab's forward chain running group by group behind the pushes give the trackers of the sequential passes"""
    axis = ps.InstanceTracker.AXES[axis_name]
    entries = _stack(SHAPE, axis, seed=31 + axis)
    want = _sequential(entries, axis_name, SHAPE)
    _assert_same(_distributed(entries, axis_name, SHAPE, world, monkeypatch, group_size=4, as_runs=True), want)


def test_empty_boundary_slices_and_one_slice_slabs(monkeypatch):
    n = SHAPE[0]
    b3 = multigpu.slab_bounds(n, 3)
    empty = {b3[0][1] - 1, b3[0][1], b3[1][1], 0, n - 1}         # last / first slices of slabs, both ends of the stack
    entries = _stack(SHAPE, 0, seed=3, empty=empty)
    want = _sequential(entries, 'xy', SHAPE)
    _assert_same(_distributed(entries, 'xy', SHAPE, 3, monkeypatch), want)
    # as many ranks as slices: every slab is a single slice between two ghosts
    short = entries[:7]
    shape7 = (7,) + SHAPE[1:]
    _assert_same(_distributed(short, 'xy', shape7, 7, monkeypatch, group_size=1), _sequential(short, 'xy', shape7))


def test_state_roundtrip_and_import_checks():
    entries = _stack((6, 24, 24), 0, seed=5)
    sm = ps.StackMatcher(1, DIV, 0.25, 0.25)
    for e in entries:
        sm.push_objects(e[1])
    sm.prepare()
    sm.forward()
    labels, off, mem, nl = sm.export_state(3)
    assert len(off) == len(labels) + 1 and off[-1] == len(mem) and nl > DIV
    assert [int(k) for k in sm.slice_objects(3)] == labels.tolist()
    twin = ps.StackMatcher(1, DIV, 0.25, 0.25)
    twin.push_objects(entries[3][1])
    twin.import_state(0, (labels, off, mem, nl), assign_new=True)
    a, b = sm.slice_objects(3), twin.slice_objects(0)
    assert list(a) == list(b)
    for k in a:
        assert a[k]['box'] == b[k]['box']
        np.testing.assert_array_equal(a[k]['starts'], b[k]['starts'])
    with pytest.raises(Exception, match='listed twice|out of range'):
        twin.import_state(0, (labels, off, np.zeros_like(mem), nl), assign_new=True)
