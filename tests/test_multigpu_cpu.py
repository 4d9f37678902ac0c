"""N > 1 path on CPU: world_size-2 (and 3) gloo groups run the z-slab driver of
empanada-napari_amd/multigpu.py with the oracle's arithmetic plugged in, and must reproduce the
single-process reference trace (tests/golden/median3d.npz: PanopticDeepLabRenderEngine3d)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ks, out_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import multigpu
    from oracle import postprocess as opp
    from oracle import sparse as osp
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'median3d.npz'))
    n = g['sem_logits'].shape[0]
    eng = opp.RenderEngine(None, [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                           coarse_boundaries=True)

    def forward_fn(lo, hi):
        return [{'sem': torch.from_numpy(opp.logits_to_prob(g['sem_logits'][z])), 'ctr_hmp': g['ctr_hmp'][z],
                 'offsets': g['offsets'][z]} for z in range(lo, hi)]

    def median_fn(maps):
        st = np.stack([m.numpy() for m in maps])
        return torch.from_numpy(np.sort(st, axis=0)[(len(maps) - 1) // 2])

    def segment_fn(item):
        cells = eng.cells(item['ctr_hmp'], item['offsets'], 1)
        return eng.postprocess(item['sem'].numpy(), cells)[0]

    def to_rle_fn(pans):
        return [osp.pan_seg_to_rle_seg(p, [1], 1000, [1], force_connected=True) for p in pans]

    segs = multigpu.distributed_stack_inference(n, forward_fn, median_fn, segment_fn, to_rle_fn, ks)
    if rank == 0:
        pans = np.stack([osp.rle_seg_to_pan_seg(s, (64, 64)) for s in segs]).astype(np.int64)
        np.save(out_path, pans)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,ks', [(2, 3), (2, 5), (3, 3), (2, 1)])
def test_zslab_driver_reproduces_reference_trace(tmp_path, golden_dir, world, ks):
    from oracle import sparse as osp
    out = str(tmp_path / 'pans.npy')
    mp.spawn(_worker, args=(world, _free_port(), ks, out), nprocs=world, join=True)
    got = np.load(out)
    g = np.load(os.path.join(golden_dir, 'median3d.npz'))
    ref = g[f'pan_ks{ks}'][:, 0].astype(np.int64)     # reference engine trace, (n,1,64,64)
    # compare through the same dense -> RLE -> dense round trip (connected components renumber instances)
    want = np.stack([osp.rle_seg_to_pan_seg(osp.pan_seg_to_rle_seg(p, [1], 1000, [1], True), (64, 64)) for p in ref])
    np.testing.assert_array_equal(got, want.astype(np.int64))


def test_slab_bounds_and_single_process_filter():
    from empanada_napari_amd import multigpu
    assert multigpu.slab_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert multigpu.slab_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    vals = [5, 1, 9, 0, 7, 2]
    med = lambda xs: sorted(xs)[(len(xs) - 1) // 2]
    assert multigpu.filtered_stack(vals, 3, med) == [5, 5, 5, 5, 5, 2]
    # split after slice 3: second slab needs the filtered carry [f3] and the first the raw look-ahead [7]
    a = multigpu.filtered_stack(vals[:4], 3, med, None, vals[4:5], first=True, last=False)
    b = multigpu.filtered_stack(vals[4:], 3, med, a[-1:], None, first=False, last=True)
    assert a + b == [5, 5, 5, 5, 5, 2]


def _oracle_tracker(pans, shape, min_size, min_extent):
    from oracle import sparse as osp
    m = osp.RLEMatcher(1, 1000, 0.25, 0.25)
    stack = [osp.apply_matchers(osp.pan_seg_to_rle_seg(p, [1], 1000, [1], force_connected=True), [m]) for p in pans]
    m.target_rle = None
    m.assign_new = False
    tr = osp.InstanceTracker(1, 1000, shape, 'xy')
    for idx in range(len(pans) - 1, -1, -1):
        tr.update(osp.apply_matchers(stack[idx], [m])[1], idx)
    tr.finish()
    osp.remove_small_objects(tr, min_size)
    osp.remove_pancakes(tr, min_extent)
    return tr


@pytest.mark.parametrize('block', ['0', '2', '3'])
@pytest.mark.parametrize('world,ks', [(2, 3), (3, 5), (2, 7)])
def test_public_multigpu_engine_spawns_its_ranks(golden_dir, world, ks, block, monkeypatch):
    """The PUBLIC API as the widget calls it (empanada_napari/multigpu.py:121-260): construct in one process, call
    infer_on_axis, get (stack, trackers).  The engine spawns ``world`` rank processes itself (gloo here, RCCL on GPUs),
    each runs the slab pipeline -- halo send, filtered carry, in-place median, run lists gathered on the host group --
    with the oracle's arithmetic plugged in, and the calling process matches + tracks in C++.  Result: the trackers the
    single-process reference trace gives (golden median3d.npz -> oracle matcher / tracker)."""
    import mg_oracle_backend as mgb
    from empanada_napari_amd import multigpu
    # EMP_MG_BLOCK: slices per block of the interleaved schedule (rank r owns blocks r, r + W, ...: ring halo / carry, the
    # chain thread; multigpu.block_stack_inference); '0' = one contiguous slab per rank (multigpu.slab_stack_inference)
    monkeypatch.setenv('EMP_MG_BLOCK', block)
    g = np.load(os.path.join(golden_dir, 'median3d.npz'))
    n = g['sem_logits'].shape[0]
    mc = {'golden': os.path.join(golden_dir, 'median3d.npz'), 'thing_list': [1], 'labels': [1],
          'class_names': {1: 'mito'}, 'padding_factor': 16, 'norms': {'mean': 0.5, 'std': 0.1}}
    eng = multigpu.MultiGPUEngine3d(mc, label_divisor=1000, median_kernel_size=ks, nms_kernel=3, confidence_thr=0.5,
                                    min_size=10, min_extent=2, world_size=world, dist_backend='gloo',
                                    backend_factory=mgb.oracle_backend_factory)
    try:
        vol = np.zeros((n, 64, 64), np.uint8)            # the stand-in backend reads the golden head tensors instead
        for _ in range(2):                                # the rank processes persist across calls
            stack, trackers = eng.infer_on_axis(vol, 'xy')
            assert stack is None and len(trackers) == 1 and trackers[0].class_id == 1
            want = _oracle_tracker(list(g[f'pan_ks{ks}'][:, 0].astype(np.int64)), vol.shape, 10, 2)
            got = trackers[0].instances
            assert len(want.instances) > 0 and [int(k) for k in got] == [int(k) for k in want.instances]
            for k in got:
                assert tuple(int(v) for v in got[k]['box']) == tuple(int(v) for v in want.instances[k]['box'])
                np.testing.assert_array_equal(got[k]['starts'], want.instances[k]['starts'])
                np.testing.assert_array_equal(got[k]['runs'], want.instances[k]['runs'])
        assert eng.dtype == np.int32 and all(p.is_alive() for p in eng._procs)
    finally:
        eng.close()
    assert eng._procs is None


@pytest.mark.parametrize('world,block,ks,shm_min', [(2, '0', 3, None), (2, '2', 3, '0'), (2, '4', 5, None), (3, '0', 5, '0'), (3, '2', 3, None),
                                                    (3, '4', 7, '0'), (1, '3', 3, None), (4, '1', 3, '0'), (4, '2', 5, None), (2, '1', 7, None)])
def test_public_multigpu_engine_multiclass_slab_matching(world, block, ks, shm_min, monkeypatch):
    """Several classes through the public API on gloo (BASELINE configs[4]'s class structure: two instance classes and a
    semantic one): every rank matches and tracks its own slab (multigpu.SlabMatcher) -- ghost slices, forward state down
    the ranks, backward state up, partial trackers to the caller -- and the result equals the sequential C++ matcher over
    the whole stack (itself pinned by the reference goldens, tests/test_host_sparse.py), size filters included, on all
    three axes."""
    import mg_oracle_backend as mgb
    import test_slab_matcher as tsm
    from empanada_napari_amd import multigpu
    from empanada_napari_amd import sparse as ps
    monkeypatch.setenv('EMP_MG_BLOCK', block)      # '0': contiguous slabs; else blocks of that many slices, interleaved over the ranks
    if shm_min is not None:      # every run list, however short, reaches rank 0 as a file in /dev/shm (multigpu.ranks_share_host)
        monkeypatch.setenv('EMP_MG_SHM_MIN', shm_min)
    shm_before = set(os.listdir('/dev/shm')) if os.path.isdir('/dev/shm') else set()
    monkeypatch.setattr(multigpu.MultiGPUEngine3d, 'MIN_WORLD', 1)      # one rank: every neighbour of a block is the rank itself
    shape = tsm.SHAPE
    mc = {'seed': 40, 'thing_list': tsm.THINGS, 'labels': tsm.LABELS, 'class_names': {1: 'a', 2: 'b', 3: 'c'},
          'padding_factor': 16, 'norms': {'mean': 0.5, 'std': 0.1}}
    # the stand-in backend checks every look-ahead / carry map the exchange delivers (mg_oracle_backend.LabelStackBackend)
    eng = multigpu.MultiGPUEngine3d(mc, label_divisor=tsm.DIV, median_kernel_size=ks, min_size=12, min_extent=2,
                                    world_size=world, dist_backend='gloo', backend_factory=mgb.label_stack_backend_factory)
    try:
        vol = np.zeros(shape, np.uint8)
        for axis_name, axis in (('xy', 0), ('xz', 1), ('yz', 2)):
            _, trackers = eng.infer_on_axis(vol, axis_name)
            want = tsm._sequential(tsm._stack(shape, axis, 40 + axis), axis_name, shape)
            assert [t.class_id for t in trackers] == tsm.LABELS
            for tr in trackers:
                ref = ps.InstanceTracker(tr.class_id, tsm.DIV, shape, axis_name)
                ref.instances = want[tr.class_id]
                ref.finished = True
                ps.remove_small_objects(ref, min_size=12)
                ps.remove_pancakes(ref, min_span=2)
                got = tr.instances
                assert len(ref.instances) > 0 and [int(k) for k in got] == [int(k) for k in ref.instances]
                for k in got:
                    assert tuple(int(v) for v in got[k]['box']) == tuple(int(v) for v in ref.instances[k]['box'])
                    np.testing.assert_array_equal(got[k]['starts'], ref.instances[k]['starts'])
                    np.testing.assert_array_equal(got[k]['runs'], ref.instances[k]['runs'])
            assert len(eng.last_host_s) == world
    finally:
        eng.close()
    if os.path.isdir('/dev/shm'):      # every file handed over was unlinked by the receiver
        assert not [f for f in set(os.listdir('/dev/shm')) - shm_before if f.startswith('emp_mg_')]


def test_public_multigpu_engine_errors():
    from empanada_napari_amd import multigpu
    mc = {'model': 'nowhere.pth', 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.5, 'std': 0.1}}
    with pytest.raises(Exception, match='2 or more GPUs'):          # multigpu.py:143-144 (no GPU in this container)
        multigpu.MultiGPUEngine3d(mc)
    assert multigpu.active_ranks(10, 8, 5) == 5 and multigpu.active_ranks(3, 4, 1) == 3 and multigpu.active_ranks(1, 4, 7) == 1


def test_rank_failure_is_reported(golden_dir):
    """a rank that dies (here: the backend factory raises) surfaces as an exception in the caller, not as a hang"""
    import mg_oracle_backend as mgb
    from empanada_napari_amd import multigpu
    mc = {'golden': '/nonexistent/golden.npz', 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'},
          'padding_factor': 16, 'norms': {'mean': 0.5, 'std': 0.1}}
    eng = multigpu.MultiGPUEngine3d(mc, median_kernel_size=3, world_size=2, dist_backend='gloo',
                                    backend_factory=mgb.oracle_backend_factory)
    with pytest.raises(RuntimeError, match='rank'):
        eng.infer_on_axis(np.zeros((8, 64, 64), np.uint8), 'xy')
    assert eng._procs is None


def test_shared_volume_follows_the_callers_array(monkeypatch):
    """the shared-memory copy of a numpy volume that the spawned ranks map must hold what the caller
    passes NOW -- not what an array with the same id / shape / buffer address held on an earlier call (a freed and
    re-allocated volume of a per-file loop, or the same array edited in place).  The allocation is reused, the content
    is copied on every call, and close() drops it."""
    from empanada_napari_amd import multigpu
    mc = {'model': 'unused.pth', 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.5, 'std': 0.1}}
    eng = multigpu.MultiGPUEngine3d(mc, median_kernel_size=3, world_size=2, dist_backend='gloo')
    sent = []

    class Q:
        def put(self, cmd):
            sent.append(cmd)

    eng._procs, eng._cmd = [], [Q()]
    monkeypatch.setattr(eng, '_collect', lambda what: {0: None})
    rng = np.random.default_rng(0)
    storages = set()
    for i in range(6):
        v = rng.integers(0, 255, (5, 8, 8), dtype=np.uint8)      # freed at the next iteration: ids and addresses recur
        eng._segs_spawn(v, 'xy')
        np.testing.assert_array_equal(sent[-1][1].numpy(), v)
        v[2] = 255 - v[2]                                        # the same array, edited in place
        eng._segs_spawn(v, 'xz')
        np.testing.assert_array_equal(sent[-1][1].numpy(), v)
        assert sent[-1][1].is_shared()
        storages.add(sent[-1][1].data_ptr())
        del v
    assert len(storages) == 1, 'the shared-memory allocation is kept across calls of one shape'
    eng._segs_spawn(np.zeros((3, 8, 8), np.uint16), 'xy')        # another shape / dtype: a new allocation
    assert sent[-1][1].numpy().dtype == np.uint16 and tuple(sent[-1][1].shape) == (3, 8, 8)
    eng.close()
    assert '_shm' not in eng.__dict__


def _silent_peer_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import time
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import multigpu
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['EMP_MG_CHAIN_TIMEOUT'] = '2'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cg = multigpu._default_chain_group(None)
    assert multigpu._default_chain_group(None) is cg, 'one chain group per parent group, not one per call'
    if rank == 0:
        t0 = time.monotonic()
        try:
            multigpu._recv_pickled(1, cg)          # the peer "failed": it never sends
            res = 'returned'
        except Exception as e:                     # noqa: BLE001
            res = 'raised after %.1f s: %s' % (time.monotonic() - t0, type(e).__name__)
        open(out_path, 'w').write(res)
    else:
        time.sleep(6.0)
    os._exit(0)      # no orderly teardown: rank 0's group is broken by design


def test_chain_receive_times_out_instead_of_hanging(tmp_path):
    """a block-chain receive whose peer never sends (it failed elsewhere) raises after
    EMP_MG_CHAIN_TIMEOUT seconds -- the chain group is created with that timeout -- and the chain group is created once."""
    out = str(tmp_path / 'res.txt')
    mp.spawn(_silent_peer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = open(out).read()
    assert res.startswith('raised after'), res
    assert float(res.split()[2]) < 5.5, res


def _spmd_worker(rank, world, port, block, defer, out_path):
    for d in (ROOT, os.path.join(ROOT, 'tests')):
        if d not in sys.path:
            sys.path.insert(0, d)
    import pickle
    import __graft_entry__ as graft
    graft.load_package()
    import mg_oracle_backend as mgb
    import test_slab_matcher as tsm
    from empanada_napari_amd import multigpu
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), EMP_MG_BLOCK=block, EMP_MG_DEFER=defer,
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    multigpu.MultiGPUEngine3d.MIN_WORLD = 1
    mc = {'seed': 40, 'thing_list': tsm.THINGS, 'labels': tsm.LABELS, 'class_names': {1: 'a', 2: 'b', 3: 'c'},
          'padding_factor': 16, 'norms': {'mean': 0.5, 'std': 0.1}}
    eng = multigpu.MultiGPUEngine3d(mc, label_divisor=tsm.DIV, median_kernel_size=3, min_size=12, min_extent=2,
                                    dist_backend='gloo', backend_factory=mgb.label_stack_backend_factory)
    vol = np.zeros(tsm.SHAPE, np.uint8)
    res = {}
    held = []
    for axis_name in ('xy', 'xz', 'yz'):          # back to back: an axis' backward chain runs behind the next axis' GPU loop
        _, trackers = eng.infer_on_axis(vol, axis_name)
        held.append((axis_name, trackers))
    eng.wait()
    if rank == 0:
        for axis_name, trackers in held:
            res[axis_name] = {tr.class_id: {int(k): (tuple(int(v) for v in o['box']), np.asarray(o['starts']), np.asarray(o['runs']))
                                            for k, o in tr.instances.items()} for tr in trackers}
        pickle.dump(res, open(out_path, 'wb'))
    else:
        assert all(t is None for _, t in held)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,block,defer', [(2, '2', '1'), (3, '3', '1'), (2, '2', '0'), (2, '0', '1')])
def test_spmd_ranks_defer_the_backward_chain_behind_the_next_axis(tmp_path, world, block, defer):
    """Under an SPMD launch (torchrun) ``infer_on_axis`` returns when a rank's GPU phase is through; the backward chain,
    the tracking and the gather to rank 0 finish on the chain thread while the NEXT axis is already running (VERDICT r03
    item 5b).  Three axes called back to back on gloo ranks, the trackers read only at the end: identical to the
    sequential C++ matcher over the whole stack, per axis and class, size filters included; also with the deferral off
    and with contiguous slabs (which do not defer)."""
    import pickle
    import test_slab_matcher as tsm
    from empanada_napari_amd import sparse as ps
    out = str(tmp_path / 'res.pkl')
    mp.spawn(_spmd_worker, args=(world, _free_port(), block, defer, out), nprocs=world, join=True)
    res = pickle.load(open(out, 'rb'))
    for axis_name, axis in (('xy', 0), ('xz', 1), ('yz', 2)):
        want = tsm._sequential(tsm._stack(tsm.SHAPE, axis, 40 + axis), axis_name, tsm.SHAPE)
        for cid in tsm.LABELS:
            ref = ps.InstanceTracker(cid, tsm.DIV, tsm.SHAPE, axis_name)
            ref.instances = want[cid]
            ref.finished = True
            ps.remove_small_objects(ref, min_size=12)
            ps.remove_pancakes(ref, min_span=2)
            got = res[axis_name][cid]
            assert len(ref.instances) > 0 and list(got) == [int(k) for k in ref.instances]
            for k, (box, st, rn) in got.items():
                assert box == tuple(int(v) for v in ref.instances[k]['box'])
                np.testing.assert_array_equal(st, ref.instances[k]['starts'])
                np.testing.assert_array_equal(rn, ref.instances[k]['runs'])


def test_ring_shift_plan_is_lock_step():
    """The look-ahead exchange of the block schedule as RCCL needs it (multigpu.ring_shift_plan): every send of shift s
    has its receive in shift s of the destination -- so the operations of a shift can be ONE group, and no rank ever has
    a send queued in front of the receive its peer's send waits for --, every block but the last gets its look-ahead
    exactly once, from the block after it, into a buffer that exists when the shift is posted (the block of that round
    or, on the last rank, of the round before), and no rank posts more shifts than it has rounds + 1."""
    from empanada_napari_amd.multigpu import ring_shift_plan
    assert ring_shift_plan(0, 1, 9) == []
    for W in range(2, 9):
        for NB in range(1, 40):
            plans = [ring_shift_plan(r, W, NB) for r in range(W)]
            served = []
            for s in range(max(len(p) for p in plans)):
                sends, recvs = {}, {}
                for r, plan in enumerate(plans):
                    if s >= len(plan):
                        continue
                    assert s <= len(range(r, NB, W))
                    for b, dst in plan[s]['send']:
                        assert b == r + s * W and dst == (b - 1) % W and (r, dst) not in sends
                        sends[(r, dst)] = b
                    for b, src in plan[s]['recv']:
                        assert b % W == r and (b - r) // W in (s, s - 1) and src == (r + 1) % W
                        recvs[(src, r)] = b
                assert set(sends) == set(recvs)
                for key, b in sends.items():
                    assert recvs[key] == b - 1
                    served.append(b - 1)
            assert sorted(served) == list(range(NB - 1))


def _shm_part_worker(rank, world, port, out_path):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import multigpu
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), EMP_MG_SHM_MIN='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    shared = multigpu.ranks_share_host(None)
    rng = np.random.default_rng(3)
    n = 5000
    part = {1: (np.arange(3, dtype=np.int64), np.zeros((3, 6), np.int64), np.array([n - 7, 0, 7], np.int64),
                rng.integers(0, 1 << 40, n).astype(np.int64), rng.integers(1, 99, n).astype(np.int64)),
            2: (np.zeros(0, np.int64), np.zeros((0, 6), np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))}
    if rank == 1:
        multigpu._send_part({'part': part, 'host_s': 0.5}, 0, None, shared)
        assert not multigpu._SHM_PENDING
    else:
        got = multigpu._recv_part(1, None)
        ok = shared and isinstance(got['part'][1][3].base, np.memmap) or isinstance(got['part'][1][3], np.memmap)
        same = all(np.array_equal(np.asarray(a), b) for c in part for a, b in zip(got['part'][c], part[c]))
        left = [f for f in os.listdir('/dev/shm') if f.startswith('emp_mg_')] if os.path.isdir('/dev/shm') else []
        with open(out_path, 'w') as f:
            f.write(f'{int(shared)} {int(bool(ok))} {int(same)} {len(left)} {got["host_s"]}')
    dist.barrier()
    dist.destroy_process_group()


def test_run_lists_travel_through_shared_memory_between_ranks_of_one_host(tmp_path):
    """multigpu._send_part / _recv_part with ``ranks_share_host``: the run lists arrive as a mapping of a /dev/shm file that
    is already unlinked (nothing is left behind), bit for bit; small tables still travel through the group"""
    if not os.path.isdir('/dev/shm'):
        pytest.skip('no /dev/shm')
    out = str(tmp_path / 'res.txt')
    mp.spawn(_shm_part_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    shared, mapped, same, left, host_s = open(out).read().split()
    assert (shared, mapped, same, left, host_s) == ('1', '1', '1', '0', '0.5')


def test_wait_joins_every_deferred_axis_and_reraises_the_first_failure():
    """a deferred chain of an EARLIER axis that failed on this rank must not be dropped when a later axis'
    future replaces it: every future is kept until joined; wait() joins all of them and re-raises the earliest failure"""
    from concurrent.futures import Future
    from empanada_napari_amd import multigpu
    eng = multigpu.MultiGPUEngine3d.__new__(multigpu.MultiGPUEngine3d)
    a, b, c = Future(), Future(), Future()
    a.set_result(1)
    b.set_exception(RuntimeError('axis xz: chain failed'))
    c.set_exception(ValueError('axis yz: later'))
    eng._spmd_pending = [a, b, c]
    with pytest.raises(RuntimeError, match='axis xz'):
        eng.wait()
    assert not eng.__dict__.get('_spmd_pending')
    eng.wait()      # nothing left: no error twice


def _spmd_store_worker(rank, world, port, store, out_path):
    for d in (ROOT, os.path.join(ROOT, 'tests')):
        if d not in sys.path:
            sys.path.insert(0, d)
    import __graft_entry__ as graft
    graft.load_package()
    import mg_oracle_backend as mgb
    import test_slab_matcher as tsm
    from empanada_napari_amd import multigpu, sparse as ps
    from oracle import sparse as osp

    def host_fill(volume, trackers):      # the product fills on the GPU (no CPU fallback); this test is about WHO fills
        for tr in trackers:
            osp.numpy_fill_instances(volume, tr.instances)
    ps.fill_panoptic_volume = host_fill
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), EMP_MG_BLOCK='2', LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    multigpu.MultiGPUEngine3d.MIN_WORLD = 1
    mc = {'seed': 40, 'thing_list': tsm.THINGS, 'labels': tsm.LABELS, 'class_names': {1: 'a', 2: 'b', 3: 'c'},
          'padding_factor': 16, 'norms': {'mean': 0.5, 'std': 0.1}}
    eng = multigpu.MultiGPUEngine3d(mc, label_divisor=tsm.DIV, median_kernel_size=3, min_size=12, min_extent=2,
                                    dist_backend='gloo', backend_factory=mgb.label_stack_backend_factory,
                                    store_url=store, save_panoptic=True, chunk_size=(8, 8, 8))
    created = []
    make = eng.create_panoptic_stack
    eng.create_panoptic_stack = lambda *a: (created.append(a[0]), make(*a))[1]
    vol = np.zeros(tsm.SHAPE, np.uint8)
    stack, trackers = eng.infer_on_axis(vol, 'xy')
    eng.wait()
    if rank == 0:
        assert created == ['xy'] and eng.zarr_store is not None
        want = np.zeros(tsm.SHAPE, np.int32)
        ps.fill_panoptic_volume(want, trackers)
        assert want.any()
        np.testing.assert_array_equal(np.asarray(stack[...]), want)
        open(out_path, 'w').write('ok')
    else:
        # a non-zero rank neither opens (mode 'w' deletes) nor creates the dataset, and allocates no volume-sized array
        assert created == [] and eng.zarr_store is None and stack is None and trackers is None
    dist.barrier()
    dist.destroy_process_group()


def test_spmd_ranks_leave_the_panoptic_store_to_rank_0(tmp_path):
    """with save_panoptic and a store, every SPMD rank used to run create_dataset(overwrite=True) on the same
    array; now only rank 0 touches the store, after inference (the reference: multigpu.py:198-212 on the main process)"""
    out = str(tmp_path / 'ok.txt')
    mp.spawn(_spmd_store_worker, args=(2, _free_port(), str(tmp_path / 'pan.zarr'), out), nprocs=2, join=True)
    assert open(out).read() == 'ok'
