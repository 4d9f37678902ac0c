"""Multi-GPU stack inference: contiguous z-slabs, one process per GPU, RCCL over xGMI.

Reference: ``empanada_napari/multigpu.py:27-119`` shards slices round-robin with a
``DistributedSampler`` and all-gathers every dense ``sem`` / ``cells`` map to every rank
(two NCCL all-gathers per step), then runs median + post-processing on rank 0's CPU.

Re-design for MI355X: rank r owns the contiguous slab
``[lo_r, hi_r)`` of slices along the inference axis.
  * forward: embarrassingly parallel (a slice is never split: global ASPP pooling);
  * median halo: the filter is RECURSIVE (engines.py:76-84): slice z uses the already
    filtered maps of z-mid..z-1 and the raw maps of z..z+mid.  Rank r therefore needs the
    first ``mid`` RAW maps of rank r+1 (one neighbour send, posted before anything waits)
    and the last ``mid`` FILTERED maps of rank r-1 (a carry that ripples down the ranks;
    the per-pixel median is a tiny HBM-bound kernel, so the ripple costs ~W median passes);
  * post-processing (voting, merge, connected components, run extraction): per slab, on
    the GPU;
  * slice-to-slice matching and tracking (round 3): per slab too.  The passes of
    ``patterns.py:68-121`` are a chain along the axis, but a slice hands its neighbour only
    the grouping of its components into labelled objects and the label counter -- a few KB.
    Every rank builds the run-intersection tables of its slab up front (the bulk of the
    work, label-independent, overlapped with its GPU's run extraction), then the forward
    state ripples down the ranks and the backward state back up (``SlabMatcher``), each
    rank tracks its own slices, and rank 0 only concatenates the per-slab tracks.
    Round 2 gathered every run list on rank 0 and matched there: 1 ms of sequential host
    work per 1024^2 slice against 0.75 ms / ranks of GPU work.
The exchange uses point-to-point ``isend/irecv`` between neighbours -- xGMI is a
point-to-point fabric, a ring all-gather of whole maps would be bound by one 153 GB/s link.

The driver is backend-agnostic: it moves tensors with ``torch.distributed`` and calls a
backend object for the arithmetic, so the sharding / halo / carry logic is tested on CPU
with gloo (tests/test_multigpu_cpu.py, with the oracle's arithmetic plugged in) and runs
unchanged on RCCL.

Launching: like the reference (``multigpu.py:214-238``, ``mp.spawn`` inside ``infer_on_axis``)
``MultiGPUEngine3d.infer_on_axis`` starts its own ranks -- one process per GPU, started once
and kept for the following axes -- unless the caller already runs SPMD (``torchrun`` with an
initialised process group, as ``bench.py --workload stack3d`` does): then every rank calls
``infer_on_axis`` and rank 0 gets the result.
"""
import os
import socket
import traceback

import numpy as np
import torch
import torch.distributed as dist


def slab_bounds(n, world):
    """Contiguous, balanced split of n slices over ``world`` ranks -> list of (lo, hi)."""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def filtered_stack(raw, ks, median_fn, prev_filtered=None, next_raw=None, first=True, last=True):
    """Recursive median of a slab.

    raw: list of per-slice maps of this slab; prev_filtered: the ``mid`` filtered maps just before the
    slab (None on the first slab); next_raw: the ``mid`` raw maps just after it (None on the last slab).
    The first / last ``mid`` slices of the WHOLE stack stay unfiltered (engines.py:70-72,89-90).
    """
    mid = (ks - 1) // 2
    if mid == 0:
        return list(raw)
    n = len(raw)
    hist = list(prev_filtered) if prev_filtered is not None else []
    ahead = list(raw) + (list(next_raw) if next_raw is not None else [])
    out = []
    for z in range(n):
        head = first and (len(hist) < mid)        # one of the first mid slices of the stack
        tail = last and (z + mid >= n)            # one of the last mid slices of the stack
        if head or tail:
            f = raw[z]
        else:
            f = median_fn(hist[-mid:] + ahead[z:z + mid + 1])
        out.append(f)
        hist.append(f)
    return out


def distributed_filtered_sem(raw, ks, median_fn, group=None):
    """Each rank passes the raw maps of its slab (torch tensors, same shape) and gets the filtered ones."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mid = (ks - 1) // 2
    if world == 1 or mid == 0:
        return filtered_stack(raw, ks, median_fn)
    assert len(raw) >= mid, 'every slab must hold at least (ks-1)/2 slices'
    reqs = []
    next_raw = None
    if rank > 0:       # my first raw maps are the look-ahead of the previous rank
        reqs.append(dist.isend(torch.stack(raw[:mid]).contiguous(), dst=rank - 1, group=group))
    if rank < world - 1:
        buf = torch.empty((mid,) + tuple(raw[0].shape), dtype=raw[0].dtype, device=raw[0].device)
        dist.recv(buf, src=rank + 1, group=group)
        next_raw = list(buf.unbind(0))
    prev = None
    if rank > 0:       # carry: filtered tail of the previous slab
        buf = torch.empty((mid,) + tuple(raw[0].shape), dtype=raw[0].dtype, device=raw[0].device)
        dist.recv(buf, src=rank - 1, group=group)
        prev = list(buf.unbind(0))
    out = filtered_stack(raw, ks, median_fn, prev, next_raw, first=rank == 0, last=rank == world - 1)
    if rank < world - 1:
        reqs.append(dist.isend(torch.stack(out[-mid:]).contiguous(), dst=rank + 1, group=group))
    for r in reqs:
        r.wait()
    return out


def distributed_stack_inference(n_slices, forward_fn, median_fn, segment_fn, to_rle_fn, ks, group=None,
                                segment_batch_fn=None):
    """SPMD body of one axis.

    forward_fn(lo, hi)  -> list of per-slice dicts with at least 'sem' (tensor) for slices [lo, hi)
    median_fn(list)     -> per-pixel median of an odd number of 'sem' tensors
    segment_fn(item)    -> panoptic map of one slice (any array type to_rle_fn accepts)
    segment_batch_fn(items) -> the same for a list of slices at once (optional; replaces segment_fn)
    to_rle_fn(list)     -> list of rle_seg dicts for a list of panoptic maps
    Returns on rank 0 the list of rle_seg dicts of ALL slices in order, elsewhere None.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = slab_bounds(n_slices, world)[rank]
    items = forward_fn(lo, hi)
    sems = distributed_filtered_sem([it['sem'] for it in items], ks, median_fn, group)
    items = [dict(it, sem=s) for it, s in zip(items, sems)]
    pans = segment_batch_fn(items) if segment_batch_fn is not None else [segment_fn(it) for it in items]
    segs = to_rle_fn(pans) if pans else []
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(segs, gathered, dst=0, group=group)
    if rank != 0:
        return None
    return [s for part in gathered for s in part]


# ----------------------------------------------------------------------------
# slab pipeline: forward of the whole slab -> halo / carry exchange -> ONE in-place recursive-median launch ->
# voting / merge / connected components / run extraction -> run lists to rank 0
# ----------------------------------------------------------------------------
def active_ranks(n_slices, world, ks):
    """Ranks that get a slab: every slab must hold at least ``mid`` slices (its head is somebody's look-ahead and its
    tail somebody's carry); with fewer slices than that per rank the last ranks idle."""
    mid = (ks - 1) // 2
    return max(1, min(world, n_slices // max(1, mid)))


def _via_host(t, group):
    """gloo moves host tensors only: device maps are staged through the host on that backend (the CPU tests, and the
    GPU test that runs two ranks on ONE device); RCCL sends them as they are."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


class _Sent:
    """an isend in flight together with the buffer it reads"""

    def __init__(self, work, buf):
        self.work, self.buf = work, buf

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.work = self.buf = None


def _isend(t, dst, group):
    buf = t.contiguous()
    if _via_host(buf, group):
        buf = buf.cpu()
    return _Sent(dist.isend(buf, dst=dst, group=group), buf)


def _recv(t, src, group):
    if _via_host(t, group):
        buf = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(buf, src=src, group=group)
        t.copy_(buf)
    else:
        dist.recv(t, src=src, group=group)


class _Shift:
    """a grouped neighbour exchange in flight: the send buffers it reads and, for a group that moves host tensors only, the
    host copies of what it receives"""

    def __init__(self, works, staged, keep):
        self.works, self.staged, self.keep = list(works), staged, keep

    def wait(self):
        for w in self.works:
            w.wait()
        for dst, buf in self.staged:
            dst.copy_(buf)
        self.works, self.staged, self.keep = [], [], []


def _ring_shift(sends, recvs, group):
    """``sends`` [(tensor, dst)] and ``recvs`` [(tensor, src)] posted as ONE group (dist.batch_isend_irecv; RCCL: one
    ncclGroupStart / End, executed by one kernel that serves all of its operations at once).  Stream-ordered RCCL
    point-to-point calls issued one by one block each other when two ranks each have a send queued in front of the
    receive the other side's send waits for; the operations of a group have no order among themselves, so ranks that all
    post "first maps to the left, look-ahead from the right" at the same point of the program cannot."""
    ops, staged, keep = [], [], []
    for t, dst in sends:
        buf = t.contiguous()
        if _via_host(buf, group):
            buf = buf.cpu()
        keep.append(buf)
        ops.append(dist.P2POp(dist.isend, buf, dst, group))
    for t, src in recvs:
        buf = t
        if _via_host(t, group):
            buf = torch.empty(t.shape, dtype=t.dtype)
            staged.append((t, buf))
        ops.append(dist.P2POp(dist.irecv, buf, src, group))
    return _Shift(dist.batch_isend_irecv(ops) if ops else [], staged, keep)


def _send_obj(obj, dst, group):
    dist.send_object_list([obj], dst=dst, group=group)


def _recv_obj(src, group):
    box = [None]
    dist.recv_object_list(box, src=src, group=group)
    return box[0]


class _SentObj:
    """a pickled object on its way (two isends: size, bytes) together with the buffers they read"""

    def __init__(self, works, bufs):
        self.works, self.bufs = works, bufs

    def wait(self):
        for w in self.works:
            w.wait()
        self.bufs = None


def _isend_pickled(obj, dst, group):
    """non-blocking counterpart of _send_obj for the chain thread (received with _recv_pickled): the sender never waits for
    the receiver to come round to it"""
    import pickle
    data = torch.frombuffer(bytearray(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)), dtype=torch.uint8)
    size = torch.tensor([data.numel()], dtype=torch.int64)
    return _SentObj([dist.isend(size, dst=dst, group=group), dist.isend(data, dst=dst, group=group)], [size, data])


def chain_timeout():
    """timeout of the block chain's host group (EMP_MG_CHAIN_TIMEOUT seconds, default 600): a chain receive whose peer
    failed -- and therefore never sends -- raises after this long instead of hanging the job for ever.  [gloo completes an
    irecv only inside wait() (is_completed() stays False with the data already there), so the bound is the GROUP's timeout
    on a blocking receive, not a poll.]"""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get('EMP_MG_CHAIN_TIMEOUT', '600')))


def _recv_pickled(src, group):
    import pickle
    size = torch.zeros(1, dtype=torch.int64)
    dist.recv(size, src=src, group=group)
    data = torch.empty(int(size.item()), dtype=torch.uint8)
    dist.recv(data, src=src, group=group)
    return pickle.loads(data.numpy().tobytes())


_SHM_DIR = '/dev/shm'
_SHM_MIN = 1 << 20          # bytes of run lists below which a part travels through the group
_SHM_SAME_HOST = {}         # group -> do all its ranks share this host's /dev/shm?
_SHM_PENDING = []           # files written here that the receiver has not been told about yet (removed at exit)


def _host_id():
    import socket
    try:
        with open('/proc/sys/kernel/random/boot_id') as f:
            boot = f.read().strip()
    except OSError:
        boot = ''
    return socket.gethostname() + ':' + boot


def ranks_share_host(group):
    """True when every rank of ``group`` sees the same /dev/shm (same hostname and boot id, the directory exists,
    EMP_MG_SHM != 0): the run lists of the per-slab tracks then reach rank 0 as files in shared memory, written by all
    ranks at once and mapped by rank 0, instead of one after the other through the group's sockets (gloo over loopback:
    2.3 GB/s measured; at 4096^2 the tracks are 3.3 MB per slice -- 11.8 GB for the seven other ranks of a 4096-slice job).
    COLLECTIVE on first use per group (every rank calls it at the same point); cached."""
    # keyed by the group's MEMBERSHIP (an id() alone can be reused by another group once this one is destroyed; two groups
    # over the same global ranks share the answer)
    try:
        key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    except Exception:       # noqa: BLE001
        key = (id(group), dist.get_world_size(group))
    if key not in _SHM_SAME_HOST:
        mine = _host_id() if (os.environ.get('EMP_MG_SHM', '1') != '0' and os.path.isdir(_SHM_DIR)) else None
        ids = [None] * dist.get_world_size(group)
        dist.all_gather_object(ids, mine, group=group)
        same = mine is not None and all(i == mine for i in ids)
        # hostname + boot id do not prove a shared tmpfs (pods on one node: same ids, private /dev/shm mounts): the first
        # rank of the group writes a token file, every rank says whether it sees it.  Collective as well --
        # every rank reaches it, whatever `same` says on it (the ids were all-gathered: `same` agrees on all ranks).
        if same:
            import uuid
            tok = [None]
            first = dist.get_rank(group) == 0
            if first:
                tok[0] = os.path.join(_SHM_DIR, 'emp_mg_probe_%d_%s' % (os.getpid(), uuid.uuid4().hex))
                try:
                    with open(tok[0], 'w') as f:
                        f.write('1')
                except OSError:
                    tok[0] = None
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(tok, src=src, group=group)
            seen = [None] * dist.get_world_size(group)
            dist.all_gather_object(seen, bool(tok[0]) and os.path.exists(tok[0]), group=group)
            same = all(seen)
            if first and tok[0]:
                try:
                    os.unlink(tok[0])
                except OSError:
                    pass
        _SHM_SAME_HOST[key] = same
    return _SHM_SAME_HOST[key]


def _shm_cleanup():
    for path in list(_SHM_PENDING):
        try:
            os.unlink(path)
        except OSError:
            pass


def _shm_write(starts, runs):
    """[starts | runs] as one raw int64 file in shared memory -> path, or None when it cannot be written (full tmpfs)"""
    import atexit
    import uuid
    if not _SHM_PENDING:
        atexit.register(_shm_cleanup)
    path = os.path.join(_SHM_DIR, 'emp_mg_%d_%s.i64' % (os.getpid(), uuid.uuid4().hex))
    n = len(starts)
    try:
        _SHM_PENDING.append(path)
        mm = np.memmap(path, dtype=np.int64, mode='w+', shape=(2 * n,))
        mm[:n] = starts
        mm[n:] = runs
        mm.flush()
        del mm
        return path
    except (OSError, ValueError):
        _SHM_PENDING.remove(path)
        try:
            os.unlink(path)
        except OSError:
            pass
        return None


def _send_part(res, dst, group, shm=False):
    """a rank's result of slab_stack_inference (None for a rank without slices) to ``dst``.  ``shm``: the ranks share a
    host (``ranks_share_host``) -- run lists of a megabyte and more are handed over as files in /dev/shm"""
    if res is None:
        _send_obj(None, dst, group)
        return
    files = {}
    if shm:
        for c, p in res['part'].items():
            if len(p[3]) and len(p[3]) * 16 >= int(os.environ.get('EMP_MG_SHM_MIN', _SHM_MIN)):
                path = _shm_write(p[3], p[4])
                if path is not None:
                    files[c] = path
    head = dict(res, part={c: (p[0], p[1], p[2]) for c, p in res['part'].items()}, shm_files=files)
    _send_obj(head, dst, group)
    for path in files.values():        # from here on the receiver owns them
        _SHM_PENDING.remove(path)
    for c, p in res['part'].items():
        if len(p[3]) and c not in files:
            dist.send(torch.from_numpy(p[3]), dst=dst, group=group)
            dist.send(torch.from_numpy(p[4]), dst=dst, group=group)


def _recv_part(src, group):
    res = _recv_obj(src, group)
    if res is None:
        return None
    files = res.pop('shm_files', {})
    part = {}
    for c, (labels, boxes, counts) in res['part'].items():
        n = int(counts.sum())
        if c in files:
            # mapped, then unlinked at once: the pages live as long as the mapping (the merge copies out of it)
            mm = np.memmap(files[c], dtype=np.int64, mode='r', shape=(2 * n,))
            os.unlink(files[c])
            starts, runs = mm[:n], mm[n:]
        else:
            starts, runs = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
            if n:
                dist.recv(torch.from_numpy(starts), src=src, group=group)
                dist.recv(torch.from_numpy(runs), src=src, group=group)
        part[c] = (labels, boxes, counts, starts, runs)
    res['part'] = part
    return res


class SlabMatcher:
    """Matching + tracking of ONE rank's slab of slices (all classes), bit-identical to the sequential passes over the
    whole stack (reference: empanada/inference/patterns.py:68-134, matcher.py:234-326, tracker.py:61-123).

    Local slice order in every class matcher: the rank's own slices [0, n_own) as they are pushed, then the ghost of
    slice lo-1 (``prev``), then the ghost of slice hi (``next``).  Ghosts are pushed from the owner's entry, so their
    components carry the owner's indices and an exported state can be imported verbatim.

    ``push(entries)`` may be called per group of slices while the GPU extracts the next group: it runs on a worker
    thread (the C++ calls release the GIL) and builds the pair tables of consecutive own slices as they arrive.
    ``finish(...)`` does the neighbour exchange and the two chains and returns the partial trackers of this slab."""

    def __init__(self, labels, thing_list, label_divisor, iou_thr, ioa_thr, width, head=False):
        from concurrent.futures import ThreadPoolExecutor
        from . import sparse
        self.labels, self.width = list(labels), int(width)
        self.things = [c for c in self.labels if c in thing_list]
        self.sm = {c: sparse.StackMatcher(c, label_divisor, iou_thr, ioa_thr, match=c in thing_list) for c in self.labels}
        self.n_own = 0
        self.head = bool(head)     # first slab of the stack: its forward chain needs no neighbour and runs as the groups arrive
        self.first_entry = self.last_entry = None
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='emp-slab-match')
        self._jobs = []
        self.host_s = 0.0          # host time of this rank's matcher (push + tables + chains + tracking), for the bench
        self.tail_s = 0.0          # the part of it behind the GPU work (ghosts, the two chains, tracking): not overlapped

    def _push_entry(self, entry):
        for c, sm in self.sm.items():
            v = entry[c]
            if isinstance(v, tuple):
                sm.push_runs(v[0], self.width, v[1])
            else:
                sm.push_objects(v)

    def _push_job(self, entries, start):
        import time
        t0 = time.perf_counter()
        for c, sm in self.sm.items():      # a group from the run extractor goes in as one call (built on the library's threads)
            vals = [e[c] for e in entries]
            if all(isinstance(v, tuple) for v in vals) and len({v[1] for v in vals}) == 1:
                sm.push_runs_many([v[0] for v in vals], self.width, vals[0][1])
            else:
                for v in vals:
                    if isinstance(v, tuple):
                        sm.push_runs(v[0], self.width, v[1])
                    else:
                        sm.push_objects(v)
        for c in self.things:      # pair tables (start-1 .. start+len-1]: label-independent, ahead of the chain
            self.sm[c].prepare(max(0, start - 1), start + len(entries) - 1)
            if self.head:          # forward steps of the new slices (patterns.py:68-100), behind the GPU work of the next group
                self.sm[c].run_range(start, start + len(entries) - 1, +1)
        self.host_s += time.perf_counter() - t0

    def push(self, entries):
        entries = list(entries)
        if not entries:
            return
        if self.first_entry is None:
            self.first_entry = entries[0]
        self.last_entry = entries[-1]
        self._jobs.append(self._pool.submit(self._push_job, entries, self.n_own))
        self.n_own += len(entries)

    # ---- the phases of a slab, in the order the chain runs them (finish() strings them together for ONE slab per rank) ----
    def wait_pushed(self):
        for j in self._jobs:
            j.result()
        self._pool.shutdown()
        self._pushed_s = self.host_s
        self.i_prev = self.i_next = None
        self.phase_s = {'ghosts': 0.0, 'forward': 0.0, 'backward': 0.0, 'track': 0.0}

    def _timed(self, key, t0):
        import time
        dt = time.perf_counter() - t0
        self.phase_s[key] += dt
        self.host_s += dt

    def push_prev_ghost(self, entry):
        """the raw entry of slice lo-1 (the last slice of the slab before) as a ghost behind the own slices"""
        import time
        t0 = time.perf_counter()
        self._push_entry(entry)
        self.i_prev = self.n_own
        self._timed('ghosts', t0)

    def push_next_ghost(self, entry):
        """the raw entry of slice hi (the first slice of the slab behind) as a ghost (after the previous-slice ghost, if any)"""
        import time
        t0 = time.perf_counter()
        self._push_entry(entry)
        self.i_next = self.n_own + (1 if self.i_prev is not None else 0)
        self._timed('ghosts', t0)

    def forward_chain(self, states=None):
        """forward pass over the slab (patterns.py:68-100).  ``states``: the matching state of slice lo-1 from the slab
        before ({class: state}; its ghost is pushed) or None on the first slab (whose chain already ran behind its pushes
        when ``head``) -> the state of this slab's last slice for the next one."""
        import time
        n = self.n_own
        t0 = time.perf_counter()
        for c in self.things:
            sm = self.sm[c]
            if states is not None:
                sm.import_state(self.i_prev, states[c], assign_new=True)
            if not self.head:
                sm.run_range(0, n - 1, +1)
        out = {c: self.sm[c].export_state(n - 1) for c in self.things}
        self._timed('forward', t0)
        return out

    def backward_chain(self, states=None):
        """backward pass (patterns.py:102-121: fresh target, no new labels).  ``states``: the backward state of slice hi from
        the slab behind (its ghost is pushed) or None on the last slab -> the state of this slab's first slice."""
        import time
        n = self.n_own
        t0 = time.perf_counter()
        for c in self.things:
            sm = self.sm[c]
            if states is not None:
                sm.import_state(self.i_next, (states[c][0], states[c][1], states[c][2], -1), assign_new=False)
            else:
                sm.begin_backward()
            sm.run_range(0, n - 1, -1)
        out = {c: self.sm[c].export_state(0) for c in self.things}
        self._timed('backward', t0)
        return out

    def track(self, lo, axis_name, shape3d):
        """tracker of the slab's own slices at their global positions (tracker.py:61-123), walking downwards -> {class_id:
        partial instances as flat arrays (sparse.StackMatcher.instances_packed; track order: first seen)}"""
        import time
        t0 = time.perf_counter()
        part = {c: self.sm[c].track_range(axis_name, shape3d, 0, self.n_own - 1, lo, packed=True) for c in self.labels}
        self._timed('track', t0)
        self.tail_s = self.host_s - self._pushed_s
        return part

    def finish(self, rank, aw, lo, axis_name, shape3d, group):
        """ONE contiguous slab per rank: neighbour exchange, the two chains, tracking -> the slab's partial trackers."""
        self.wait_pushed()
        has_prev, has_next = rank > 0, rank < aw - 1
        # boundary slices to the neighbours (raw entries: components as extracted); a chain, so the order below cannot
        # deadlock even where a send blocks until its receive is posted: the last / first rank only receives
        if has_next:
            _send_obj(self.last_entry, rank + 1, group)
        prev_entry = _recv_obj(rank - 1, group) if has_prev else None
        if has_prev:
            _send_obj(self.first_entry, rank - 1, group)
        next_entry = _recv_obj(rank + 1, group) if has_next else None
        if has_prev:
            self.push_prev_ghost(prev_entry)
        if has_next:
            self.push_next_ghost(next_entry)
        states = _recv_obj(rank - 1, group) if has_prev and self.things else None
        out_states = self.forward_chain((states if states is not None else {}) if has_prev else None)
        if has_next and self.things:
            _send_obj(out_states, rank + 1, group)
        states = _recv_obj(rank + 1, group) if has_next and self.things else None
        out_states = self.backward_chain((states if states is not None else {}) if has_next else None)
        if has_prev and self.things:
            _send_obj(out_states, rank - 1, group)
        return self.track(lo, axis_name, shape3d)


def _flat_part(p):
    """a part's class entry as flat arrays (labels, boxes (n,6), counts, starts, runs): the packed form as it is, or an
    instances dict flattened in its own order"""
    if not isinstance(p, dict):
        labels, boxes, counts, starts, runs = p
        return (np.asarray(labels, np.int64), np.asarray(boxes, np.int64).reshape(-1, 6), np.asarray(counts, np.int64),
                np.asarray(starts, np.int64), np.asarray(runs, np.int64))
    labels = np.fromiter((int(k) for k in p), np.int64, len(p))
    boxes = np.array([a['box'] for a in p.values()], np.int64).reshape(-1, 6)
    counts = np.fromiter((len(a['starts']) for a in p.values()), np.int64, len(p))
    cat = lambda key: (np.concatenate([np.asarray(a[key], np.int64) for a in p.values()]) if len(p) else np.zeros(0, np.int64))
    return labels, boxes, counts, cat('starts'), cat('runs')


def merge_partial_trackers(parts, axis_name, min_size=None, min_extent=None):
    """Per-slab partial trackers (rank order) -> one instances dict per class, as ONE tracker fed downwards over the whole
    stack would hold it (tracker.py:61-123): labels in first-seen order walking from the last slab to the first; xy / xz
    runs concatenated in that order (a slab's runs are final); yz runs -- re-encoded per object by ``finish()`` from ALL
    its voxels (tracker.py:111-120) -- joined across slabs: the slabs cut the x axis, so a run can continue in the next.

    Flat arithmetic: a part is (labels, boxes, counts, starts, runs) with an object's runs contiguous; the merge is a
    segment table (part, offset, count per object and slab) sorted by (first-seen rank of the label, walk order) and ONE
    threaded gather of all runs into label-major order (``emp_gather_segments_i64``: contiguous memcpys); the result dict
    holds views into the two gathered arrays.  At 4096^2 a block of 8 slices carries ~4 300 objects and 1.6 M runs: the
    per-object Python loop this replaces cost 2-5 ms per SLICE on rank 0 -- serial, proportional to the whole stack, the
    one cost of the multi-GPU job no rank count hides.

    ``min_size`` / ``min_extent``: the size filters the caller would apply next (filters.py: remove_small_objects drops an
    object whose runs sum to less than ``min_size`` voxels, remove_pancakes one whose box is thinner than ``min_extent``
    along any axis) applied HERE, on the flat arrays, so that no dict entry is built for an object that is dropped."""
    import ctypes as C
    from . import _abi
    classes = list(parts[0].keys()) if parts else []
    out = {}
    for c in classes:
        flat = [_flat_part(part[c]) for part in reversed(parts)]          # walk order: last slab first
        seg_label = np.concatenate([f[0] for f in flat]) if flat else np.zeros(0, np.int64)
        if len(seg_label) == 0:
            out[c] = {}
            continue
        seg_box = np.concatenate([f[1] for f in flat])
        seg_cnt = np.concatenate([f[2] for f in flat])
        seg_src = np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(flat)])
        seg_off = np.concatenate([np.concatenate([[0], np.cumsum(f[2])[:-1]]) if len(f[2]) else np.zeros(0, np.int64) for f in flat])
        # label -> rank of its first appearance in walk order
        uniq, first, inv = np.unique(seg_label, return_index=True, return_inverse=True)
        rank_of_uniq = np.empty(len(uniq), np.int64)
        rank_of_uniq[np.argsort(first, kind='stable')] = np.arange(len(uniq))
        seg_rank = rank_of_uniq[inv]
        order = np.argsort(seg_rank, kind='stable')                       # segments label-major, walk order inside a label
        cnt_o = np.ascontiguousarray(seg_cnt[order])
        off_o = np.ascontiguousarray(seg_off[order].astype(np.int64))
        src_o = np.ascontiguousarray(seg_src[order])
        out_off = np.concatenate([[0], np.cumsum(cnt_o)]).astype(np.int64)
        n_runs = int(out_off[-1])
        st, rn = np.empty(n_runs, np.int64), np.empty(n_runs, np.int64)
        keep = [(np.ascontiguousarray(f[3]), np.ascontiguousarray(f[4])) for f in flat]
        pa = (C.c_void_p * len(keep))(*[k[0].ctypes.data for k in keep])
        pb = (C.c_void_p * len(keep))(*[k[1].ctypes.data for k in keep])
        _abi.check(_abi.load().emp_gather_segments_i64(pa, pb, src_o.ctypes.data_as(C.c_void_p), off_o.ctypes.data_as(C.c_void_p),
                                                       cnt_o.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(C.c_void_p), len(order),
                                                       st.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p)),
                   'emp_gather_segments_i64')
        # per label: run range, merged box
        lab_sorted = seg_rank[order]
        lab_edges = np.concatenate([[0], np.flatnonzero(lab_sorted[1:] != lab_sorted[:-1]) + 1, [len(order)]])
        run_lo, run_hi = out_off[lab_edges[:-1]], out_off[lab_edges[1:]]
        box_o = seg_box[order]
        box_min = np.minimum.reduceat(box_o[:, :3], lab_edges[:-1], axis=0)
        box_max = np.maximum.reduceat(box_o[:, 3:], lab_edges[:-1], axis=0)
        labels_out = uniq[np.argsort(rank_of_uniq)]                          # labels in first-seen order
        if axis_name == 'yz' and len(flat) > 1:
            # per object: runs sorted by start, touching runs of neighbouring slabs joined
            obj_of_run = np.repeat(np.arange(len(labels_out)), run_hi - run_lo)
            o2 = np.lexsort((st, obj_of_run))
            st, rn, obj_of_run = st[o2], rn[o2], obj_of_run[o2]
            new = np.ones(len(st), bool)
            if len(st) > 1:
                new[1:] = (obj_of_run[1:] != obj_of_run[:-1]) | (st[1:] != st[:-1] + rn[:-1])
            heads = np.flatnonzero(new)
            ends = st + rn
            last = np.concatenate([heads[1:], [len(st)]]) - 1
            st, rn = st[heads], ends[last] - st[heads]
            cnt_obj = np.bincount(obj_of_run[heads], minlength=len(labels_out))
            run_hi = np.cumsum(cnt_obj)
            run_lo = run_hi - cnt_obj
        keep = np.ones(len(labels_out), bool)
        if min_size is not None:
            csum = np.concatenate([[0], np.cumsum(rn)])
            keep &= (csum[run_hi] - csum[run_lo]) >= min_size
        if min_extent is not None:
            keep &= (box_max - box_min).min(axis=1) >= min_extent
        acc = {}
        for k in np.flatnonzero(keep).tolist():
            lo, hi = int(run_lo[k]), int(run_hi[k])
            acc[int(labels_out[k])] = {'box': tuple(box_min[k].tolist() + box_max[k].tolist()), 'starts': st[lo:hi], 'runs': rn[lo:hi]}
        out[c] = acc
    return out


def slab_stack_inference(n_slices, backend, ks, group=None, host_group=None, match=None):
    """SPMD body of one axis on one rank.

    backend.forward(lo, hi, n_ahead)  -> (sem, stash): ``sem`` a tensor (hi-lo + n_ahead, ...) whose first hi-lo rows
                                         are the slab's probability maps (the rest is filled by the exchange)
    backend.median_inplace(sem, n_own, hist, n_ahead, first, last, ks)
                                      -> filters rows [0, n_own) of ``sem`` in place; ``hist``: the ``mid`` filtered
                                         maps before the slab (None on the first slab)
    backend.runs(sem_own, stash)      -> one entry per slice: {class: (runs (n,3) int64, id offset) | instance dict};
                                         ``backend.runs_iter`` (optional) yields the entries group by group instead
    match: dict(labels, thing_list, label_divisor, iou_thr, ioa_thr, width, axis_name, shape3d) -> the slab is matched
    and tracked HERE (``SlabMatcher``) and rank 0 gets the list of per-slab partial trackers in rank order together
    with every rank's host time {'parts': [...], 'host_s': [...]}; without it (round-2 behaviour, kept for the driver
    tests) rank 0 gets the per-slice entries of ALL slices in order.  Elsewhere None.  Tensors travel over ``group``
    (RCCL on GPUs: neighbour send / recv), host data -- boundary slices, matcher states, tracks -- over
    ``host_group`` (gloo; no pickling through device memory)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mid = (ks - 1) // 2
    aw = active_ranks(n_slices, world, ks)
    bounds = slab_bounds(n_slices, aw) + [(n_slices, n_slices)] * (world - aw)
    lo, hi = bounds[rank]
    n_own = hi - lo
    per_slice = []
    import time
    t_start = time.perf_counter()
    if n_own > 0:
        has_prev, has_next = rank > 0, rank < aw - 1
        n_ahead = mid if has_next else 0
        sem, stash = backend.forward(lo, hi, n_ahead)
        reqs = []
        if mid and has_prev:          # my first raw maps are the look-ahead of the previous rank
            reqs.append(_isend(sem[:mid], rank - 1, group))
        if mid and has_next:          # straight into the tail of the slab buffer
            _recv(sem[n_own:], rank + 1, group)
        hist = None
        if mid and has_prev:          # carry: filtered tail of the previous slab (ripples down the ranks)
            hist = torch.empty_like(sem[:mid])
            _recv(hist, rank - 1, group)
        for r in reqs:
            r.wait()                  # the median below overwrites the raw maps that send is reading
        backend.median_inplace(sem, n_own, hist, n_ahead, rank == 0, rank == aw - 1, ks)
        if mid and has_next:
            reqs.append(_isend(sem[n_own - mid:n_own], rank + 1, group))
        hg = host_group if host_group is not None else group
        if match is None:
            per_slice = backend.runs(sem[:n_own], stash)
        else:
            sm = SlabMatcher(match['labels'], match['thing_list'], match['label_divisor'], match['iou_thr'], match['ioa_thr'],
                             match['width'], head=rank == 0)
            if hasattr(backend, 'runs_iter'):      # host: push + pair tables of a group while the GPU extracts the next
                for entries in backend.runs_iter(sem[:n_own], stash):
                    sm.push(entries)
            else:
                sm.push(backend.runs(sem[:n_own], stash))
            gpu_s = time.perf_counter() - t_start       # forward + exchange + median + run extraction of the slab
            part = sm.finish(rank, aw, lo, match['axis_name'], match['shape3d'], hg)
            per_slice = {'part': part, 'host_s': sm.host_s, 'tail_s': sm.tail_s, 'gpu_s': gpu_s, 'slices': n_own,
                         'phases': dict(sm.phase_s)}
        for r in reqs:
            r.wait()
    elif match is not None:
        per_slice = None
    hg = host_group if host_group is not None else group
    if match is None:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(per_slice, gathered, dst=0, group=hg)
        return [s for part in gathered for s in part] if rank == 0 else None
    shm = ranks_share_host(hg) if world > 1 else False      # (collective on the group's first use; every rank is here)
    # per-slab tracks to rank 0: the small tables (labels, boxes, counts, timings) as one object, the run lists as the int64
    # tensors they are -- nothing of size is pickled, and rank 0's own slab does not travel at all
    if rank != 0:
        _send_part(per_slice, 0, hg, shm)
        return None
    gathered = [per_slice] + [_recv_part(r, hg) for r in range(1, world)]
    live = [g for g in gathered if g is not None]
    return {'parts': [g['part'] for g in live], 'host_s': [g['host_s'] for g in live],
            'timing': [{k: g[k] for k in ('host_s', 'tail_s', 'gpu_s', 'slices', 'phases')} for g in live]}


def block_slices(backend):
    """slices per block of the interleaved schedule: EMP_MG_BLOCK (0 = one contiguous slab per rank, the round-3a schedule),
    else what the backend forwards at a time (``backend.block_slices``), else 16"""
    e = os.environ.get('EMP_MG_BLOCK')
    if e is not None:
        return max(0, int(e))
    return int(getattr(backend, 'block_slices', 16))


_CHAIN_GROUPS = {}
_GATHER_GROUPS = {}
_CARRY_GROUPS = {}
_CHAIN_EXEC = None


def _chain_executor():
    """ONE worker thread per process for the block chains: the chain job of an axis, then its gather to rank 0, then the
    next axis' chain ... in submission order -- every rank submits the same sequence, so the messages of consecutive axes
    on the chain / gather groups cannot interleave although an axis' backward pass may still run while the next axis'
    GPU phase is under way (``defer``)."""
    global _CHAIN_EXEC
    if _CHAIN_EXEC is None:
        from concurrent.futures import ThreadPoolExecutor
        _CHAIN_EXEC = ThreadPoolExecutor(max_workers=1, thread_name_prefix='emp-block-chain')
    return _CHAIN_EXEC


def _default_gather_group(group):
    """a gloo group of its own for the per-block tracks' way to rank 0 when that gather is deferred behind the caller
    (it must not share a group with the caller's halo / carry messages of the NEXT axis)"""
    key = id(group) if group is not None else None
    g = _GATHER_GROUPS.get(key)
    if g is None:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        g = _GATHER_GROUPS[key] = dist.new_group(ranks=ranks, backend='gloo', timeout=chain_timeout())
    return g



def _default_chain_group(group):
    """ONE gloo group per parent group for the block chain's messages, created on first use and kept (a new group per
    call -- one per axis -- would leak a process group each time; every rank calls this in the same order)"""
    key = id(group) if group is not None else None
    g = _CHAIN_GROUPS.get(key)
    if g is None:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        g = _CHAIN_GROUPS[key] = dist.new_group(ranks=ranks, backend='gloo', timeout=chain_timeout())
    return g


def _default_carry_group(group):
    """a group of the tensor backend for the filtered carries alone.  The look-ahead maps travel as grouped ring shifts on
    ``group``, the carries one by one down the ranks; on a communicator of their own the two kinds never have to agree on
    ONE order per pair of ranks (with W = 2 both neighbours are the same rank, and a shift of round k + 1 is posted before
    the carry of round k is received)."""
    key = id(group) if group is not None else None
    g = _CARRY_GROUPS.get(key)
    if g is None:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        kw = {'timeout': chain_timeout()} if dist.get_backend(group) == 'gloo' else {}
        g = _CARRY_GROUPS[key] = dist.new_group(ranks=ranks, backend=dist.get_backend(group), **kw)
        if dist.get_backend(group) == 'nccl':
            # a grouped point-to-point call must not be the first communication on a group unless every rank takes part
            # in it (a rank without a block posts no shift): bring the communicator up here, where every rank is
            dist.all_reduce(torch.zeros(1, device='cuda'), group=group)
    return g


class _Done:
    """a finished result behind the Future interface ``stack_inference(defer=True)`` returns"""

    def __init__(self, value):
        self.value = value

    def result(self, timeout=None):
        return self.value

    def done(self):
        return True


def stack_inference(n_slices, backend, ks, group=None, host_group=None, match=None, chain_group=None, defer=False):
    """the schedule the engine runs: block-interleaved when the slices are matched on the ranks, contiguous slabs otherwise.
    ``defer``: return when this rank's GPU phase is through -- a Future whose ``result()`` is what the call returns
    otherwise (rank 0: the parts, elsewhere None): the rest of the forward chain, the backward chain, the tracking and the
    gather to rank 0 finish on the chain thread, behind the caller (the next axis' GPU phase, in the ortho-plane job)."""
    blk = block_slices(backend) if match is not None else 0
    if blk <= 0:
        res = slab_stack_inference(n_slices, backend, ks, group, host_group, match)
        return _Done(res) if defer else res
    blk = min(blk, max(1, n_slices // dist.get_world_size(group)))      # a short stack: still a block for every rank
    if chain_group is None:          # the chain thread's messages must not share a group with this thread's
        chain_group = _default_chain_group(group)
    if defer and (host_group is None or host_group is group):
        host_group = _default_gather_group(group)
    return block_stack_inference(n_slices, backend, ks, match, blk, group, host_group, chain_group, defer)


def block_bounds(n_slices, block, ks):
    """near-equal blocks of about ``block`` slices, none shorter than the median's look-ahead"""
    mid = (ks - 1) // 2
    b = max(int(block), mid, 1)
    return slab_bounds(n_slices, max(1, n_slices // b))


def ring_shift_plan(rank, world, n_blocks):
    """The look-ahead traffic of the block-interleaved schedule as lock-step shifts of the ring of ranks.  Rank r owns
    blocks r, r + W, ...; shift s is posted by every rank at the same point of its program -- right after the forward of
    its block of round s (r + sW), or where that forward would be if the rank has no such block.  In shift s a rank
    sends the first raw maps of its block of round s to the owner of the block before it (its left neighbour) and
    receives what its right neighbour sends: the look-ahead of the block before the neighbour's, which is this rank's
    block of round s -- or, on the last rank (whose right neighbour, rank 0, is a round ahead), of round s - 1.

    -> per shift {'send': [(block whose first maps go out, destination rank)], 'recv': [(block whose look-ahead comes
    in, source rank)]}; every send of a shift has its receive in the SAME shift of the destination, which is what lets
    the operations of a shift be one RCCL group."""
    W, NB = int(world), int(n_blocks)
    if W <= 1:
        return []
    right = (rank + 1) % W
    plan = []
    s = 0
    while True:
        b, bR = rank + s * W, right + s * W
        ent = {'send': [(b, (b - 1) % W)] if 0 < b < NB else [], 'recv': [(bR - 1, right)] if 0 < bR < NB else []}
        if b >= NB and bR >= NB:
            break
        plan.append(ent)
        s += 1
    while plan and not plan[-1]['send'] and not plan[-1]['recv']:
        plan.pop()
    return plan


def block_stack_inference(n_slices, backend, ks, match, block, group=None, host_group=None, chain_group=None, defer=False):
    """SPMD body of one axis on one rank, BLOCK-INTERLEAVED: the stack is cut into blocks of about ``block`` slices (one
    forward batch) and rank r owns blocks r, r + W, r + 2W, ...  Same results as ``slab_stack_inference`` (one contiguous
    slab per rank = W blocks); what changes is WHEN things can run:

      * the forward matching chain is serial over the slices of the whole stack (SlabMatcher); with one slab per rank it
        starts on rank r only when ranks 0..r-1 are through theirs, i.e. behind everybody's GPU phase.  Here block b's chain
        step runs as soon as block b-1's has (a chain thread per rank, host messages on ``chain_group``) while the GPUs
        are already on the next round of blocks: the chain costs one block per block, hidden behind the GPU work as long
        as W chain steps fit into one round (0.08 ms against 0.9 ms per 1024^2 slice: W <= 11);
      * only the backward chain (fresh target, no new labels: the cheap one) and the tracking of the last blocks stay
        behind the last round.

    GPU side of a round (block b of this rank; left / right = the owners of blocks b-1 / b+1, a ring): the forward of the
    NEXT round is enqueued first and its first raw maps go to the left neighbour at once -- the look-ahead a block's median
    waits for never waits for a forward that has not been started (the ring's wrap-around would otherwise idle the last
    rank for a round); then filtered carry in, median in place, carry out, voting / merge / run extraction, push to the
    block's matcher.  The look-ahead maps travel as lock-step shifts of the ring (``ring_shift_plan``: send left + receive
    from the right as ONE RCCL group on ``group``, posted by every rank at the same point of its program), the carries one
    by one down the ranks on a communicator of their own (``_default_carry_group``) where a pair of ranks sees them in one
    order on both sides: no rank ever queues a send in front of the receive that its peer's send is waiting for, which is
    what stream-ordered RCCL point-to-point calls need (gloo's buffered sends, the CPU tests' transport, would forgive it).

    -> on rank 0: {'parts': per-block partial trackers in block order, 'host_s', 'timing'}, elsewhere None."""
    import threading
    import time
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mid = (ks - 1) // 2
    bounds = block_bounds(n_slices, block, ks)
    NB = len(bounds)
    owner = lambda b: b % world
    mine = [b for b in range(NB) if owner(b) == rank]
    hg = host_group if host_group is not None else group
    cg = chain_group if chain_group is not None else hg
    carry_group = _default_carry_group(group) if world > 1 and mid else group
    # do the ranks share a host?  (collective on the group's FIRST use -- before this call's chain job exists, and with no
    # deferred job of an earlier axis on this group: those were submitted after the answer was cached)
    shm = ranks_share_host(hg) if world > 1 else False
    sms = [SlabMatcher(match['labels'], match['thing_list'], match['label_divisor'], match['iou_thr'], match['ioa_thr'],
                       match['width'], head=(b == 0)) for b in mine]
    pushed = [threading.Event() for _ in mine]
    parts = [None] * len(mine)
    err = []
    t_start = time.perf_counter()
    t_chain_end = [t_start]

    def chain():
        """this rank's blocks through the forward chain in block order, then through the backward chain in reverse.  What a
        hop of the chain waits for is small (the O(objects) state of one slice): the boundary slices themselves -- the
        ghosts a block's pair tables need, 360 KB at 1024^2 -- leave as soon as a block is extracted (non-blocking sends)
        and are pushed while the state is still on its way.  Per block the messages to a neighbour go out in one fixed
        order (first entry leftwards, last entry rightwards, forward state rightwards; in the second phase the backward
        state leftwards) and are received in that order, so that one group carries them without tags (W = 2: both
        neighbours are the same rank)."""
        try:
            sent = []
            fwd_state = None                      # W = 1: handed from block to block here
            def drain_next_ghost(k):              # block mine[k]'s next ghost = the first entry of block mine[k] + 1
                b = mine[k]
                if b + 1 < NB:
                    e = sms[k + 1].first_entry if owner(b + 1) == rank else _recv_pickled(owner(b + 1), cg)
                    sms[k].push_next_ghost(e)
            for k, b in enumerate(mine):
                pushed[k].wait()
                if err:
                    return
                sm = sms[k]
                sm.wait_pushed()
                if k > 0 and owner(mine[k - 1] + 1) == rank:
                    drain_next_ghost(k - 1)       # (this very block: its first entry exists now)
                if b > 0 and owner(b - 1) != rank:
                    sent.append(_isend_pickled(sm.first_entry, owner(b - 1), cg))
                if b + 1 < NB and owner(b + 1) != rank:
                    sent.append(_isend_pickled(sm.last_entry, owner(b + 1), cg))
                if k > 0 and owner(mine[k - 1] + 1) != rank:
                    drain_next_ghost(k - 1)
                if b > 0:
                    if owner(b - 1) == rank:
                        sm.push_prev_ghost(sms[k - 1].last_entry)
                        states = fwd_state
                    else:
                        sm.push_prev_ghost(_recv_pickled(owner(b - 1), cg))
                        states = _recv_pickled(owner(b - 1), cg)
                    out = sm.forward_chain(states)
                else:
                    out = sm.forward_chain()
                if b + 1 < NB:
                    if owner(b + 1) == rank:
                        fwd_state = out
                    else:
                        sent.append(_isend_pickled(out, owner(b + 1), cg))
            if mine and owner(mine[-1] + 1) != rank:
                drain_next_ghost(len(mine) - 1)
            bwd_state = None
            for k in range(len(mine) - 1, -1, -1):
                b, sm = mine[k], sms[k]
                if b + 1 < NB:
                    out = sm.backward_chain(bwd_state if owner(b + 1) == rank else _recv_pickled(owner(b + 1), cg))
                else:
                    out = sm.backward_chain()
                if b > 0:
                    if owner(b - 1) == rank:
                        bwd_state = out
                    else:
                        sent.append(_isend_pickled(out, owner(b - 1), cg))
                parts[k] = sm.track(bounds[b][0], match['axis_name'], match['shape3d'])
            for w in sent:
                w.wait()
            t_chain_end[0] = time.perf_counter()
        except Exception:       # noqa: BLE001 -- re-raised by the main thread
            err.append(traceback.format_exc())
            for e in pushed:
                e.set()

    th = _chain_executor().submit(chain)      # behind the previous axis' chain + gather, if those are still running
    reqs = []
    try:
        plan = ring_shift_plan(rank, world, NB) if mid else []
        la = {}                                              # block -> the shift that brings its look-ahead
        posted = {}                                          # round -> the shift that sends that block's first raw maps

        def shift(s):
            """shift ``s`` of the ring (ring_shift_plan), posted by every rank right after the forward of its block of
            round ``s``: that block's first raw maps to the owner of the block before it, and what the right neighbour
            sends at this very point -- the look-ahead of the block before ITS block -- into that block's buffer"""
            if s >= len(plan):
                return
            sends = [(fw[(b - rank) // world][0][:mid], dst) for b, dst in plan[s]['send']]
            recvs = []
            for b, src in plan[s]['recv']:
                recvs.append((fw[(b - rank) // world][0][bounds[b][1] - bounds[b][0]:], src))
            if sends or recvs:
                sh = _ring_shift(sends, recvs, group)
                reqs.append(sh)
                posted[s] = sh
                for b, _ in plan[s]['recv']:
                    la[b] = sh

        def forward(k):
            b = mine[k]
            lo, hi = bounds[b]
            fw[k] = backend.forward(lo, hi, mid if b + 1 < NB else 0)
            shift(k)

        fw = {}
        if mine:
            forward(0)
        carry = None                                         # W = 1: the filtered tail of the block before, kept here
        for k, b in enumerate(mine):
            if k + 1 < len(mine):
                forward(k + 1)
            else:
                shift(k + 1)       # no block of mine in the next round: the last rank may still be owed a look-ahead
            sem, stash = fw.pop(k)
            lo, hi = bounds[b]
            n_own = hi - lo
            has_prev, has_next = b > 0, b + 1 < NB
            if mid and has_next:
                if owner(b + 1) == rank:
                    sem[n_own:].copy_(fw[k + 1][0][:mid])
                else:
                    la.pop(b).wait()
            if k in posted:
                posted.pop(k).wait()      # the median below overwrites the maps that shift is sending
            hist = None
            if mid and has_prev:
                if owner(b - 1) == rank:
                    hist = carry
                else:
                    hist = torch.empty_like(sem[:mid])
                    _recv(hist, owner(b - 1), carry_group)
            backend.median_inplace(sem, n_own, hist, mid if has_next else 0, b == 0, b == NB - 1, ks)
            if mid and has_next:
                if owner(b + 1) == rank:
                    carry = sem[n_own - mid:n_own].clone()
                else:
                    reqs.append(_isend(sem[n_own - mid:n_own], owner(b + 1), carry_group))
            if hasattr(backend, 'runs_iter'):
                for entries in backend.runs_iter(sem[:n_own], stash):
                    sms[k].push(entries)
            else:
                sms[k].push(backend.runs(sem[:n_own], stash))
            pushed[k].set()
            if err:
                break
        gpu_s = time.perf_counter() - t_start
        if not err:      # after a chain failure the peers may never post the matching receives
            for r in reqs:
                r.wait()
    except BaseException:
        # this rank's GPU side failed: report it now -- the chain thread may sit in a receive that its peers will never
        # answer (it is a daemon thread; the caller ends the ranks)
        err.append('main thread failed')
        for e in pushed:
            e.set()
        raise
    for e in pushed:
        e.set()

    def finish():
        """chain through, then the per-block tracks to rank 0 (host group); on the caller's thread, or -- ``defer`` -- as the
        chain thread's next job"""
        th.result()
        if err:
            raise RuntimeError('block chain failed on rank %d:\n%s' % (rank, err[0]))
        res = None
        if mine:
            ph = {key: sum(sm.phase_s[key] for sm in sms) for key in ('ghosts', 'forward', 'backward', 'track')}
            res = {'blocks': [{'block': b, 'part': p} for b, p in zip(mine, parts)],
                   'host_s': sum(sm.host_s for sm in sms), 'tail_s': max(0.0, t_chain_end[0] - t_start - gpu_s), 'gpu_s': gpu_s,
                   'slices': sum(bounds[b][1] - bounds[b][0] for b in mine), 'phases': ph}
        if rank != 0:
            _send_blocks(res, 0, hg, shm)
            return None
        gathered = [res] + [_recv_blocks(r, hg) for r in range(1, world)]
        live = [g for g in gathered if g is not None]
        by_block = {blk['block']: blk['part'] for g in live for blk in g['blocks']}
        return {'parts': [by_block[b] for b in range(NB)], 'host_s': [g['host_s'] for g in live],
                'timing': [{k: g[k] for k in ('host_s', 'tail_s', 'gpu_s', 'slices', 'phases')} for g in live]}

    if defer:
        return _chain_executor().submit(finish)
    return finish()


def _send_blocks(res, dst, group, shm=False):
    if res is None:
        _send_obj(None, dst, group)
        return
    _send_obj({k: v for k, v in res.items() if k != 'blocks'} | {'n_blocks': len(res['blocks'])}, dst, group)
    for blk in res['blocks']:
        _send_part({'block': blk['block'], 'part': blk['part']}, dst, group, shm)


def _recv_blocks(src, group):
    res = _recv_obj(src, group)
    if res is None:
        return None
    res['blocks'] = [_recv_part(src, group) for _ in range(res.pop('n_blocks'))]
    return res


class HipSlabBackend:
    """The arithmetic of one rank on its MI355X (through ``Engine3d``'s engine and the C ABI)."""

    def __init__(self, engine3d, volume, axis):
        self.e3, self.eng = engine3d, engine3d.engine
        self.volume, self.axis = volume, axis
        e3 = engine3d
        if e3.inference_scale != 1:
            # the reference's multi-GPU worker calls infer(image) with the default render_steps and then upsamples
            # the cells (multigpu.py:83-87): for inference_scale > 1 the two no longer have one size and its
            # get_panoptic_seg raises -- this path exists at native scale only
            raise Exception('MultiGPU inference runs at inference_scale = 1 only (empanada_napari/multigpu.py:83-87)')
        shp = [s for i, s in enumerate(volume.shape) if i != axis]
        self.size = (int(shp[0]), int(shp[1]))
        pf = self.eng.padding_factor
        self.pad_to = (-(-self.size[0] // pf) * pf, -(-self.size[1] // pf) * pf)
        vdt = np.dtype(volume.dtype)
        self.raw = vdt in (np.dtype(np.uint8), np.dtype(np.uint16))
        if self.raw:
            from .preprocess import normalize_params
            self.sub, self.mul = normalize_params(e3.preprocessor.mean, e3.preprocessor.std, np.iinfo(vdt).max)

    @property
    def block_slices(self):
        """slices per block of the interleaved schedule: one forward batch, but at least 8 slices -- a block costs two
        ghost slices and two chain hops whatever its length (a 4096^2 slice is a forward batch of its own)"""
        return max(8, int(self.e3.slice_batch(self.pad_to)))

    def _block(self, i0, i1):
        """slices [i0, i1) along the axis as an (n,1,h,w) tensor on the model's device"""
        v, dev = self.volume, self.eng.model.device
        if hasattr(v, 'block'):                        # synth.ProceduralVolume: synthesised where it is used
            return v.block(self.axis, i0, i1, dev)[:, None]
        idx = tuple(slice(i0, i1) if a == self.axis else slice(None) for a in range(3))
        blk = np.moveaxis(np.asarray(v[idx]), self.axis, 0)
        return torch.from_numpy(np.ascontiguousarray(blk))[:, None].to(dev, non_blocking=True)

    @torch.no_grad()
    def forward(self, lo, hi, n_ahead):
        from .engines import factor_pad, logits_to_prob
        e3, eng = self.e3, self.eng
        dev = eng.model.device
        n_own = hi - lo
        bs = e3.slice_batch(self.pad_to)
        sem = ctr = off = None
        for i0 in range(lo, hi, bs):
            i1 = min(hi, i0 + bs)
            xb = self._block(i0, i1)
            if self.raw:   # raw integers go up, normalisation + factor_pad run inside the stem kernel
                mo = eng.model(xb, 2, interpolate_ins=not eng.coarse_boundaries, sub=float(self.sub), mul=float(self.mul),
                               pad_to=self.pad_to)
            else:
                imgs = [e3.preprocessor(np.asarray(x[0].cpu()))['image'] for x in xb]
                mo = eng.model(eng.to_model_device(factor_pad(torch.stack(imgs), eng.padding_factor)), 2,
                               interpolate_ins=not eng.coarse_boundaries)
            if sem is None:      # ONE slab-sized buffer per head: the median runs in place, nothing is concatenated
                sem = torch.empty((n_own + n_ahead,) + tuple(mo['sem_logits'].shape[1:]), dtype=torch.float32, device=dev)
                ctr = torch.empty((n_own,) + tuple(mo['ctr_hmp'].shape[1:]), dtype=torch.float32, device=dev)
                off = torch.empty((n_own,) + tuple(mo['offsets'].shape[1:]), dtype=torch.float32, device=dev)
            logits_to_prob(mo['sem_logits'], out=sem[i0 - lo:i1 - lo])
            ctr[i0 - lo:i1 - lo].copy_(mo['ctr_hmp'])
            off[i0 - lo:i1 - lo].copy_(mo['offsets'])
        return sem, (ctr, off)

    @torch.no_grad()
    def median_inplace(self, sem, n_own, hist, n_ahead, first, last, ks):
        from . import _abi
        mid = (ks - 1) // 2
        if mid == 0:
            return
        a = mid if first else 0                 # the first / last mid slices of the STACK stay unfiltered
        b = n_own - (mid if last else 0)
        if b <= a:
            return
        h = sem[:mid] if first else hist        # on the first slab the history is its own (unfiltered) head
        raw = sem[a:]
        assert raw.shape[0] >= (b - a) + mid and h.shape[0] == mid and h.is_contiguous() and raw.is_contiguous()
        _abi.check(self.eng.lib.emp_median_recursive(_abi.ptr(h), _abi.ptr(raw), raw.shape[0], ks, b - a, _abi.ptr(raw),
                                                     sem[0].numel(), _abi.stream_ptr(sem.device)), 'emp_median_recursive')

    @torch.no_grad()
    def runs_iter(self, sem, stash):
        """per-slice entries in launch groups of 64 slices (the caller's matcher works on a group while the GPU extracts
        the next one)"""
        from . import sparse
        e3, eng = self.e3, self.eng
        ctr, off = stash
        h, w = self.size
        for i0 in range(0, sem.shape[0], 64):
            sl = slice(i0, i0 + 64)
            cells, _, _, kmax = eng.instance_cells_int(ctr[sl], off[sl], 1)
            pan = eng.panoptic_merge_int(sem[sl], cells, kmax)[:, :h, :w]
            per_label = sparse.pan_stack_to_runs(pan, e3.labels, e3.label_divisor, e3.thing_list, force_connected=True)
            yield [{label: (rl[j], o) for label, (rl, o) in per_label.items()} for j in range(pan.shape[0])]

    def runs(self, sem, stash):
        return [e for group in self.runs_iter(sem, stash) for e in group]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _default_backend_factory(model_config, engine_kwargs, rank):
    """-> callable (volume, axis) -> backend, on this rank's GPU (``engine_kwargs['devices']``: rank -> device index)"""
    from .inference import Engine3d
    engine_kwargs = dict(engine_kwargs)
    devices = engine_kwargs.pop('devices', None)
    dev = torch.device('cuda', devices[rank] if devices else rank)
    torch.cuda.set_device(dev)
    e3 = Engine3d(model_config, device=dev, **engine_kwargs)
    return lambda volume, axis: HipSlabBackend(e3, volume, axis)


def _rank_main(rank, world, port, dist_backend, model_config, engine_kwargs, backend_factory, cmd_q, res_q):
    """One worker process = one rank = one GPU; serves ``infer_on_axis`` calls until told to stop."""
    try:
        os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if dist_backend == 'nccl':
            devices = engine_kwargs.get('devices')
            dev = torch.device('cuda', devices[rank] if devices else rank)
            torch.cuda.set_device(dev)
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            host_group = dist.new_group(backend='gloo')
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)
            host_group = None
        # the block chain's own messages (its thread runs beside the GPU loop); a receive whose peer failed times out
        chain_group = dist.new_group(backend='gloo', timeout=chain_timeout())
        make = (backend_factory or _default_backend_factory)(model_config, engine_kwargs, rank)
        res_q.put(('ready', rank, None))
        while True:
            cmd = cmd_q.get()
            if cmd[0] == 'stop':
                break
            _, volume, axis_name, ks, match = cmd
            if isinstance(volume, torch.Tensor):       # a numpy volume travels as a shared-memory tensor
                volume = volume.numpy()
            axis = {'xy': 0, 'xz': 1, 'yz': 2}[axis_name]
            segs = stack_inference(volume.shape[axis], make(volume, axis), ks, None, host_group, match, chain_group)
            res_q.put(('done', rank, segs))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        res_q.put(('error', rank, traceback.format_exc()))


class MultiGPUEngine3d:
    """``empanada_napari.multigpu.MultiGPUEngine3d`` (multigpu.py:121-260): same constructor arguments, ``dtype``,
    ``create_trackers`` / ``create_panoptic_stack`` and ``infer_on_axis(volume, axis_name) -> (stack, trackers)``.

    Differences behind the same results (module docstring): contiguous z-slabs instead of round-robin slices + dense
    all-gathers; the ranks are started by the first ``infer_on_axis`` (``mp.spawn``, one process per GPU, as
    multigpu.py:214-238) and KEPT for the next axes instead of reloading the model per call (``close()`` ends them);
    forward matching, backward matching and tracking run in the calling process in C++.  Quirks kept: post-processing
    uses ``stuff_area = 32`` whatever the constructor was given (patterns.py:258,289), and only native scale works
    (multigpu.py:83-87).  ``model_config['model']`` must be picklable in spawn mode (a TorchScript path or a state dict,
    as the reference's ``model_url``).

    Under an SPMD launch (``torchrun``: ``torch.distributed`` already initialised) nothing is spawned: every rank
    constructs the engine and calls ``infer_on_axis``; rank 0 returns ``(stack, trackers)``, the others ``(None, None)``.

    Extra keyword arguments (not in the reference): ``world_size`` (default ``torch.cuda.device_count()``),
    ``dist_backend`` ('nccl' = RCCL; 'gloo' for the CPU tests -- device maps are then staged through the host),
    ``backend_factory`` (the per-rank arithmetic; tests plug the oracle in), ``devices`` (rank -> device index; default
    rank r on cuda:r), ``batch_size``, ``group``."""
    MIN_WORLD = 2
    MULTIGPU_STUFF_AREA = 32      # patterns.py:258,289

    def __init__(self, model_config, inference_scale=1, label_divisor=1000, median_kernel_size=5, stuff_area=64,
                 void_label=0, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.3, force_connected=True,
                 min_size=500, min_extent=4, fine_boundaries=False, semantic_only=False, store_url=None,
                 chunk_size=(256, 256, 256), save_panoptic=False, batch_size=None, group=None, world_size=None,
                 dist_backend='nccl', backend_factory=None, devices=None):
        self.spmd = dist.is_initialized()
        if self.spmd:
            world = dist.get_world_size(group)
        else:
            world = int(world_size) if world_size else torch.cuda.device_count()     # no GPU context in this process
        if world < self.MIN_WORLD:
            raise Exception('MultiGPU inference requires 2 or more GPUs! Run torch.cuda.device_count()')  # multigpu.py:143-144
        self.world, self.group, self.dist_backend, self.backend_factory = world, group, dist_backend, backend_factory
        self.model_config = model_config
        self.labels = model_config['labels']
        self.thing_list = [] if semantic_only else model_config['thing_list']
        self.label_divisor, self.ks = label_divisor, median_kernel_size
        self.inference_scale = inference_scale
        self.min_size, self.min_extent = min_size, min_extent
        self.merge_iou_thr = self.merge_ioa_thr = 0.25
        self.save_panoptic, self.chunk_size = save_panoptic, chunk_size
        self.axes = {'xy': 0, 'xz': 1, 'yz': 2}
        self.dtype = np.int32
        self.engine_kwargs = dict(
            inference_scale=inference_scale, label_divisor=label_divisor, median_kernel_size=median_kernel_size,
            stuff_area=self.MULTIGPU_STUFF_AREA, void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
            confidence_thr=confidence_thr, force_connected=force_connected, min_size=min_size, min_extent=min_extent,
            fine_boundaries=fine_boundaries, semantic_only=semantic_only, batch_size=batch_size)
        if devices is not None:       # rank -> device index (default: rank r on cuda:r)
            self.engine_kwargs['devices'] = list(devices)
        from .inference import _open_zarr
        # mode 'w' deletes what is there: under an SPMD launch only rank 0 -- the one rank that writes the stack -- opens it
        opens = store_url is not None and (not self.spmd or dist.get_rank(group) == 0)
        self.zarr_store = _open_zarr(store_url, mode='w') if opens else None
        self._procs = None
        self._make = None
        self._host_group = None
        self._chain_group = None

    # ---- the reference's helpers (multigpu.py:186-212) ----
    def create_trackers(self, shape3d, axis_name):
        from . import sparse
        return [sparse.InstanceTracker(label, self.label_divisor, shape3d, axis_name) for label in self.labels]

    def create_panoptic_stack(self, axis_name, shape3d):
        if self.zarr_store is not None and self.save_panoptic:
            return self.zarr_store.create_dataset(f'panoptic_{axis_name}', shape=shape3d, dtype=self.dtype,
                                                  chunks=self.chunk_size, overwrite=True)
        if self.save_panoptic:
            return np.zeros(shape3d, dtype=self.dtype)
        return None

    # ---- rank processes (spawn mode) ----
    def _start(self):
        import torch.multiprocessing as mp
        m = self.model_config.get('model')
        if not isinstance(m, (str, dict)) and self.backend_factory is None:
            raise TypeError("MultiGPUEngine3d starts one process per GPU: model_config['model'] must be a TorchScript "
                            'path or a state dict (picklable), not a built engine')
        ctx = mp.get_context('spawn')
        self._cmd = [ctx.Queue() for _ in range(self.world)]
        self._res = ctx.Queue()
        port = _free_port()
        import runpy
        here = os.path.dirname(os.path.abspath(__file__))
        self._procs = [ctx.Process(target=runpy.run_path, daemon=True, args=(os.path.join(here, '_rank_worker.py'),),
                                   kwargs={'init_globals': {
                                       'PKG_INIT': os.path.join(here, '__init__.py'),
                                       'ARGS': (r, self.world, port, self.dist_backend, self.model_config,
                                                self.engine_kwargs, self.backend_factory, self._cmd[r], self._res)}})
                       for r in range(self.world)]
        for p in self._procs:
            p.start()
        self._collect('ready')

    def _collect(self, what):
        import queue
        import time
        out = {}
        # overall deadline (EMP_MG_TIMEOUT seconds, default one hour per call): ranks that are all alive but blocked --
        # e.g. in a recv whose peer never sends -- must surface as an error, not as a hang of the caller
        deadline = time.monotonic() + float(os.environ.get('EMP_MG_TIMEOUT', '3600'))
        while len(out) < self.world:
            try:
                kind, rank, payload = self._res.get(timeout=5.0)
            except queue.Empty:     # make sure nobody died without a message
                dead = [i for i, p in enumerate(self._procs) if not p.is_alive() and i not in out]
                if dead:
                    self.close(kill=True)
                    raise RuntimeError(f'multi-GPU rank process(es) {dead} exited unexpectedly')
                if time.monotonic() > deadline:
                    self.close(kill=True)
                    raise RuntimeError(f'multi-GPU ranks did not answer within EMP_MG_TIMEOUT (waiting for {what!r} from '
                                       f'{sorted(set(range(self.world)) - set(out))})')
                continue
            if kind == 'error':
                self.close(kill=True)
                raise RuntimeError(f'multi-GPU rank {rank} failed:\n{payload}')
            assert kind == what, (kind, what)
            out[rank] = payload
        return out

    def close(self, kill=False):
        procs, self._procs = self._procs, None
        self.__dict__.pop('_shm', None)      # the shared-memory copy of the last volume
        if not procs:
            return
        if not kill:
            for q in self._cmd:
                q.put(('stop',))
        for p in procs:
            p.join(timeout=0.1 if kill else 30)
            if p.is_alive():
                p.terminate()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inference ----
    def _segs_spawn(self, volume, axis_name):
        if self._procs is None:
            self._start()
        payload = volume
        if isinstance(volume, np.ndarray):
            # one copy, mapped by every rank.  The shared-memory ALLOCATION is kept for the following calls with the same
            # shape / dtype (the widget calls xy, xz, yz on one array: _volume_inference.py:336-348), the CONTENT is copied
            # in on every call: identity of the caller's array proves nothing -- ids and buffer addresses are reused once
            # an array is freed, and the same array may have been edited in place (round 3 keyed a cache on them and could
            # segment the previous volume).  A memcpy costs what a content check would.
            shm = self.__dict__.get('_shm')
            if shm is None or tuple(shm.shape) != tuple(volume.shape) or shm.numpy().dtype != volume.dtype:
                shm = self._shm = torch.empty(tuple(volume.shape), dtype=torch.from_numpy(np.empty(0, volume.dtype)).dtype).share_memory_()
            np.copyto(shm.numpy(), volume)
            payload = shm
        for q in self._cmd:
            q.put(('axis', payload, axis_name, self.ks, self._match_desc(volume.shape, axis_name)))
        return self._collect('done')[0]

    def _segs_spmd(self, volume, axis_name, defer=False):
        if self._make is None:
            rank = dist.get_rank(self.group)
            factory = self.backend_factory or _default_backend_factory
            local = int(os.environ.get('LOCAL_RANK', rank))
            self._make = factory(self.model_config, self.engine_kwargs, local)
            if dist.get_backend(self.group) == 'nccl':
                self._host_group = dist.new_group(backend='gloo')
            self._chain_group = dist.new_group(backend='gloo', timeout=chain_timeout())
        # every deferred future of the earlier axes is kept until it has been joined: the finished ones here (a failure of an
        # earlier axis' deferred chain on THIS rank is re-raised on this rank, at the next call at the latest), the
        # running ones stay in the list for the next call / wait() -- none is dropped
        pending = self.__dict__.get('_spmd_pending') or []
        self._spmd_pending = [f for f in pending if not f.done()]
        for f in pending:
            if f.done():
                f.result()
        axis = self.axes[axis_name]
        return stack_inference(volume.shape[axis], self._make(volume, axis), self.ks, self.group, self._host_group,
                               self._match_desc(volume.shape, axis_name), self._chain_group, defer=defer)

    def wait(self):
        """SPMD mode: joins this rank's deferred chain work (every rank calls it before the job's clock stops / before the
        process group is torn down); re-raises what it raised"""
        pending = self.__dict__.pop('_spmd_pending', None) or []
        first = None
        for f in pending:           # join ALL of them, then re-raise the earliest failure
            try:
                f.result()
            except BaseException as e:      # noqa: BLE001
                first = first or e
        if first is not None:
            raise first

    def _match_desc(self, shape, axis_name):
        shape = tuple(int(v) for v in shape)
        width = [v for i, v in enumerate(shape) if i != self.axes[axis_name]][1]
        return dict(labels=list(self.labels), thing_list=list(self.thing_list), label_divisor=self.label_divisor,
                    iou_thr=self.merge_iou_thr, ioa_thr=self.merge_ioa_thr, width=width, axis_name=axis_name, shape3d=shape)

    def infer_on_axis(self, volume, axis_name):
        from . import sparse
        shape = tuple(int(s) for s in volume.shape)
        # SPMD ranks: the call returns when this rank's GPU phase is through; the backward chain, the tracking and the gather
        # to rank 0 finish on the chain thread, behind the NEXT axis' GPU phase (patterns.py:102-134 ran them after the
        # forward pass, in the caller).  EMP_MG_DEFER=0: synchronous, as round 3.  Whether a dense stack is wanted follows
        # from the engine's settings alone, so every rank decides alike WITHOUT creating it: the stack (a zarr dataset
        # made with overwrite=True, or a volume-sized array) is created on rank 0 only, after inference, as the reference
        # does (multigpu.py:198-212) -- W ranks creating the same dataset raced its delete / create
        defer = self.spmd and not self.save_panoptic and os.environ.get('EMP_MG_DEFER', '1') != '0'
        segs = self._segs_spmd(volume, axis_name, defer) if self.spmd else self._segs_spawn(volume, axis_name)
        if defer:
            if dist.get_rank(self.group) != 0:
                self._spmd_pending.append(segs)    # joined by a later call / wait()
                return None, None
            segs_fut = segs
        else:
            if segs is None:
                return None, None
            segs_fut = _Done(segs)
        stack = self.create_panoptic_stack(axis_name, shape)
        trackers = self.create_trackers(shape, axis_name)
        priv = self.create_trackers(shape, axis_name)
        min_size, min_extent = self.min_size, self.min_extent

        def tail():
            # forward matching (patterns.py:279-350), backward matching and tracking (multigpu.py:240-252) ran per slab on
            # the ranks (SlabMatcher); what is left here is the concatenation of the per-slab tracks and the size filters
            import time
            segs = segs_fut.result()
            self.last_host_s = list(segs['host_s'])      # every rank's matcher time
            self.last_timing = list(segs['timing'])      # per rank: matcher time, its un-overlapped tail, GPU phase, slices
            t0 = time.perf_counter()
            # (the size filters of multigpu.py:254-256 -- remove_small_objects, remove_pancakes -- inside the merge)
            merged = merge_partial_trackers(segs['parts'], axis_name, min_size=min_size, min_extent=min_extent)
            for tr in priv:
                tr.instances = merged[tr.class_id]
                tr.finished = True
            for tr, pv in zip(trackers, priv):
                tr.__dict__['_instances'] = pv.instances
                tr.finished = True
            self.last_merge_s = time.perf_counter() - t0

        if stack is None:
            # host-only work: runs behind the caller (the next axis' GPU work, typically); reading ``tracker.instances``
            # joins it, as with Engine3d.infer_on_axis
            fut = self._host_worker().submit(tail)
            for tr in trackers:
                tr.__dict__['_pending'] = fut
            return None, trackers
        tail()
        if isinstance(stack, np.ndarray):
            sparse.fill_panoptic_volume(stack, trackers)
        else:       # chunked store: compose every class in memory, one pass over the store's chunks
            tmp = np.zeros(shape, dtype=self.dtype)
            sparse.fill_panoptic_volume(tmp, trackers)
            stack[...] = tmp
        return stack, trackers

    def _host_worker(self):
        w = self.__dict__.get('_worker')
        if w is None:
            from concurrent.futures import ThreadPoolExecutor
            w = self.__dict__['_worker'] = ThreadPoolExecutor(max_workers=1, thread_name_prefix='emp-mg-match')
        return w
