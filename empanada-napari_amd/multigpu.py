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
    the GPU; only run-length lists travel to rank 0 (``gather_object``), which does the
    inherently sequential slice-to-slice matching.
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
import math
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
        self.work.wait()
        self.buf = None


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


def slab_stack_inference(n_slices, backend, ks, group=None, host_group=None):
    """SPMD body of one axis on one rank.

    backend.forward(lo, hi, n_ahead)  -> (sem, stash): ``sem`` a tensor (hi-lo + n_ahead, ...) whose first hi-lo rows
                                         are the slab's probability maps (the rest is filled by the exchange)
    backend.median_inplace(sem, n_own, hist, n_ahead, first, last, ks)
                                      -> filters rows [0, n_own) of ``sem`` in place; ``hist``: the ``mid`` filtered
                                         maps before the slab (None on the first slab)
    backend.runs(sem_own, stash)      -> one entry per slice: {class: (runs (n,3) int64, id offset) | instance dict}
    Returns on rank 0 the per-slice entries of ALL slices in order, elsewhere None.  Tensors travel over ``group``
    (RCCL on GPUs: neighbour send / recv), the run lists over ``host_group`` (gloo: they are host data that the
    sequential matcher on rank 0's host consumes; no pickling through device memory)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mid = (ks - 1) // 2
    aw = active_ranks(n_slices, world, ks)
    bounds = slab_bounds(n_slices, aw) + [(n_slices, n_slices)] * (world - aw)
    lo, hi = bounds[rank]
    n_own = hi - lo
    per_slice = []
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
        backend.median_inplace(sem, n_own, hist, n_ahead, rank == 0, rank == aw - 1, ks)
        if mid and has_next:
            reqs.append(_isend(sem[n_own - mid:n_own], rank + 1, group))
        per_slice = backend.runs(sem[:n_own], stash)
        for r in reqs:
            r.wait()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(per_slice, gathered, dst=0, group=host_group if host_group is not None else group)
    if rank != 0:
        return None
    return [s for part in gathered for s in part]


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
    def runs(self, sem, stash):
        from . import sparse
        e3, eng = self.e3, self.eng
        ctr, off = stash
        h, w = self.size
        out = []
        for i0 in range(0, sem.shape[0], 64):
            sl = slice(i0, i0 + 64)
            cells, _, _, kmax = eng.instance_cells_int(ctr[sl], off[sl], 1)
            pan = eng.panoptic_merge_int(sem[sl], cells, kmax)[:, :h, :w]
            per_label = sparse.pan_stack_to_runs(pan, e3.labels, e3.label_divisor, e3.thing_list, force_connected=True)
            for j in range(pan.shape[0]):
                out.append({label: (rl[j], o) for label, (rl, o) in per_label.items()})
        return out


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
        make = (backend_factory or _default_backend_factory)(model_config, engine_kwargs, rank)
        res_q.put(('ready', rank, None))
        while True:
            cmd = cmd_q.get()
            if cmd[0] == 'stop':
                break
            _, volume, axis_name, ks = cmd
            if isinstance(volume, torch.Tensor):       # a numpy volume travels as a shared-memory tensor
                volume = volume.numpy()
            axis = {'xy': 0, 'xz': 1, 'yz': 2}[axis_name]
            segs = slab_stack_inference(volume.shape[axis], make(volume, axis), ks, None, host_group)
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
        self.zarr_store = _open_zarr(store_url, mode='w') if store_url is not None else None
        self._procs = None
        self._make = None
        self._host_group = None

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
        out = {}
        while len(out) < self.world:
            try:
                kind, rank, payload = self._res.get(timeout=5.0)
            except Exception:       # queue.Empty: make sure nobody died without a message
                dead = [i for i, p in enumerate(self._procs) if not p.is_alive() and i not in out]
                if dead:
                    self.close(kill=True)
                    raise RuntimeError(f'multi-GPU rank process(es) {dead} exited unexpectedly')
                continue
            if kind == 'error':
                self.close(kill=True)
                raise RuntimeError(f'multi-GPU rank {rank} failed:\n{payload}')
            assert kind == what, (kind, what)
            out[rank] = payload
        return out

    def close(self, kill=False):
        procs, self._procs = self._procs, None
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
            payload = torch.from_numpy(np.ascontiguousarray(volume)).share_memory_()   # one copy, mapped by every rank
        for q in self._cmd:
            q.put(('axis', payload, axis_name, self.ks))
        return self._collect('done')[0]

    def _segs_spmd(self, volume, axis_name):
        if self._make is None:
            rank = dist.get_rank(self.group)
            factory = self.backend_factory or _default_backend_factory
            local = int(os.environ.get('LOCAL_RANK', rank))
            self._make = factory(self.model_config, self.engine_kwargs, local)
            if dist.get_backend(self.group) == 'nccl':
                self._host_group = dist.new_group(backend='gloo')
        axis = self.axes[axis_name]
        return slab_stack_inference(volume.shape[axis], self._make(volume, axis), self.ks, self.group, self._host_group)

    def infer_on_axis(self, volume, axis_name):
        from . import sparse
        segs = self._segs_spmd(volume, axis_name) if self.spmd else self._segs_spawn(volume, axis_name)
        if segs is None:
            return None, None
        shape = tuple(int(s) for s in volume.shape)
        assert len(segs) == shape[self.axes[axis_name]]
        trackers = self.create_trackers(shape, axis_name)
        priv = self.create_trackers(shape, axis_name)
        width = [s for i, s in enumerate(shape) if i != self.axes[axis_name]][1]
        min_size, min_extent = self.min_size, self.min_extent
        stack = self.create_panoptic_stack(axis_name, shape)

        def tail():
            # forward matching (patterns.py:279-350), backward matching and tracking (multigpu.py:240-252) of the
            # gathered run lists, per class, in C++ (sparse.StackMatcher)
            for tr in priv:
                sm = sparse.StackMatcher(tr.class_id, self.label_divisor, self.merge_iou_thr, self.merge_ioa_thr,
                                         match=tr.class_id in self.thing_list)
                for s in segs:
                    v = s[tr.class_id]
                    if isinstance(v, tuple):
                        sm.push_runs(v[0], width, v[1])
                    else:
                        sm.push_objects(v)
                sm.forward()
                tr.instances = sm.backward_and_track(axis_name, shape)
                tr.finished = True
            for tr in priv:
                sparse.remove_small_objects(tr, min_size=min_size)
                sparse.remove_pancakes(tr, min_span=min_extent)
            for tr, pv in zip(trackers, priv):
                tr.__dict__['_instances'] = pv.instances
                tr.finished = True

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
