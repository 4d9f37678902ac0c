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

The driver is backend-agnostic: it moves tensors with ``torch.distributed`` and calls
user-supplied callables for the arithmetic, so the sharding / halo / carry logic is tested
on CPU with gloo (tests/test_multigpu_cpu.py) and runs unchanged on RCCL.
"""
import math

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


class MultiGPUEngine3d:
    """``empanada_napari.multigpu.MultiGPUEngine3d`` (multigpu.py:121-260) for an SPMD launch:
    every rank of an initialised ``torch.distributed`` group (``torchrun``, backend ``nccl`` = RCCL)
    constructs the engine and calls ``infer_on_axis``; rank 0 gets ``(stack, trackers)``, the others
    ``(None, None)``."""
    MIN_WORLD = 2

    def __init__(self, model_config, inference_scale=1, label_divisor=1000, median_kernel_size=5, stuff_area=64,
                 void_label=0, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.3, force_connected=True,
                 min_size=500, min_extent=4, fine_boundaries=False, semantic_only=False, store_url=None,
                 chunk_size=(256, 256, 256), save_panoptic=False, batch_size=None, group=None):
        from .inference import Engine3d
        if not dist.is_initialized():
            raise Exception('MultiGPUEngine3d needs an initialised torch.distributed process group')
        if dist.get_world_size(group) < self.MIN_WORLD:
            raise Exception('MultiGPU inference requires 2 or more GPUs!')   # multigpu.py:143-144
        self.group = group
        self.local = Engine3d(model_config, inference_scale, label_divisor, median_kernel_size, stuff_area, void_label,
                              nms_threshold, nms_kernel, confidence_thr, force_connected, min_size, min_extent,
                              fine_boundaries, semantic_only, True, False, store_url, chunk_size, save_panoptic,
                              batch_size=batch_size)
        self.dtype = self.local.dtype

    @torch.no_grad()
    def infer_on_axis(self, volume, axis_name):
        from . import sparse
        from .engines import factor_pad, logits_to_prob
        e3, eng = self.local, self.local.engine
        axis = e3.axes[axis_name]
        rs = int(2 + math.log(e3.inference_scale, 2))

        raw_path = isinstance(volume, np.ndarray) and volume.dtype in (np.uint8, np.uint16) and e3.inference_scale == 1
        if raw_path:   # as Engine3d.predict_slices: raw integers go up, normalisation + factor_pad run in the stem kernel
            from .preprocess import normalize_params
            sub, mul = normalize_params(e3.preprocessor.mean, e3.preprocessor.std, np.iinfo(volume.dtype).max)
            moved = np.moveaxis(volume, axis, 0)
            pf = eng.padding_factor
            size = tuple(moved.shape[1:])
            pad_to = (-(-size[0] // pf) * pf, -(-size[1] // pf) * pf)
            bs = e3.slice_batch(pad_to)

        def forward_fn(lo, hi):
            from .inference import take
            items = []
            if raw_path:
                for i0 in range(lo, hi, bs):
                    xb = torch.from_numpy(np.ascontiguousarray(moved[i0:min(hi, i0 + bs)]))[:, None]
                    mo = eng.model(xb.to(eng.model.device, non_blocking=True), rs, interpolate_ins=not eng.coarse_boundaries,
                                   sub=float(sub), mul=float(mul), pad_to=pad_to)
                    sem = logits_to_prob(mo['sem_logits'])
                    for j in range(xb.shape[0]):
                        items.append({'ctr_hmp': mo['ctr_hmp'][j:j + 1], 'offsets': mo['offsets'][j:j + 1],
                                      'sem': sem[j:j + 1], 'size': size})
                return items
            for i0 in range(lo, hi, e3.batch_size or 8):
                imgs = [e3.preprocessor(np.asarray(take(volume, i, axis)))['image']
                        for i in range(i0, min(hi, i0 + (e3.batch_size or 8)))]
                size_ = tuple(imgs[0].shape[-2:])
                x = eng.to_model_device(factor_pad(torch.stack(imgs), eng.padding_factor))
                mo = eng.model(x, rs, interpolate_ins=not eng.coarse_boundaries)
                sem = logits_to_prob(mo['sem_logits'])
                for j in range(x.shape[0]):
                    items.append({'ctr_hmp': mo['ctr_hmp'][j:j + 1], 'offsets': mo['offsets'][j:j + 1],
                                  'sem': sem[j:j + 1], 'size': size_})
            return items

        def median_fn(maps):
            eng.median_queue.clear()
            for m in maps:
                eng.median_queue.append({'sem': m})
            out = eng.get_median('sem')
            eng.median_queue.clear()
            return out

        def segment_fn(item):
            h, w = item['size']
            return eng._segment(item, e3.inference_scale)[0, :h, :w]

        def segment_batch_fn(items):
            """voting + merge of the slab in launch groups of up to 64 slices (one host sync per group)"""
            pans = []
            for i0 in range(0, len(items), 64):
                grp = items[i0:i0 + 64]
                cells, _, _, kmax = eng.instance_cells_int(torch.cat([it['ctr_hmp'] for it in grp]),
                                                           torch.cat([it['offsets'] for it in grp]), e3.inference_scale)
                pan = eng.panoptic_merge_int(torch.cat([it['sem'] for it in grp]), cells, kmax)
                for j, it in enumerate(grp):
                    h, w = it['size']
                    pans.append(pan[j, :h, :w])
            return pans

        def to_rle_fn(pans):
            out = []
            for i0 in range(0, len(pans), 64):
                out += sparse.pan_stack_to_rle_segs(torch.stack(pans[i0:i0 + 64]), e3.labels, e3.label_divisor,
                                                    e3.thing_list, force_connected=True)
            return out

        n = volume.shape[axis]
        segs = distributed_stack_inference(n, forward_fn, median_fn, segment_fn, to_rle_fn, eng.ks, self.group,
                                           segment_batch_fn=segment_batch_fn)
        if segs is None:
            return None, None
        trackers = e3.create_trackers(volume.shape, axis_name)
        # sequential matching + tracking of the gathered run lists on rank 0, in C++ (sparse.StackMatcher)
        for tr in trackers:
            sm = sparse.StackMatcher(tr.class_id, e3.label_divisor, e3.merge_iou_thr, e3.merge_ioa_thr,
                                     match=tr.class_id in e3.thing_list)
            for s in segs:
                sm.push_objects(s[tr.class_id])
            sm.forward()
            tr.instances = sm.backward_and_track(axis_name, volume.shape)
            tr.finished = True
        for tr in trackers:
            sparse.remove_small_objects(tr, min_size=e3.min_size)
            sparse.remove_pancakes(tr, min_span=e3.min_extent)
        stack = e3.create_panoptic_stack(axis_name, volume.shape)
        if stack is not None:
            sparse.fill_panoptic_volume(stack, trackers)
        return stack, trackers
