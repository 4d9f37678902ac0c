"""MI355X mirror of ``empanada_napari.inference`` (upper drop-in boundary).

``Engine2d``, ``Engine3d``, ``tracker_consensus`` and ``stack_postprocessing`` keep
the reference's constructor arguments, attributes, ``update_params`` and return
types (empanada_napari/inference.py:56-578) so the napari widgets can swap the
import.  Differences, all behind the same results:
  * the network forward runs batched on the GPU (slices are independent; the
    recursive median queue is then fed strictly in slice order);
  * dense -> RLE (connected components + run extraction) runs on the GPU for a
    whole chunk of slices; no worker process, no pickled dense maps;
  * ``model_config['model']`` may be a TorchScript file (the reference's export),
    a state dict, or an already built ``HipPanopticDeepLab``.
zarr stores (``store_url``, zarr-backed input volumes) go through ``zstore.open_store``: the zarr package when it
is installed, otherwise this package's own reader / writer of the zarr v2 directory layout.
"""
import contextlib
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import sparse, weights, zstore
from .engines import (HipPanopticDeepLab, PanopticDeepLabRenderEngine, PanopticDeepLabRenderEngine3d,
                      factor_pad, logits_to_prob)
from .preprocess import Preprocessor, resize_by_factor

try:  # the widgets run these generators in a Qt worker thread
    from napari.qt.threading import thread_worker
except Exception:  # napari is not part of the hot path
    def thread_worker(fn):
        return fn

__all__ = ['Engine2d', 'Engine3d', 'tracker_consensus', 'stack_postprocessing', 'instance_relabel', 'take']

instance_relabel = sparse.instance_relabel


def take(array, indices, axis=0):
    """array_utils.py:10-27."""
    idx = tuple(slice(None) if n != axis else indices for n in range(array.ndim))
    return array[idx]


def resolve_model_file(fpath_or_url):
    """utils.load_model_to_device (empanada_napari/utils.py:80-106): a local file is used as it is; anything else is a URL
    whose download the reference caches as ``<torch.hub.get_dir()>/<basename of the URL path>``.  The engine has no
    network access: it accepts that cached file when a previous run of the plugin (or the user) has put it there."""
    if os.path.isfile(fpath_or_url):
        return fpath_or_url
    from urllib.parse import urlparse
    filename = os.path.basename(urlparse(fpath_or_url).path)
    cached = os.path.join(torch.hub.get_dir(), filename) if filename else None
    if cached and os.path.isfile(cached):
        return cached
    raise FileNotFoundError(f'model {fpath_or_url!r} is neither a local TorchScript file nor in the torch-hub cache '
                            f'({cached}); the engine does not download (no network access)')


def load_model_spec(model_config):
    """-> (state dict, configuration) of ``model_config['model']``: a TorchScript export (file, or URL in the torch-hub
    cache) or a state dict.  The architecture is read from the export (``weights.infer_cfg``: the reference's YAMLs do not
    state it); an ``'arch'`` dict in ``model_config`` overrides single keys."""
    m = model_config['model']
    if isinstance(m, dict):
        sd, mod = m, None
    else:
        mod = torch.jit.load(resolve_model_file(m), map_location='cpu')
        sd = mod.state_dict()
    cfg = weights.infer_cfg(sd, mod)
    cfg.update(model_config.get('arch', {}))
    return sd, cfg


def load_model(model_config, device):
    """utils.load_model_to_device (empanada_napari/utils.py:80-106) for the HIP engine."""
    m = model_config['model']
    if isinstance(m, HipPanopticDeepLab):
        return m
    sd, cfg = load_model_spec(model_config)
    # 'precision' ('fp16' / 'fp32' / 'fp16x3'): an optional key of the model config (no reference counterpart); None = the
    # environment's EMP_PRECISION, else the library's default 'fp16x3' -- the mode within 1e-3 (max norm) of the reference's fp32
    # forward; 'fp16' is the throughput opt-in
    return HipPanopticDeepLab(sd, cfg, device=device, precision=model_config.get('precision'))


def _is_chunked_store(v):
    """a zarr array, this package's zstore.DirArray or a dask array: known to support full ``[...]`` indexing"""
    if isinstance(v, zstore.DirArray):
        return True
    mod = type(v).__module__.split('.')[0]
    return mod in ('zarr', 'dask') and hasattr(v, 'dtype') and hasattr(v, 'shape')


def _require_scale_one(scale):
    """inference_scale is a power of two (volume_dataset.py:26-27)."""
    if not math.log(scale, 2).is_integer():
        raise Exception(f'Image rescaling must be log base 2, got {scale}')


def _open_zarr(store_url, mode=None):
    """``zarr.open(store_url[, mode])`` (inference.py:58,113,404,464).  A store created here is zarr v2 unless
    ``EMP_ZARR_FORMAT=3`` asks for the v3 layout (what zarr-python 3's ``create_array`` writes by default)."""
    fmt = os.environ.get('EMP_ZARR_FORMAT')
    return zstore.open_store(store_url, mode, zarr_format=int(fmt) if fmt else None)


def _device(device=None):
    """GPU of this engine: the caller's choice, else the process's current device (the reference picks
    ``cuda:{gpu}`` per rank, multigpu.py:35, and the default device otherwise)."""
    if device is None:
        return torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    return torch.device('cuda', torch.cuda.current_device()) if device.index is None else device


def _class_volume(zarr_store, name, shape, dtype, chunks):
    if zarr_store is not None:
        return zarr_store.create_array(name, shape=shape, dtype=dtype, overwrite=True, chunks=chunks)
    return np.empty(shape, dtype=dtype)      # every voxel is written by _fill(..., fresh=True)


def _fill(volume, instances):
    if isinstance(volume, np.ndarray):
        sparse.fill_volume(volume, instances, fresh=True)
    else:  # chunked store (zarr): stream slab by slab
        sparse.chunked_fill(volume, instances)


@thread_worker
def stack_postprocessing(trackers, store_url, model_config, label_divisor=1000, min_size=200, min_extent=4,
                         dtype=np.uint32, chunk_size=(256, 256, 256)):
    """empanada_napari/inference.py:56-109."""
    thing_list = model_config['thing_list']
    zarr_store = _open_zarr(store_url) if store_url is not None else None
    for class_id, class_name in model_config['class_names'].items():
        class_tracker = sparse.get_axis_trackers_by_class(trackers, class_id)[0]
        shape3d = class_tracker.shape3d
        stack_tracker = sparse.InstanceTracker(class_id, label_divisor, shape3d, 'xy')
        stack_tracker.instances = instance_relabel(class_tracker)
        if class_id in thing_list:
            sparse.remove_small_objects(stack_tracker, min_size=min_size)
            sparse.remove_pancakes(stack_tracker, min_span=min_extent)
            class_dtype = dtype
        else:
            class_dtype = np.uint8
        vol = _class_volume(zarr_store, f'{class_name}', shape3d, class_dtype if zarr_store is not None else dtype,
                            chunk_size)
        _fill(vol, stack_tracker.instances)
        yield vol, class_name, stack_tracker.instances


@thread_worker
def tracker_consensus(trackers, store_url, model_config, label_divisor=1000, pixel_vote_thr=2, cluster_iou_thr=0.75,
                      allow_one_view=False, min_size=200, min_extent=4, dtype=np.uint32, chunk_size=(256, 256, 256)):
    """empanada_napari/inference.py:111-169."""
    thing_list = model_config['thing_list']
    zarr_store = _open_zarr(store_url) if store_url is not None else None
    for class_id, class_name in model_config['class_names'].items():
        class_trackers = sparse.get_axis_trackers_by_class(trackers, class_id)
        shape3d = class_trackers[0].shape3d
        if class_id in thing_list:
            consensus = sparse.create_instance_consensus(class_trackers, pixel_vote_thr, cluster_iou_thr, allow_one_view)
            sparse.remove_small_objects(consensus, min_size=min_size)
            sparse.remove_pancakes(consensus, min_span=min_extent)
            class_dtype = dtype
        else:
            consensus = sparse.create_semantic_consensus(class_trackers, pixel_vote_thr)
            class_dtype = np.uint8
        vol = _class_volume(zarr_store, f'{class_name}', shape3d, class_dtype if zarr_store is not None else dtype,
                            chunk_size)
        _fill(vol, consensus.instances)
        yield vol, class_name, consensus.instances


class Engine2d:
    """empanada_napari/inference.py:171-325."""

    def __init__(self, model_config, inference_scale=1, label_divisor=1000, nms_threshold=0.1, nms_kernel=3,
                 confidence_thr=0.3, semantic_only=False, fine_boundaries=False, tile_size=0, use_gpu=True,
                 use_quantized=False, device=None):
        if not (torch.cuda.is_available() and use_gpu):
            raise RuntimeError('Engine2d: the MI355X engine has no CPU path (use_gpu=True and a HIP device required)')
        device = self.device = _device(device)
        model = load_model(model_config, device)
        self.thing_list = model_config['thing_list']
        self.labels = model_config['labels']
        self.class_names = model_config['class_names']
        self.label_divisor = label_divisor
        self.padding_factor = model_config['padding_factor']
        self.inference_scale = inference_scale
        self.fine_boundaries = fine_boundaries
        self.tile_size = tile_size
        self.engine = PanopticDeepLabRenderEngine(
            model, thing_list=[] if semantic_only else self.thing_list, label_divisor=label_divisor,
            nms_threshold=nms_threshold, nms_kernel=nms_kernel, confidence_thr=confidence_thr,
            padding_factor=self.padding_factor, coarse_boundaries=not fine_boundaries)
        self.preprocessor = Preprocessor(**model_config['norms'])

    def update_params(self, inference_scale, label_divisor, nms_threshold, nms_kernel, confidence_thr,
                      fine_boundaries, semantic_only=False, tile_size=0):
        self.inference_scale = inference_scale
        self.engine.input_scale = inference_scale
        self.label_divisor = label_divisor
        self.engine.label_divisor = label_divisor
        self.nms_threshold = nms_threshold
        self.engine.nms_threshold = nms_threshold
        self.nms_kernel = nms_kernel
        self.engine.nms_kernel = nms_kernel
        self.confidence_thr = confidence_thr
        self.engine.confidence_thr = confidence_thr
        self.fine_boundaries = fine_boundaries
        self.engine.coarse_boundaries = not fine_boundaries
        self.engine.thing_list = [] if semantic_only else self.thing_list
        self.tile_size = tile_size

    @torch.no_grad()
    def force_connected(self, pan_seg):
        """:263-279 on the GPU (one fused call: class-range select, 8-connected components, + min id; int64 -> int32 on
        the device); accepts a device label map or a numpy array and returns an int32 numpy array."""
        return self._force_connected_dev(pan_seg).cpu().numpy()

    def _force_connected_dev(self, pan_seg, out=None):
        pan = pan_seg if isinstance(pan_seg, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pan_seg))
        pan = pan.to(self.device, non_blocking=True).to(torch.int64)
        squeeze = pan.ndim == 2
        res = sparse.force_connected(pan[None] if squeeze else pan, list(self.engine.thing_list), self.label_divisor, out)
        return res[0] if squeeze else res

    def infer(self, image):
        if self.tile_size > 0 and any(s > self.tile_size for s in image.shape):
            return self._infer_tiled(image)
        _require_scale_one(self.inference_scale)
        size = image.shape
        if (isinstance(image, np.ndarray) and image.ndim == 2 and image.dtype in (np.uint8, np.uint16)
                and self.inference_scale == 1):
            # raw integers go up (1-2 B/pixel); normalisation + factor_pad are fused into the stem kernel
            from .preprocess import normalize_params
            sub, mul = normalize_params(self.preprocessor.mean, self.preprocessor.std, np.iinfo(image.dtype).max)
            x = torch.from_numpy(np.ascontiguousarray(image))[None, None]
            return self.force_connected(self.engine.call_raw(x, sub, mul).squeeze(0))
        x = self.preprocessor(resize_by_factor(image, self.inference_scale))['image'].unsqueeze(0)
        pan_seg = self.engine(x, size, upsampling=self.inference_scale)
        return self.force_connected(pan_seg.squeeze(0))

    @torch.no_grad()
    def infer_batch(self, images, batch=32):
        """``[self.infer(im) for im in images]`` for equally sized 2-D uint8 / uint16 images at native scale, as a
        pipeline (no reference counterpart: the reference's engine asserts batch 1, engines.py:306, and a caller loops):
        ``batch`` images per forward; the next batch's upload (pinned staging buffer, copy stream) and the previous
        batch's int32 label maps going back (second copy stream, straight into one pinned result block) overlap the
        current batch's forward + voting + merge + force_connected.  Returns a list of int32 (h,w) arrays (views of
        that block).  On the fp16 engine bit-identical to the per-image calls (batch invariance, DESIGN.md finding 13); in the
        default fp16x3 precision the kernels a layer runs on depend on the batch (hl32 plane region from 128 pixel tiles, split-K
        below half the chip), so the float heads of a tile agree between batch sizes to fp32 rounding (~1e-6), not to the bit."""
        from .preprocess import normalize_params
        images = list(images)
        if not images:
            return []
        h, w = images[0].shape
        dt = images[0].dtype
        assert all(im.ndim == 2 and im.shape == (h, w) and im.dtype == dt for im in images), 'one size and dtype per call'
        assert dt in (np.uint8, np.uint16) and self.inference_scale == 1 and not (self.tile_size > 0 and max(h, w) > self.tile_size)
        sub, mul = normalize_params(self.preprocessor.mean, self.preprocessor.std, np.iinfo(dt).max)
        eng, dev = self.engine, self.device
        pf = self.padding_factor
        pad_to = (-(-h // pf) * pf, -(-w // pf) * pf)
        n = len(images)
        tdt = torch.uint8 if dt == np.uint8 else torch.uint16
        # staging: two pinned input and two pinned output blocks + their device twins, kept on the engine (pinning
        # hundreds of MB per call costs more than a batch's forward); results are ordinary numpy arrays filled from
        # the pinned blocks by a copier thread (the copies release the GIL) while the GPU runs the next batch
        key = (batch, h, w, str(dt))
        if self.__dict__.get('_stage_key') != key:
            self._stage_key = key
            self._stage = dict(
                hin=[torch.empty((batch, 1, h, w), dtype=tdt, pin_memory=True) for _ in range(2)],
                hout=[torch.empty((batch, h, w), dtype=torch.int32, pin_memory=True) for _ in range(2)],
                din=[torch.empty((batch, 1, h, w), dtype=tdt, device=dev) for _ in range(2)],
                dout=[torch.empty((batch, h, w), dtype=torch.int32, device=dev) for _ in range(2)])
        st_ = self._stage
        stage, hout, d_in, d_out = st_['hin'], st_['hout'], st_['din'], st_['dout']
        result = np.empty((n, h, w), dtype=np.int32)
        copier = self._host_copier()
        main = torch.cuda.current_stream(dev)
        up, down = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_in = [None, None]       # upload of the buffer finished
        ev_free = [None, None]     # forward that read the input buffer finished
        ev_done = [None, None]     # label maps of the buffer complete
        ev_back = [None, None]     # download of the output buffer finished
        host_free = [None, None]   # copier has emptied the pinned output block
        chunks = [(i0, min(n, i0 + batch)) for i0 in range(0, n, batch)]

        def upload(k):
            """host staging copy + H2D of batch k, on the uploader thread (numpy copies and the enqueue release the GIL)"""
            i0, i1 = chunks[k]
            b = k & 1
            if ev_in[b] is not None:
                ev_in[b].synchronize()          # the staging buffer's previous upload has left the host
            st = stage[b].numpy()
            for j in range(i1 - i0):
                st[j, 0] = images[i0 + j]
            with torch.cuda.stream(up):
                if ev_free[b] is not None:
                    up.wait_event(ev_free[b])
                d_in[b][:i1 - i0].copy_(stage[b][:i1 - i0], non_blocking=True)
                e = torch.cuda.Event()
                e.record(up)
            ev_in[b] = e

        def drain(b, i0, i1, ev):
            ev.synchronize()
            result[i0:i1] = hout[b][:i1 - i0].numpy()

        uploader = self._host_copier('_uploader')
        pending_up = uploader.submit(upload, 0)
        for k, (i0, i1) in enumerate(chunks):
            b, m = k & 1, i1 - i0
            pending_up.result()
            main.wait_event(ev_in[b])
            mo = eng.model(d_in[b][:m], 2, interpolate_ins=not eng.coarse_boundaries, sub=float(sub), mul=float(mul),
                           pad_to=pad_to)
            ev_free[b] = torch.cuda.Event()
            ev_free[b].record(main)
            if k + 1 < len(chunks):      # the other input buffer's last reader (forward k-1) is already recorded
                pending_up = uploader.submit(upload, k + 1)
            sem = logits_to_prob(mo['sem_logits'])
            cells, _, _, kmax = eng.instance_cells_int(mo['ctr_hmp'], mo['offsets'], 1)
            pan = eng.panoptic_merge_int(sem, cells, kmax)[:, :h, :w]
            if ev_back[b] is not None:
                main.wait_event(ev_back[b])
            sparse.force_connected(pan.contiguous(), list(eng.thing_list), self.label_divisor, d_out[b][:m])
            ev_done[b] = torch.cuda.Event()
            ev_done[b].record(main)
            if host_free[b] is not None:
                host_free[b].result()           # the copier has emptied this pinned block (two batches ago)
            with torch.cuda.stream(down):
                down.wait_event(ev_done[b])
                hout[b][:m].copy_(d_out[b][:m], non_blocking=True)
                ev_back[b] = torch.cuda.Event()
                ev_back[b].record(down)
            host_free[b] = copier.submit(drain, b, i0, i1, ev_back[b])
        for f in host_free:
            if f is not None:
                f.result()
        return [result[i] for i in range(n)]

    def _host_copier(self, name='_copier'):
        c = self.__dict__.get(name)
        if c is None:
            c = self.__dict__[name] = ThreadPoolExecutor(max_workers=1, thread_name_prefix='emp' + name)
        return c


    def _infer_tiled(self, image, batched=True):
        """inference.py:283-318: per tile engine call -> RLE on the GPU -> image frame -> tile consensus -> dense.
        The tile rectangles come from ``tile.Tiler`` (cztile stand-in, parity unpinned).  The tiles have one size, so for
        raw integer images at native scale they go through the engine in batches (``batched``; same label maps as the
        per-tile calls on the fp16 engine -- batch invariance; fp16x3: heads to fp32 rounding, see infer_batch) and the dense -> RLE
        step runs on a whole batch."""
        from .tile import Tiler
        tiler = Tiler(image.shape, tile_size=self.tile_size, overlap_width=min(128, int(self.tile_size * 0.1)))
        self.last_tiler = tiler
        rle_segs = []
        raw = (batched and isinstance(image, np.ndarray) and image.dtype in (np.uint8, np.uint16) and
               self.inference_scale == 1)
        if raw:
            from .preprocess import normalize_params
            eng = self.engine
            sub, mul = normalize_params(self.preprocessor.mean, self.preprocessor.std, np.iinfo(image.dtype).max)
            th, tw = tiler(image, 0).shape
            pf = self.padding_factor
            pad_to = (-(-th // pf) * pf, -(-tw // pf) * pf)
            bs = int(max(1, min(64, (1 << 25) // max(1, pad_to[0] * pad_to[1]))))       # about 32 Mpixel per batch
            for i0 in range(0, len(tiler), bs):
                idx = range(i0, min(len(tiler), i0 + bs))
                x = torch.from_numpy(np.stack([tiler(image, i) for i in idx]))[:, None].to(self.device, non_blocking=True)
                mo = eng.model(x, 2, interpolate_ins=not eng.coarse_boundaries, sub=float(sub), mul=float(mul), pad_to=pad_to)
                sem = logits_to_prob(mo['sem_logits'])
                cells, _, _, kmax = eng.instance_cells_int(mo['ctr_hmp'], mo['offsets'], 1)
                pan = eng.panoptic_merge_int(sem, cells, kmax)[:, :th, :tw].to(torch.int32)
                segs = sparse.pan_stack_to_rle_segs(pan.contiguous(), self.labels, self.label_divisor, eng.thing_list)
                rle_segs += [tiler.translate_rle_seg(seg, i) for seg, i in zip(segs, idx)]
        else:
            for i in range(len(tiler)):
                tile = tiler(image, i)
                x = self.preprocessor(resize_by_factor(tile, self.inference_scale))['image'].unsqueeze(0)
                pan = self.engine(x, tile.shape, upsampling=self.inference_scale).squeeze(0).to(torch.int32)
                seg = sparse.pan_seg_to_rle_seg(pan, self.labels, self.label_divisor, self.engine.thing_list)
                rle_segs.append(tiler.translate_rle_seg(seg, i))
        rle_seg = {}
        for label in self.labels:
            if label in self.engine.thing_list:
                rle_seg[label] = sparse.merge_objects_from_tiles([rs[label] for rs in rle_segs], tiler.overlap_rle)
            else:
                rle_seg[label] = sparse.merge_semantic_from_tiles([rs[label] for rs in rle_segs])
        return sparse.rle_seg_to_pan_seg(rle_seg, image.shape)


class Engine3d:
    """empanada_napari/inference.py:327-578."""

    def __init__(self, model_config, inference_scale=1, label_divisor=1000, median_kernel_size=5, stuff_area=64,
                 void_label=0, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.3, force_connected=True,
                 min_size=500, min_extent=4, fine_boundaries=False, semantic_only=False, use_gpu=True,
                 use_quantized=False, store_url=None, chunk_size=(256, 256, 256), save_panoptic=False,
                 label_erosion=0, label_dilation=0, fill_holes_in_segmentation=False, batch_size=None, device=None):
        if not (torch.cuda.is_available() and use_gpu):
            raise RuntimeError('Engine3d: the MI355X engine has no CPU path (use_gpu=True and a HIP device required)')
        device = self.device = _device(device)
        model = load_model(model_config, device)
        self.model_config = model_config
        self.labels = model_config['labels']
        self.class_names = model_config['class_names']
        self.label_divisor = label_divisor
        self.padding_factor = model_config['padding_factor']
        self.inference_scale = inference_scale
        self.label_erosion = label_erosion
        self.label_dilation = label_dilation
        self.fill_holes_in_segmentation = fill_holes_in_segmentation
        self.thing_list = [] if semantic_only else model_config['thing_list']
        self.engine = PanopticDeepLabRenderEngine3d(
            model, thing_list=self.thing_list, median_kernel_size=median_kernel_size, label_divisor=label_divisor,
            stuff_area=stuff_area, void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
            confidence_thr=confidence_thr, padding_factor=self.padding_factor, coarse_boundaries=not fine_boundaries)
        self.preprocessor = Preprocessor(**model_config['norms'])
        self.axes = {'xy': 0, 'xz': 1, 'yz': 2}
        self.merge_iou_thr = 0.25
        self.merge_ioa_thr = 0.25
        self.force_connected = force_connected
        self.min_size = min_size
        self.min_extent = min_extent
        self.fine_boundaries = fine_boundaries
        self.save_panoptic = save_panoptic
        self.chunk_size = chunk_size
        self.zarr_store = _open_zarr(store_url, mode='w') if store_url is not None else None
        self.dtype = np.int32
        self.batch_size = batch_size

    def update_params(self, inference_scale, label_divisor, median_kernel_size, nms_threshold, nms_kernel,
                      confidence_thr, min_size, min_extent, fine_boundaries, semantic_only, store_url, chunk_size,
                      save_panoptic, label_erosion, label_dilation, fill_holes_in_segmentation):
        self.label_divisor = label_divisor
        self.inference_scale = inference_scale
        self.min_size = min_size
        self.min_extent = min_extent
        self.fine_boundaries = fine_boundaries
        e = self.engine
        e.label_divisor = label_divisor
        e.ks = median_kernel_size
        e.mid_idx = (median_kernel_size - 1) // 2
        e.nms_threshold = nms_threshold
        e.nms_kernel = nms_kernel
        e.confidence_thr = confidence_thr
        e.coarse_boundaries = not fine_boundaries
        self.label_erosion = label_erosion
        self.label_dilation = label_dilation
        self.fill_holes_in_segmentation = fill_holes_in_segmentation
        self.thing_list = [] if semantic_only else self.model_config['thing_list']
        e.thing_list = self.thing_list
        e.reset()
        self.save_panoptic = save_panoptic
        self.chunk_size = chunk_size
        self.zarr_store = _open_zarr(store_url, mode='w') if store_url is not None else None

    def _host_worker(self):
        """one worker thread per engine: forward-matching jobs and deferred backward passes run in submission order"""
        w = self.__dict__.get('_worker')
        if w is None:
            w = self.__dict__['_worker'] = ThreadPoolExecutor(max_workers=1, thread_name_prefix='emp-match')
        return w

    def create_trackers(self, shape3d, axis_name):
        return [sparse.InstanceTracker(label, self.label_divisor, shape3d, axis_name) for label in self.labels]

    def create_panoptic_stack(self, axis_name, shape3d):
        if self.zarr_store is not None and self.save_panoptic:
            return self.zarr_store.create_array(f'panoptic_{axis_name}', shape=shape3d, dtype=self.dtype,
                                                chunks=self.chunk_size, overwrite=True)
        if self.save_panoptic:
            return np.zeros(shape3d, dtype=self.dtype)
        return None

    @torch.no_grad()
    def slice_batch(self, padded_hw):
        """Slices per forward launch group: ``batch_size`` if given, else about 16 Mpixel per batch (64 slices of 512^2,
        16 of 1024^2, one of 4096^2) -- enough pixels to fill 256 CUs in the deep layers without growing the arena."""
        if self.batch_size:
            return int(self.batch_size)
        budget = int(os.environ.get('EMP_SLICE_BATCH_PIXELS', 1 << 24))
        return int(max(1, min(64, budget // max(1, int(padded_hw[0]) * int(padded_hw[1])))))

    def predict_slices(self, volume, axis):
        """Per-slice panoptic maps (device, int64 (h,w)) in slice order (see ``iter_slice_chunks``)."""
        return [p for chunk in self.iter_slice_chunks(volume, axis) for p in chunk]

    def iter_slice_chunks(self, volume, axis, post_stream=None):
        """Generator over consecutive groups of per-slice panoptic maps (device, int64 (h,w)), in slice order.

        ``post_stream``: run everything after the forward (median, voting, merge) on that CUDA stream and hand a group
        out only after the NEXT batch's forward has been enqueued on the caller's stream -- the caller then works on the
        group (on ``post_stream`` too) while the GPU already runs the next forward.  The group's tensors belong to
        ``post_stream``; the generator joins the two streams when it is exhausted.

        Same arithmetic as feeding PanopticDeepLabRenderEngine3d.__call__ slice by slice
        (engines.py:363-394): f[z] = raw[z] for the first / last ``mid`` slices, otherwise
        median(f[z-mid..z-1], raw[z..z+mid]).  Here the forward runs in batches and every run of
        consecutive slices goes through ONE recursive-median launch and ONE batched voting / merge
        launch group instead of a per-slice Python loop with a host sync each."""
        eng = self.engine
        lib = eng.lib
        from . import _abi
        if isinstance(volume, torch.Tensor):
            volume = volume.detach().cpu().numpy()
        elif not isinstance(volume, np.ndarray) and _is_chunked_store(volume) and \
                int(np.prod(volume.shape)) * np.dtype(volume.dtype).itemsize <= (8 << 30):
            # chunked stores (zarr arrays, zstore.DirArray, dask arrays): one sequential read of the chunks instead of a
            # strided gather per slice (the reference's VolumeDataset indexes the store per slice, volume_dataset.py:39).
            # Anything else -- e.g. synth.ProceduralVolume, which only offers .block() -- keeps per-slice access.
            volume = np.asarray(volume[...])
        n = volume.shape[axis]
        ks, mid = eng.ks, eng.mid_idx
        ups = self.inference_scale
        rs = int(2 + math.log(ups, 2))
        fhist = []            # last `mid` filtered maps, each (1,C,H,W)
        pend = []             # raw (sem, ctr, off) of slices not yet emitted, each with batch dim 1..B
        zp = 0                # global index of the first pending slice
        avail = 0
        size = None

        def emit(sem_f, ctr, off):
            cells, _, _, kmax = eng.instance_cells_int(ctr, off, ups)
            pan = eng.panoptic_merge_int(sem_f, cells, kmax)
            h, w = size
            return list(pan[:, :h, :w].unbind(0))

        def process(upto):
            """emit slices zp .. upto-1 (their look-ahead is available or they are tail slices)"""
            nonlocal zp, pend, fhist
            if upto <= zp:
                return []
            sem = torch.cat([p[0] for p in pend])
            ctr = torch.cat([p[1] for p in pend])
            off = torch.cat([p[2] for p in pend])
            cnt = upto - zp
            filt = []
            z = zp
            while z < upto:
                loc = z - zp
                if z < mid or z >= n - mid or mid == 0:     # head / tail slices stay unfiltered (engines.py:70-72,89-90)
                    f = sem[loc:loc + 1]
                    filt.append(f)
                    fhist = (fhist + [f])[-max(mid, 1):]
                    z += 1
                    continue
                run = min(upto, n - mid) - z                # consecutive filtered slices
                hist = torch.cat(fhist[-mid:]).contiguous()
                raw = sem[loc:loc + run + mid].contiguous()
                res = torch.empty((run,) + tuple(sem.shape[1:]), dtype=sem.dtype, device=sem.device)
                cntpx = sem[0].numel()
                _abi.check(lib.emp_median_recursive(_abi.ptr(hist), _abi.ptr(raw), raw.shape[0], ks, run, _abi.ptr(res),
                                                    cntpx, _abi.stream_ptr(sem.device)), 'emp_median_recursive')
                filt.append(res)
                fhist = (fhist + list(res[-mid:].split(1)))[-mid:]
                z += run
            chunk = emit(torch.cat(filt), ctr[:cnt], off[:cnt])
            rest = sem.shape[0] - cnt
            pend = [(sem[cnt:], ctr[cnt:], off[cnt:])] if rest > 0 else []
            zp = upto
            return chunk

        # Fast path for integer numpy volumes at native scale: the batch is cut out of the volume in one strided copy,
        # uploaded as raw integers (1-2 bytes per pixel instead of 4) and normalised + zero-padded inside the stem
        # kernel -- the same (x - mean*max) * 1/(std*max) arithmetic as Preprocessor + factor_pad.
        raw_path = (isinstance(volume, np.ndarray) and volume.dtype in (np.uint8, np.uint16) and ups == 1)
        if raw_path:
            from .preprocess import normalize_params
            sub, mul = normalize_params(self.preprocessor.mean, self.preprocessor.std, np.iinfo(volume.dtype).max)
            # one upload of the whole volume (1-2 bytes per voxel; chunks of up to 8 GiB stay on the host path), the
            # per-axis transposition happens on the device: a yz stack is a stride-1 gather on the host
            # (tried: an xy stack's batches uploaded one by one behind the previous forward instead of the whole volume
            # first -- the pageable copies block the host between the forwards: 512^3 xy axis 0.118 -> 0.145 s)
            # Round 4: up to EMP_VOLUME_ON_DEVICE_GIB (default 64: a 4096^3 uint8 volume next to the 30 GiB arena in 288 GB of
            # HBM) the whole volume goes up once; beyond that its batches go up through two PINNED staging buffers on a side
            # stream, one batch ahead, gathered by a helper thread (`_staged_batches`) -- round 3 fell back to pageable
            # per-batch copies above 8 GiB, which block the host between the forwards.
            # The threshold is also bounded by what the device has FREE (40 % of it: the network's arena for this slice
            # size may not be reserved yet, and ranks may time-share the GPU), and an upload that still runs out of memory
            # falls back to the staged path.
            limit = int(float(os.environ.get('EMP_VOLUME_ON_DEVICE_GIB', '64')) * (1 << 30))
            try:
                limit = min(limit, int(0.4 * torch.cuda.mem_get_info(eng.model.device)[0]))
            except Exception:       # noqa: BLE001 -- no such query on this runtime: the configured threshold alone
                pass
            on_dev = volume.nbytes <= limit
            if on_dev:
                try:
                    moved = torch.from_numpy(volume).to(eng.model.device)
                except getattr(torch, 'OutOfMemoryError', torch.cuda.OutOfMemoryError):      # (older torch builds have it under torch.cuda only)
                    torch.cuda.empty_cache()
                    on_dev = False
            if not on_dev:
                moved = volume
            moved = moved.movedim(axis, 0) if on_dev else np.moveaxis(volume, axis, 0)
            staged = None if on_dev else self._staged_batches(moved, n, self.slice_batch(
                (-(-moved.shape[1] // eng.padding_factor) * eng.padding_factor,
                 -(-moved.shape[2] // eng.padding_factor) * eng.padding_factor)), eng.model.device)
            pf = eng.padding_factor
            vh, vw = moved.shape[1:]
            pad_to = (-(-vh // pf) * pf, -(-vw // pf) * pf)
            size = (vh, vw)
        if raw_path:
            bs = self.slice_batch(pad_to)
        else:
            shp = [s for i, s in enumerate(volume.shape) if i != axis]
            bs = self.slice_batch((math.ceil(shp[0] / ups), math.ceil(shp[1] / ups)))
        held = None           # post_stream mode: the group computed last, handed out after the next forward is enqueued
        for i0 in range(0, n, bs):
            if raw_path:
                xb = (moved[i0:i0 + bs].contiguous() if on_dev else next(staged))[:, None]
                mo = eng.model(xb, rs, interpolate_ins=not eng.coarse_boundaries, sub=float(sub), mul=float(mul), pad_to=pad_to)
                nb = xb.shape[0]
            else:
                raws = [np.asarray(take(volume, i, axis)) for i in range(i0, min(n, i0 + bs))]
                size = tuple(raws[0].shape[-2:])          # label maps come back at the ORIGINAL slice size
                imgs = [self.preprocessor(resize_by_factor(r, ups))['image'] for r in raws]
                x = factor_pad(torch.stack(imgs), eng.padding_factor)
                mo = eng.model(eng.to_model_device(x), rs, interpolate_ins=not eng.coarse_boundaries)
                nb = x.shape[0]
            heads = (logits_to_prob(mo['sem_logits']), mo['ctr_hmp'].clone(), mo['offsets'].clone())
            if post_stream is not None:
                # the forward of this batch is enqueued: now let the caller work on the previous group (post stream)
                if held:
                    yield held
                    held = None
                ready = torch.cuda.Event()
                ready.record()
                for t in heads:
                    t.record_stream(post_stream)
            pend.append(heads)
            avail += nb
            with (torch.cuda.stream(post_stream) if post_stream is not None else contextlib.nullcontext()):
                if post_stream is not None:
                    post_stream.wait_event(ready)
                chunk = process(min(avail - mid, n - mid) if avail < n else n)
            if chunk:
                if post_stream is not None:
                    held = chunk
                else:
                    yield chunk
        if held:
            yield held
        with (torch.cuda.stream(post_stream) if post_stream is not None else contextlib.nullcontext()):
            chunk = process(n)
        if chunk:
            yield chunk
        if post_stream is not None:
            torch.cuda.current_stream().wait_stream(post_stream)
        eng.reset()

    @staticmethod
    def _staged_batches(moved, n, bs, dev):
        """Device tensors of slices [i0, i0 + bs) of a HOST volume (axis already moved to the front) that is too large to
        upload whole: a helper thread gathers batch k + 1 into one of two pinned buffers (a yz stack is a strided gather)
        and enqueues its host-to-device copy on a side stream while the caller's stream runs the forward of batch k; the
        consumer's stream waits for the copy's event only.  Yields exactly ceil(n / bs) tensors."""
        import queue
        import threading
        shape = (bs,) + tuple(moved.shape[1:])
        tdt = torch.from_numpy(np.empty(0, moved.dtype)).dtype
        pinned = [torch.empty(shape, dtype=tdt).pin_memory() for _ in range(2)]
        up = torch.cuda.Stream(device=dev)
        free = queue.Queue()
        ready = queue.Queue(maxsize=2)
        for i in range(2):
            free.put(i)
        STOP = -1           # pushed into `free` when the consumer stops early: the producer leaves instead of blocking

        def producer():
            try:
                for i0 in range(0, n, bs):
                    k = free.get()
                    if k == STOP:
                        return
                    nb = min(bs, n - i0)
                    np.copyto(pinned[k][:nb].numpy(), moved[i0:i0 + nb])      # contiguous or strided gather, host side
                    with torch.cuda.stream(up):
                        d = pinned[k][:nb].to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(up)
                    ready.put((k, d, ev, None))
            except BaseException as e:      # noqa: BLE001 -- handed to the consumer
                ready.put((None, None, None, e))

        th = threading.Thread(target=producer, name='emp-volume-stage', daemon=True)
        th.start()
        try:
            for _ in range(0, n, bs):
                k, d, ev, err = ready.get()
                if err is not None:
                    raise err
                torch.cuda.current_stream(dev).wait_event(ev)
                d.record_stream(torch.cuda.current_stream(dev))
                ev.synchronize()          # the pinned buffer may be refilled once its copy has left the host
                free.put(k)
                yield d
        finally:
            # also reached when the consumer stops early (an exception in a forward, the generator closed: GeneratorExit):
            # the producer must not stay blocked in free.get() holding two pinned buffers and a stream
            free.put(STOP)
            while th.is_alive():
                try:
                    ready.get(timeout=0.05)      # a producer blocked in ready.put() gets its slot
                except queue.Empty:
                    pass
            th.join()

    def infer_on_axis(self, volume, axis_name):
        """:491-578 -> (stack, trackers)."""
        _require_scale_one(self.inference_scale)
        axis = self.axes[axis_name]
        trackers = self.create_trackers(volume.shape, axis_name)
        stack = self.create_panoptic_stack(axis_name, volume.shape)
        # forward matching (patterns.py:68-100), backward matching (:102-121) and tracking (tracker.py:61-123) of every
        # class in C++ (sparse.StackMatcher): dense -> runs on the GPU in chunks, no Python object per slice object.
        # The forward matching of a chunk runs on a worker thread (its C++ / scipy calls release the GIL) while this
        # thread drives the GPU through the next chunk, as the reference overlaps them with a process (patterns.py:68).
        sms = {label: sparse.StackMatcher(label, self.label_divisor, self.merge_iou_thr, self.merge_ioa_thr,
                                          match=label in self.thing_list) for label in self.labels}
        n_seen = 0

        def match_chunk(per_label, width):
            for label, (runs_list, off) in per_label.items():
                sm = sms[label]
                first = len(sm)
                sm.push_runs_many(runs_list, width, off)      # slices and pair tables on the library's worker threads
                sm.prepare(max(0, first - 1), len(sm) - 1)
                sm.step_to(len(sm))

        worker = self._host_worker()
        jobs = []
        # two streams: the forwards stay back to back on the caller's stream; median / voting / merge of a batch, its
        # dense -> runs extraction and the host syncs that go with it run on a second one, one batch behind
        dev = self.engine.model.device
        if getattr(self, '_post_stream', None) is None or self._post_stream.device != dev:
            self._post_stream = torch.cuda.Stream(device=dev)
        post = self._post_stream if os.environ.get('EMP_STACK_TWO_STREAMS', '1') != '0' else None
        for pans in self.iter_slice_chunks(volume, axis, post_stream=post):
            n_seen += len(pans)
            with (torch.cuda.stream(post) if post is not None else contextlib.nullcontext()):
                for i0 in range(0, len(pans), 64):
                    chunk = torch.stack(pans[i0:i0 + 64])
                    per_label = sparse.pan_stack_to_runs(chunk, self.labels, self.label_divisor, self.thing_list,
                                                         force_connected=True)
                    jobs.append(worker.submit(match_chunk, per_label, chunk.shape[-1]))
        assert n_seen == volume.shape[axis]
        needs_gpu = bool(self.label_erosion > 0 or self.label_dilation > 0 or self.fill_holes_in_segmentation
                         or stack is not None)
        # the deferred pass must not see a later update_params(): freeze what it reads
        min_size, min_extent = self.min_size, self.min_extent
        erosion, dilation, fill_holes = self.label_erosion, self.label_dilation, self.fill_holes_in_segmentation
        margs = (tuple(volume.shape), list(self.labels), self.label_divisor, list(self.thing_list))
        priv = self.create_trackers(volume.shape, axis_name)      # filled by the deferred pass, published at its end

        def tail():
            """backward matching + tracking + filters of the axis (host only unless morphology / a dense stack is asked for);
            works on private trackers and publishes the result at the end"""
            for j in jobs:
                j.result()
            for tr in priv:
                sm = sms[tr.class_id]
                sm.forward()
                tr.instances = sm.backward_and_track(axis_name, volume.shape)
                tr.finished = True
            for tr in priv:
                sparse.remove_small_objects(tr, min_size=min_size)
                sparse.remove_pancakes(tr, min_span=min_extent)
            # optional morphology, in the reference's order (inference.py:560-570)
            if erosion > 0:
                for tr in priv:
                    sparse.erode(tr, *margs, iterations=erosion)
            if dilation > 0:
                for tr in priv:
                    sparse.dilate(tr, *margs, iterations=dilation)
            if fill_holes:
                for tr in priv:
                    sparse.fill_holes_in_segmentation(tr, *margs)
            if stack is not None:
                if isinstance(stack, np.ndarray):
                    sparse.fill_panoptic_volume(stack, priv)
                else:
                    tmp = np.zeros(stack.shape, dtype=stack.dtype)
                    sparse.fill_panoptic_volume(tmp, priv)
                    stack[...] = tmp
            for tr, pv in zip(trackers, priv):
                tr.__dict__['_instances'] = pv.instances
                tr.finished = True

        if needs_gpu:
            for j in jobs:
                j.result()
            tail()                       # GPU work stays on this thread
        else:
            # the backward pass needs no GPU: it runs behind the queued forward-matching jobs on the worker while the
            # caller moves on (the next axis' forward, typically); reading ``tracker.instances`` joins it
            fut = worker.submit(tail)
            for tr in trackers:
                tr.__dict__['_pending'] = fut
        self.engine.reset()
        return stack, trackers
