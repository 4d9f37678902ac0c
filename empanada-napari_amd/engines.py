"""MI355X mirror of ``empanada.inference.engines`` (Render engines + median queue).

Same class names, constructor arguments, attributes and error behaviour as the
reference (empanada/inference/engines.py:223-394) so callers such as
``empanada_napari.inference.Engine2d/Engine3d`` can swap the import.  All
arithmetic runs in libempanada_hip.so through the C ABI (``_abi``); torch is
used for device memory and streams only.  There is no CPU path: constructing
an engine without a HIP device raises.

``HipPanopticDeepLab`` plays the role of the TorchScript model
(``model(image, render_steps, interpolate_ins) -> dict``, engines.py:250).
"""
import ctypes as C
import math
from collections import deque

import numpy as np
import torch
import torch.nn.functional as F

from . import _abi, weights

__all__ = ['HipPanopticDeepLab', 'PanopticDeepLabRenderEngine', 'PanopticDeepLabRenderEngine3d',
           'factor_pad', 'logits_to_prob']

IMG_DTYPES = {torch.float32: 0, torch.uint8: 1, torch.int16: 2, torch.uint16: 2}


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise RuntimeError('empanada_napari_amd needs a HIP device (MI355X); there is no CPU fallback')
    if device is None:      # the process's current device (a multi-GPU rank selects its GPU once; cuda:0 otherwise)
        return torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    return torch.device('cuda', torch.cuda.current_device()) if device.index is None else device


def factor_pad(tensor, factor=16):
    """postprocess.py:25-36."""
    h, w = tensor.size()[2:]
    pad_bottom = factor - h % factor if h % factor != 0 else 0
    pad_right = factor - w % factor if w % factor != 0 else 0
    if pad_bottom == 0 and pad_right == 0:
        return tensor
    return F.pad(tensor, (0, pad_right, 0, pad_bottom))


class HipPanopticDeepLab:
    """Panoptic-DeepLab(+PointRend) forward on the HIP engine.

    ``state_dict``: reference-layout parameters (unfused or ``fuse_model()``
    layout; torch tensors or numpy arrays), or an already folded dict from
    ``weights.fold_state_dict``.
    """

    PRECISIONS = ('fp16', 'fp32', 'fp16x3')      # emp_pdl_set_precision's 0 / 1 / 2

    def __init__(self, state_dict, cfg=None, device=None, folded=False, precision=None):
        """``precision``: None -- the environment variable EMP_PRECISION if set, else **'fp16x3'**, the mode that meets the
        contract: the reference computes this path in fp32 (engines.py:248-255) and the float heads of this mode are
        within 1e-3 of it in the MAX norm on every tile and weight draw tested (csrc/conv16x3.hip, conv16x3p.hip: the fp32
        graph with every convolution on the FP16 matrix pipe, operands split into fp16 pairs, three MFMAs per product into
        an fp32 accumulator).  'fp16' -- the fp16 ENGINE, the explicit throughput opt-in (3x the rate; ~1e-3 of the fp32
        forward in rms and ~5e-3 in the max norm: bench.py's headline, since BASELINE's metric is quoted in fp16).
        'fp32' -- the fp32 REFERENCE MODE (csrc/ref32.hip: fp32 maps and weights, exact fp32 matrix pipe, no fusion;
        ~10x slower than the fp16 engine; 1e-4 of the oracle): the device-side comparator."""
        self.device = _require_gpu(device)
        bifpn = 'BiFPN' in (cfg or {}).get('arch', '')
        self.cfg = dict(weights.MITONET_MINI_CFG if bifpn else weights.MITONET_PDL_CFG, **(cfg or {}))
        self.lib = _abi.load()
        c = _abi.PdlConfig()
        cfgd = self.cfg
        c.num_classes = cfgd['num_classes']
        if bifpn:
            c.arch, c.fpn_dim, c.fpn_layers, c.stage4_stride = 1, cfgd['fpn_dim'], cfgd['fpn_layers'], 32
            c.decoder_channels = cfgd['fpn_dim']
        else:
            c.stage4_stride = cfgd['stage4_stride']
            c.decoder_channels = cfgd['decoder_channels']
            c.aspp_channels = cfgd['aspp_channels'] or 0
            stages = cfgd['low_level_stages']
            c.n_stages = len(stages)
            ins_proj = weights.ins_projection_widths(cfgd)
            for i, s in enumerate(stages):
                c.low_level_stages[i] = s
                c.low_level_proj_sem[i] = cfgd['low_level_channels_project'][i]
                c.low_level_proj_ins[i] = ins_proj[i]
            for i, r in enumerate(cfgd['atrous_rates']):
                c.atrous_rates[i] = r
        c.ins_decoder = int(bool(cfgd['ins_decoder']))
        c.num_fc = cfgd['num_fc']
        c.subdivision_num_points = cfgd['subdivision_num_points']
        if weights.is_regnet(cfgd):
            # RegNet encoders (encoders/regnet.py): the default 'fp16x3' mode like every network (grouped 3x3 included);
            # precision='fp16' asks for the fp16 engine (generic implicit-GEMM convs, the grouped 3x3 as one launch; no
            # layer fusion, and no parity gate at the north star's 1e-3: tests/test_gpu_regnet.py states what it measures)
            r = self.cfg['regnet'] = weights.regnet_cfg(cfgd)
            c.encoder, c.rn_stem, c.rn_se = 1, r['w_stem'], int(r['use_se'])
            for i, st in enumerate(weights.regnet_stage_strides(cfgd)):
                c.rn_widths[i], c.rn_depths[i], c.rn_groups[i], c.rn_strides[i] = r['widths'][i], r['depths'][i], r['groups'][i], st
        elif cfgd.get('encoder', 'resnet50') != 'resnet50':
            raise NotImplementedError(f"encoder {cfgd['encoder']!r}: resnet50 and the RegNets are built")
        self._h = C.c_void_p()
        torch.cuda.set_device(self.device)
        _abi.check(self.lib.emp_pdl_create(C.byref(c), C.byref(self._h)), 'emp_pdl_create')
        if precision is not None:
            if precision not in self.PRECISIONS:
                raise ValueError(f"precision must be one of {self.PRECISIONS}, got {precision!r}")
            _abi.check(self.lib.emp_pdl_set_precision(self._h, self.PRECISIONS.index(precision)), 'emp_pdl_set_precision')
        self.precision = self.PRECISIONS[self.lib.emp_pdl_precision(self._h)]
        P = state_dict if folded else weights.fold_state_dict(state_dict, self.cfg)
        n = self.lib.emp_pdl_num_params(self._h)
        names = [self.lib.emp_pdl_param_name(self._h, i).decode() for i in range(n)]
        missing = [k for k in names if k not in P]
        if missing:
            raise KeyError(f'missing parameters: {missing[:5]}...')
        for k in names:
            w, b = P[k]
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            shape = (C.c_int64 * w.ndim)(*w.shape)
            _abi.check(self.lib.emp_pdl_set_param(self._h, k.encode(), w.ctypes.data_as(C.c_void_p), shape, w.ndim,
                                                  b.ctypes.data_as(C.c_void_p)), f'set_param({k})')
        _abi.check(self.lib.emp_pdl_finalize(self._h), 'emp_pdl_finalize')
        self._dummy = torch.zeros(1, device=self.device)
        self.num_classes = cfgd['num_classes']
        self.forward_calls = 0      # forward calls through __call__ (bench.py reports it; profile tools cross-check it)

    # --- the reference's model contract (engines.py:34,41,250) ---
    def eval(self):
        return self

    def parameters(self):
        yield self._dummy

    def __del__(self):
        try:
            if self._h:
                self.lib.emp_pdl_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def reserve(self, N, H, W):
        _abi.check(self.lib.emp_pdl_reserve(self._h, N, H, W), 'emp_pdl_reserve')

    def arena_bytes(self):
        return int(self.lib.emp_pdl_arena_bytes(self._h))

    def profile(self, enable=True):
        """Bracket every launch of the dominant kernel class (256x256 conv tile) with HIP events."""
        _abi.check(self.lib.emp_pdl_profile(self._h, int(enable)), 'emp_pdl_profile')

    def profile_read(self):
        """-> (milliseconds, algorithmic FLOPs, launches) of those launches since the last read."""
        import ctypes as C
        ms, fl, n = C.c_double(0), C.c_double(0), C.c_int(0)
        _abi.check(self.lib.emp_pdl_profile_read(self._h, C.byref(ms), C.byref(fl), C.byref(n)), 'emp_pdl_profile_read')
        return ms.value, fl.value, n.value

    def last_flops(self):
        return float(self.lib.emp_pdl_flops(self._h, 0, 0, 0, 0))

    @torch.no_grad()
    def __call__(self, image, render_steps=2, interpolate_ins=True, sub=0.0, mul=1.0, out=None, pad_to=None):
        """image: (N,1,H,W) cuda tensor, float32 (normalised) or uint8/uint16 (raw; then
        ``sub``/``mul`` are the normalisation constants).  ``pad_to=(Hp,Wp)``: run at that padded size with the
        reference's ``factor_pad`` (zeros after normalisation) fused into the stem.  Returns the reference's dict."""
        assert image.ndim == 4 and image.size(1) == 1, 'expected (N,1,H,W)'
        self.forward_calls += 1
        if image.device != self.device:
            image = image.to(self.device, non_blocking=True)
        image = image.contiguous()
        dt = IMG_DTYPES.get(image.dtype)
        if dt is None:
            raise TypeError(f'unsupported image dtype {image.dtype}')
        N, _, vh, vw = image.shape
        H, W = (vh, vw) if pad_to is None else (int(pad_to[0]), int(pad_to[1]))
        up = 2 ** (render_steps - 2) if render_steps >= 2 else 1.0 / (2 ** (2 - render_steps))
        Hs, Ws = int(H * up), int(W * up)
        hq, wq = (H, W) if interpolate_ins else (H // 4, W // 4)
        if out is None:
            sem = torch.empty((N, self.num_classes, Hs, Ws), dtype=torch.float32, device=self.device)
            ctr = torch.empty((N, 1, hq, wq), dtype=torch.float32, device=self.device)
            off = torch.empty((N, 2, hq, wq), dtype=torch.float32, device=self.device)
        else:
            sem, ctr, off = out
        _abi.check(self.lib.emp_pdl_forward_padded(self._h, _abi.ptr(image), dt, float(sub), float(mul), N, vh, vw, H, W,
                                                   int(render_steps), int(bool(interpolate_ins)), _abi.ptr(sem),
                                                   _abi.ptr(ctr), _abi.ptr(off), _abi.stream_ptr(self.device)),
                   'emp_pdl_forward')
        return {'sem_logits': sem, 'ctr_hmp': ctr, 'offsets': off}

    def tap(self, name):
        """fp16 NHWC activation of the last forward, as a (N,H,W,C) torch view (parity tests)."""
        p = C.c_void_p()
        shp = (C.c_int64 * 5)()
        _abi.check(self.lib.emp_pdl_tap(self._h, name.encode(), C.byref(p), shp), f'tap({name})')
        N, H, W, Cc, ld = [int(v) for v in shp]
        t = torch.empty((N, H, W, ld), dtype=torch.float16, device=self.device)
        _abi.check(self.lib.emp_copy_d2d(_abi.ptr(t), p, t.numel() * 2, _abi.stream_ptr(self.device)), 'emp_copy_d2d')
        return t[..., :Cc]

    def tap_raw(self, name, shape):
        """fp32 buffer of the last forward (``emp_pdl_tap_raw``) as a torch tensor of ``shape`` (parity tests)."""
        p, nb = C.c_void_p(), C.c_int64()
        _abi.check(self.lib.emp_pdl_tap_raw(self._h, name.encode(), C.byref(p), C.byref(nb)), f'tap_raw({name})')
        t = torch.empty(shape, dtype=torch.float32, device=self.device)
        assert t.numel() * 4 <= nb.value, (name, shape, nb.value)
        _abi.check(self.lib.emp_copy_d2d(_abi.ptr(t), p, t.numel() * 4, _abi.stream_ptr(self.device)), 'emp_copy_d2d')
        return t

    def tap_names(self):
        n = self.lib.emp_pdl_num_taps(self._h)
        return [self.lib.emp_pdl_tap_name(self._h, i).decode() for i in range(n)]


@torch.no_grad()
def logits_to_prob(logits, out=None):
    """engines.py:22-30 on the device (``out``: optional contiguous destination of the same shape)."""
    lib = _abi.load()
    logits = logits.contiguous()
    if out is None:
        out = torch.empty_like(logits)
    assert out.shape == logits.shape and out.is_contiguous() and out.dtype == torch.float32
    N, Cc, H, W = logits.shape
    _abi.check(lib.emp_logits_to_prob(_abi.ptr(logits), _abi.ptr(out), N, Cc, H, W, _abi.stream_ptr(logits.device)),
               'emp_logits_to_prob')
    return out


class _Engine:
    def __init__(self, model):
        self.model = model.eval()

    def to_model_device(self, tensor):
        device = next(self.model.parameters()).device
        return tensor.to(device, non_blocking=True)


class _MedianQueue:
    """engines.py:47-90; the median runs as one HIP kernel over the queued maps
    and overwrites the middle item's ``sem`` in place (recursive filter, :76-84)."""

    def __init__(self, median_kernel_size, **kwargs):
        super().__init__(**kwargs)
        assert median_kernel_size % 2 == 1, 'Kernel size must be odd integer!'
        self.ks = median_kernel_size
        self.mid_idx = (median_kernel_size - 1) // 2
        self.median_queue = deque(maxlen=median_kernel_size)

    def reset(self):
        self.median_queue = deque(maxlen=self.ks)

    @torch.no_grad()
    def get_median(self, key):
        lib = _abi.load()
        maps = [o[key].contiguous() for o in self.median_queue]
        ks = len(maps)
        out = torch.empty_like(maps[0])
        ptrs = (C.c_void_p * ks)(*[m.data_ptr() for m in maps])
        _abi.check(lib.emp_median_slices(ptrs, ks, _abi.ptr(out), maps[0].numel(), _abi.stream_ptr(out.device)),
                   'emp_median_slices')
        return out

    def get_next(self, keys):
        nq = len(self.median_queue)
        if nq <= self.mid_idx:
            output = self.median_queue[-1]
        elif nq > self.mid_idx and nq < self.ks:
            return None
        elif nq == self.ks:
            output = self.median_queue[self.mid_idx]
            for key in keys:
                output[key] = self.get_median(key)
        return output

    def enqueue(self, item):
        self.median_queue.append(item)

    def end(self):
        return list(self.median_queue)[self.mid_idx + 1:]


class PanopticDeepLabRenderEngine(_Engine):
    """engines.py:223-325."""
    MAX_CENTERS = 4096

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, padding_factor=16, coarse_boundaries=True, **kwargs):
        super().__init__(model=model)
        self.thing_list = thing_list
        self.label_divisor = label_divisor
        self.stuff_area = stuff_area
        self.void_label = void_label
        self.nms_threshold = nms_threshold
        self.nms_kernel = nms_kernel
        self.confidence_thr = confidence_thr
        self.padding_factor = padding_factor
        self.coarse_boundaries = coarse_boundaries
        self.lib = _abi.load()

    @torch.no_grad()
    def infer(self, image, render_steps=2):
        model_out = self.model(image, render_steps, interpolate_ins=not self.coarse_boundaries)
        model_out['sem'] = logits_to_prob(model_out['sem_logits'])
        return model_out

    # ---- batched building blocks (N images per launch group) ----
    @torch.no_grad()
    def instance_cells_int(self, ctr_hmp, offsets, upsampling=1):
        """-> (cells (N,H,W) int32, centers (N,MAX,2) int32, num_centers (N,) int32)."""
        ctr_hmp = ctr_hmp.contiguous().float()
        offsets = offsets.contiguous().float()
        N, _, h, w = ctr_hmp.shape
        step = 4 if self.coarse_boundaries else 1
        up = int(upsampling * step)
        dev = ctr_hmp.device
        cells = torch.empty((N, h * up, w * up), dtype=torch.int32, device=dev)
        max_c = self.MAX_CENTERS
        while True:
            centers = torch.empty((N, max_c, 2), dtype=torch.int32, device=dev)
            num = torch.empty((N,), dtype=torch.int32, device=dev)
            work = torch.empty((int(self.lib.emp_instance_cells_work_bytes(N, h, w)),), dtype=torch.uint8, device=dev)
            _abi.check(self.lib.emp_instance_cells(_abi.ptr(ctr_hmp), _abi.ptr(offsets), N, h, w,
                                                   float(self.nms_threshold), int(self.nms_kernel), step, up,
                                                   _abi.ptr(cells), _abi.ptr(centers), _abi.ptr(num), max_c,
                                                   _abi.ptr(work), _abi.stream_ptr(dev)), 'emp_instance_cells')
            kmax = int(num.max().item())
            if kmax <= max_c:
                return cells, centers, num, kmax
            max_c = 1 << (kmax - 1).bit_length()  # rare: more centres than the default bound, redo exactly

    @torch.no_grad()
    def panoptic_merge_int(self, sem, cells, max_ids):
        """sem (N,C,H,W) probabilities, cells (N,H,W) int32 -> pan (N,H,W) int64."""
        sem = sem.contiguous().float()
        N, Cc, H, W = sem.shape
        assert cells.shape == (N, H, W), f'{cells.shape} vs {(N, H, W)}'
        dev = sem.device
        pan = torch.empty((N, H, W), dtype=torch.int64, device=dev)
        work = torch.empty((int(self.lib.emp_panoptic_merge_work_bytes(N, Cc, max_ids)),), dtype=torch.uint8, device=dev)
        tl = (C.c_int32 * max(1, len(self.thing_list)))(*self.thing_list)
        _abi.check(self.lib.emp_panoptic_merge(_abi.ptr(sem), _abi.ptr(cells.contiguous()), N, Cc, H, W,
                                               float(self.confidence_thr), tl, len(self.thing_list),
                                               int(self.label_divisor), int(self.stuff_area), int(self.void_label),
                                               int(max_ids), _abi.ptr(pan), _abi.ptr(work), _abi.stream_ptr(dev)),
                   'emp_panoptic_merge')
        return pan

    # ---- the reference's single-image API ----
    @torch.no_grad()
    def get_instance_cells(self, ctr_hmp, offsets, upsampling=1):
        cells, _, _, _ = self.instance_cells_int(ctr_hmp, offsets, upsampling)
        return cells.float()[:, None]  # (1,1,H,W) float, as engines.py:271-275

    @torch.no_grad()
    def _harden_seg(self, sem):
        """engines.py:114-121: (N,C,H,W) probabilities -> (N,1,H,W) int64 class map.  API surface only: the hot path
        hardens inside the merge kernel (``panoptic_merge_int``) and never materialises this map."""
        if sem.size(1) > 1:
            return torch.argmax(sem, dim=1, keepdim=True)
        return (sem >= self.confidence_thr).long()

    @torch.no_grad()
    def get_panoptic_seg(self, sem, instance_cells):
        """engines.py:277-292: ``sem`` is the HARDENED class map (1,H,W) int64, ``instance_cells`` (1,1,H,W).
        The class map re-enters the merge kernel as an exact one-hot 'probability' (argmax / >= 0.5 give it back)."""
        sem = sem.reshape(-1, sem.shape[-2], sem.shape[-1]).long()
        ncls = int(max([int(sem.max().item())] + list(self.thing_list))) + 1
        if ncls <= 2 and self.confidence_thr <= 1.0 and self.confidence_thr > 0.0:
            onehot = (sem == 1).float()[:, None]
        else:
            onehot = torch.stack([(sem == c).float() for c in range(ncls)], dim=1)
        cells = instance_cells.reshape(sem.shape).to(torch.int32)
        return self.panoptic_merge_int(onehot, cells, int(cells.max().item()))

    @torch.no_grad()
    def postprocess(self, sem, instance_cells):
        cells = instance_cells.reshape(sem.shape[0], sem.shape[-2], sem.shape[-1]).to(torch.int32)
        max_ids = int(cells.max().item())
        return self.panoptic_merge_int(sem, cells, max_ids)

    def __call__(self, image, size, upsampling=1):
        assert math.log(upsampling, 2).is_integer(), 'Upsampling factor not log base 2!'
        assert image.ndim == 4 and image.size(0) == 1
        h, w = size
        image = factor_pad(image, self.padding_factor)
        image = self.to_model_device(image)
        model_out = self.infer(image, int(2 + math.log(upsampling, 2)))
        cells, _, _, kmax = self.instance_cells_int(model_out['ctr_hmp'], model_out['offsets'], upsampling)
        pan_seg = self.panoptic_merge_int(model_out['sem'], cells, kmax)
        return pan_seg[..., :h, :w]

    @torch.no_grad()
    def call_raw(self, image, sub, mul):
        """``__call__`` for a raw integer image (1,1,h,w) uint8/uint16 at native scale: Preprocessor's
        (x - mean*max) * 1/(std*max) and ``factor_pad`` run inside the stem kernel (1-2 bytes per pixel uploaded
        instead of 4).  Same result as ``self(preprocessor(image), (h, w))``."""
        assert image.ndim == 4 and image.size(0) == 1
        h, w = image.shape[-2:]
        pf = self.padding_factor
        pad_to = (-(-h // pf) * pf, -(-w // pf) * pf)
        out = self.model(self.to_model_device(image), 2, interpolate_ins=not self.coarse_boundaries, sub=float(sub),
                         mul=float(mul), pad_to=pad_to)
        sem = logits_to_prob(out['sem_logits'])
        cells, _, _, kmax = self.instance_cells_int(out['ctr_hmp'], out['offsets'], 1)
        return self.panoptic_merge_int(sem, cells, kmax)[..., :h, :w]

    # ---- batched extension (config 2: "equal to N sequential reference calls") ----
    @torch.no_grad()
    def infer_batch(self, images, sizes=None, upsampling=1, sub=0.0, mul=1.0):
        """images (N,1,H,W) already padded to ``padding_factor``; returns pan (N,H,W) int64."""
        rs = int(2 + math.log(upsampling, 2))
        out = self.model(images, rs, interpolate_ins=not self.coarse_boundaries, sub=sub, mul=mul)
        sem = logits_to_prob(out['sem_logits'])
        cells, _, _, kmax = self.instance_cells_int(out['ctr_hmp'], out['offsets'], upsampling)
        return self.panoptic_merge_int(sem, cells, kmax)


class PanopticDeepLabRenderEngine3d(_MedianQueue, PanopticDeepLabRenderEngine):
    """engines.py:327-394."""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, padding_factor=16, coarse_boundaries=True,
                 **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr, median_kernel_size=median_kernel_size,
                         padding_factor=padding_factor, coarse_boundaries=coarse_boundaries)

    def _segment(self, model_out, upsampling):
        cells, _, _, kmax = self.instance_cells_int(model_out['ctr_hmp'], model_out['offsets'], upsampling)
        return self.panoptic_merge_int(model_out['sem'], cells, kmax)

    def end(self, upsampling=1):
        final_segs = []
        for model_out in list(self.median_queue)[self.mid_idx + 1:]:
            h, w = model_out['size']
            final_segs.append(self._segment(model_out, upsampling)[..., :h, :w])
        return final_segs

    def __call__(self, image, size, upsampling=1):
        assert math.log(upsampling, 2).is_integer(), 'Upsampling factor not log base 2!'
        assert image.ndim == 4 and image.size(0) == 1
        h, w = size
        image = factor_pad(image, self.padding_factor)
        image = self.to_model_device(image)
        model_out = self.infer(image, int(2 + math.log(upsampling, 2)))
        model_out['size'] = size
        self.enqueue(model_out)
        median_out = self.get_next(keys=['sem'])
        if median_out is None:
            return None
        return self._segment(median_out, upsampling)[..., :h, :w]
