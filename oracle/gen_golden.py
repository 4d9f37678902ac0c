"""Generate tests/golden/*.npz by importing the REFERENCE (build container only).

    python oracle/gen_golden.py            # needs /root/reference, CPU only

The reference cannot travel to the GPU box, so its outputs on seeded inputs are
committed as small fixtures; the inputs are either stored next to them or are
re-derivable from a seed through ``empanada-napari_amd/synth.py`` /
``weights.py`` (both deterministic numpy generators).  This script is the
"generating script" the fixtures are committed with.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


# ----------------------------------------------------------------------------
# A. network forward
# ----------------------------------------------------------------------------
def ref_model(cfg, sd):
    from empanada.models.quantization.panoptic_deeplab import QuantizablePanopticDeepLabPR
    kw = {k: v for k, v in cfg.items() if k != 'arch'}
    m = QuantizablePanopticDeepLabPR(quantize=False, **kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return m.eval()


def norm_image(img_u8, mean=0.57571, std=0.12765):
    from empanada_napari_amd.preprocess import normalize
    return normalize(img_u8, mean, std)


def gen_model():
    cfg = dict(weights.MITONET_PDL_CFG)
    sd = weights.seeded_state_dict(cfg, seed=0)
    m = ref_model(cfg, sd)
    cases = {
        'a': (np.stack([synth.blob_image(64, 64, seed=1), synth.blob_image(64, 64, seed=2)]), 2, False),
        'b': (np.stack([synth.blob_image(64, 64, seed=3)]), 3, True),
        'c': (np.stack([synth.em_tiles(1, 128, seed=5)[0]]), 2, False),
        'd': (np.stack([synth.blob_image(96, 160, seed=7)]), 2, True),
    }
    out = {}
    with torch.no_grad():
        for k, (img, rs, interp) in cases.items():
            x = torch.from_numpy(norm_image(img))[:, None]
            o = m(x, rs, interp)
            out[f'{k}_image'] = img
            out[f'{k}_render_steps'] = np.int64(rs)
            out[f'{k}_interpolate_ins'] = np.int64(interp)
            for name in ('sem_logits', 'ctr_hmp', 'offsets'):
                out[f'{k}_{name}'] = o[name].numpy()
        # fused model == unfused model, and fold() of both key layouts agree
        x = torch.from_numpy(norm_image(cases['a'][0]))[:, None]
        ref = m(x, 2, False)
        m.fuse_model()
        fo = m(x, 2, False)
        for name in ('sem_logits', 'ctr_hmp', 'offsets'):
            assert torch.allclose(ref[name], fo[name], atol=2e-4, rtol=1e-4), name
        fsd = {k: v.numpy() for k, v in m.state_dict().items()}
        A = weights.fold_state_dict(sd, cfg)
        B = weights.fold_state_dict(fsd, cfg)
        for n in A:
            assert np.allclose(A[n][0], B[n][0], atol=1e-6, rtol=1e-5), n
            assert np.allclose(A[n][1], B[n][1], atol=1e-5, rtol=1e-4), n
        out['fused_keys'] = np.array(sorted(fsd.keys()))
    save('pdl_forward', **out)


def gen_bifpn():
    """MitoNet_v1_mini-class (PanopticBiFPNPR) goldens, incl. a 4-class model (BASELINE configs[4])."""
    from empanada.models.quantization.panoptic_bifpn import QuantizablePanopticBiFPNPR
    out = {}
    for tag, ncls in (('m1', 1), ('m4', 4)):
        cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
        sd = weights.seeded_state_dict(cfg, seed=3)
        m = QuantizablePanopticBiFPNPR(quantize=False, **{k: v for k, v in cfg.items() if k != 'arch'})
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        cases = {'a': (np.stack([synth.blob_image(128, 128, seed=1), synth.em_tiles(1, 128, seed=2)[0]]), 2, False),
                 'b': (np.stack([synth.blob_image(128, 256, seed=3)]), 2, True)}
        with torch.no_grad():
            for k, (img, rs, interp) in cases.items():
                o = m(torch.from_numpy(norm_image(img))[:, None], rs, interp)
                out[f'{tag}{k}_image'] = img
                out[f'{tag}{k}_render_steps'] = np.int64(rs)
                out[f'{tag}{k}_interpolate_ins'] = np.int64(interp)
                for name in ('sem_logits', 'ctr_hmp', 'offsets'):
                    out[f'{tag}{k}_{name}'] = o[name].numpy().astype(np.float32)
            # the exported model only fuses the encoder (quantization/panoptic_bifpn.py:163-164)
            x = torch.from_numpy(norm_image(cases['a'][0]))[:, None]
            ref = m(x, 2, False)
            m.fuse_model()
            fo = m(x, 2, False)
            assert all(torch.allclose(ref[n], fo[n], atol=5e-4, rtol=1e-4) for n in ref)
            A = weights.fold_state_dict(sd, cfg)
            B = weights.fold_state_dict({k: v.numpy() for k, v in m.state_dict().items()}, cfg)
            for n in A:
                assert np.allclose(A[n][0], B[n][0], atol=1e-6, rtol=1e-5) and np.allclose(A[n][1], B[n][1], atol=1e-5, rtol=1e-4), n
    save('bifpn_forward', **out)


def gen_regnet():
    """RegNet encoders (the two the reference can export, quantization/encoders/__init__.py): PanopticBiFPNPR on
    regnety_6p4gf (grouped 3x3 + the per-pixel squeeze-excite gate, encoder at stride 32) and PanopticDeepLabPR on
    regnetx_6p4gf (whose stage 4 stays at stride 2: ``stage4_stride`` never reaches a RegNet built by name, see
    weights.regnet_stage_strides)."""
    from empanada.models.quantization.panoptic_bifpn import QuantizablePanopticBiFPNPR
    from empanada.models.quantization.panoptic_deeplab import QuantizablePanopticDeepLabPR
    out = {}
    specs = (('y', QuantizablePanopticBiFPNPR, dict(weights.MITONET_MINI_CFG, encoder='regnety_6p4gf', num_classes=2), 11,
              {'a': (np.stack([synth.blob_image(128, 128, seed=1), synth.em_tiles(1, 128, seed=2)[0]]), 2, False),
               'b': (np.stack([synth.blob_image(128, 256, seed=3)]), 2, True)}),
             ('x', QuantizablePanopticDeepLabPR, dict(weights.MITONET_PDL_CFG, encoder='regnetx_6p4gf'), 12,
              {'a': (np.stack([synth.blob_image(64, 64, seed=1), synth.blob_image(64, 64, seed=2)]), 2, False),
               'b': (np.stack([synth.blob_image(96, 160, seed=7)]), 3, True)}))
    for tag, cls, cfg, seed, cases in specs:
        sd = weights.seeded_state_dict(cfg, seed=seed)
        m = cls(quantize=False, **{k: v for k, v in cfg.items() if k != 'arch'})
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
        m.eval()
        with torch.no_grad():
            for k, (img, rs, interp) in cases.items():
                o = m(torch.from_numpy(norm_image(img))[:, None], rs, interp)
                out[f'{tag}{k}_image'] = img
                out[f'{tag}{k}_render_steps'] = np.int64(rs)
                out[f'{tag}{k}_interpolate_ins'] = np.int64(interp)
                for name in ('sem_logits', 'ctr_hmp', 'offsets'):
                    out[f'{tag}{k}_{name}'] = o[name].numpy().astype(np.float32)
            x = torch.from_numpy(norm_image(cases['a'][0]))[:, None]
            pyr = m.encoder(x)
            for i, f in enumerate(pyr):          # the pyramid itself: per-level mean |x| and a strided sample
                out[f'{tag}_pyr{i}_shape'] = np.array(f.shape, np.int64)
                out[f'{tag}_pyr{i}_absmean'] = np.float64(f.abs().mean().item())
                out[f'{tag}_pyr{i}_sample'] = f[:, ::7, ::3, ::3].numpy().astype(np.float32)
            ref = m(x, 2, False)
            m.fuse_model()
            fo = m(x, 2, False)
            assert all(torch.allclose(ref[n], fo[n], atol=5e-4, rtol=1e-4) for n in ref)
            fsd = {k: v.numpy() for k, v in m.state_dict().items()}
            A = weights.fold_state_dict(sd, cfg)
            B = weights.fold_state_dict(fsd, cfg)
            for n in A:
                assert np.allclose(A[n][0], B[n][0], atol=1e-6, rtol=1e-5) and np.allclose(A[n][1], B[n][1], atol=1e-5, rtol=1e-4), n
            # the export's key layout with the shapes behind it: weights.infer_cfg reads the architecture from exactly this
            out[f'{tag}_fused_keys'] = np.array(sorted(fsd.keys()))
            out[f'{tag}_fused_shapes'] = np.array([','.join(str(d) for d in fsd[k].shape) for k in sorted(fsd.keys())])
    save('regnet_forward', **out)


# ----------------------------------------------------------------------------
# B. post-processing on synthetic head tensors
# ----------------------------------------------------------------------------
def gen_postprocess():
    from empanada.inference import postprocess as pp
    from empanada.inference.engines import PanopticDeepLabRenderEngine

    class Fake(torch.nn.Module):
        def __init__(self, outs):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.outs = outs

        def forward(self, x, render_steps: int = 2, interpolate_ins: bool = True):
            return {k: v.clone() for k, v in self.outs.items()}

    out = {}
    specs = [  # (H, W, n_inst, coarse, num_classes, nms_kernel, thr, plateau)
        (64, 64, 0, True, 1, 3, 0.1, False),
        (64, 96, 1, True, 1, 3, 0.1, False),
        (128, 128, 19, True, 1, 3, 0.1, False),
        (128, 128, 20, True, 1, 7, 0.1, True),
        (128, 160, 21, True, 1, 3, 0.1, True),
        (256, 256, 200, True, 1, 3, 0.05, False),
        (64, 64, 12, False, 1, 3, 0.1, True),
        (96, 64, 30, False, 1, 4, 0.1, False),   # even NMS kernel (postprocess.py:63-65)
        (128, 128, 25, True, 4, 3, 0.1, False),  # multi-class softmax / argmax / torch.mode
    ]
    for i, (H, W, n, coarse, ncls, k, thr, plat) in enumerate(specs):
        sem_logits, ctr, off = synth.head_outputs(H, W, n, seed=100 + i, coarse=coarse,
                                                  num_classes=ncls, plateau=plat)
        tctr, toff = torch.from_numpy(ctr), torch.from_numpy(off)
        centers = pp.find_instance_center(tctr.clone(), thr, k)
        out[f'{i}_spec'] = np.array([H, W, n, int(coarse), ncls, k], dtype=np.int64)
        out[f'{i}_thr'] = np.float64(thr)
        out[f'{i}_centers'] = centers.numpy()
        step = 4 if coarse else 1
        if centers.size(0) > 0:
            out[f'{i}_groups'] = pp.group_pixels(centers, toff, step=step).numpy()
        thing_list = [1] if ncls == 1 else [1, 2]
        for divisor, conf in ((1000, 0.5), (10000, 0.3)):
            eng = PanopticDeepLabRenderEngine(
                Fake({'sem_logits': torch.from_numpy(sem_logits), 'ctr_hmp': tctr, 'offsets': toff}),
                thing_list=thing_list, label_divisor=divisor, stuff_area=64, void_label=0,
                nms_threshold=thr, nms_kernel=k, confidence_thr=conf, padding_factor=16,
                coarse_boundaries=coarse)
            image = torch.zeros(1, 1, H, W)
            pan = eng(image, (H - 3, W - 5), upsampling=1)
            out[f'{i}_pan_{divisor}'] = pan.numpy()
            cells = eng.get_instance_cells(tctr.clone(), toff, 1)
            out[f'{i}_cells'] = cells.numpy().astype(np.int32)
    save('postprocess', **out)


# ----------------------------------------------------------------------------
# C. recursive median queue / 3d engine traces
# ----------------------------------------------------------------------------
def gen_median():
    from empanada.inference.engines import PanopticDeepLabRenderEngine3d

    class Seq(torch.nn.Module):
        def __init__(self, outs):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.outs = outs
            self.i = 0

        def forward(self, x, render_steps: int = 2, interpolate_ins: bool = True):
            o = {k: v.clone() for k, v in self.outs[self.i].items()}
            self.i += 1
            return o

    out = {}
    H = W = 64
    nslices = 9
    rng = np.random.default_rng(7)
    base = [synth.head_outputs(H, W, 6, seed=300 + (z // 3), coarse=True) for z in range(nslices)]
    slices = []
    for z, (sem, ctr, off) in enumerate(base):
        sem = sem + rng.standard_normal(sem.shape).astype(np.float32) * 1.0
        slices.append({'sem_logits': torch.from_numpy(sem.astype(np.float32)),
                       'ctr_hmp': torch.from_numpy(ctr), 'offsets': torch.from_numpy(off)})
    out['sem_logits'] = np.stack([s['sem_logits'].numpy() for s in slices])
    out['ctr_hmp'] = np.stack([s['ctr_hmp'].numpy() for s in slices])
    out['offsets'] = np.stack([s['offsets'].numpy() for s in slices])
    for ks in (1, 3, 5, 7):
        eng = PanopticDeepLabRenderEngine3d(
            Seq(slices), thing_list=[1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3,
            confidence_thr=0.5, median_kernel_size=ks, padding_factor=16, coarse_boundaries=True)
        segs = []
        for z in range(nslices):
            r = eng(torch.zeros(1, 1, H, W), (H, W), 1)
            if r is not None:
                segs.append(r.numpy())
        segs += [s.numpy() for s in eng.end(1)]
        assert len(segs) == nslices, (ks, len(segs))
        out[f'pan_ks{ks}'] = np.stack(segs).astype(np.int32)
    # scalar trace of the recursion
    from empanada.inference.engines import _MedianQueue
    q = _MedianQueue(3)
    vals = [5, 1, 9, 0, 7, 2]
    tr = []
    for v in vals:
        q.enqueue({'sem': torch.tensor([[float(v)]])})
        o = q.get_next(['sem'])
        if o is not None:
            tr.append(float(o['sem']))
    tr += [float(o['sem']) for o in q.end()]
    out['scalar_in'] = np.array(vals, dtype=np.float32)
    out['scalar_out'] = np.array(tr, dtype=np.float32)
    save('median3d', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['model', 'bifpn', 'postprocess', 'median']
    if 'bifpn' in which:
        gen_bifpn()
    torch.manual_seed(0)
    if 'model' in which:
        gen_model()
    if 'postprocess' in which:
        gen_postprocess()
    if 'median' in which:
        gen_median()


# ----------------------------------------------------------------------------
# D. sparse label algebra: matcher / tracker / voting / consensus
# ----------------------------------------------------------------------------
def _stub_numba_skimage():
    """array_utils / consensus / tracker / matcher import numba (and matcher imports skimage.measure
    without using it in rle_matcher): stub both so the plain-Python bodies run."""
    import types
    if 'numba' not in sys.modules:
        nb = types.ModuleType('numba')
        nb.jit = lambda *a, **k: (lambda f: f)
        nb.int64 = int
        nb.types = types.ModuleType('numba.types')
        nb.typed = types.ModuleType('numba.typed')
        nb.typed.List = list
        sys.modules['numba'] = nb
        sys.modules['numba.types'] = nb.types
        sys.modules['numba.typed'] = nb.typed
    if 'skimage' not in sys.modules:
        sk = types.ModuleType('skimage')
        sk.measure = types.ModuleType('skimage.measure')
        sys.modules['skimage'] = sk
        sys.modules['skimage.measure'] = sk.measure


def synth_label_volume(shape, n_obj, seed):
    """Ground-truth-like 3-D instance labels: random ellipsoids, later ids overwrite earlier ones."""
    rng = np.random.default_rng(seed)
    d, h, w = shape
    zz, yy, xx = np.mgrid[0:d, 0:h, 0:w].astype(np.float32)
    vol = np.zeros(shape, dtype=np.int64)
    for i in range(1, n_obj + 1):
        c = rng.uniform(0.15, 0.85, 3) * np.array(shape)
        r = rng.uniform(0.08, 0.22, 3) * np.array(shape)
        m = ((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1
        vol[m] = i
    return vol


def axis_pan_slices(vol, axis, divisor, seed):
    """Per-slice 'panoptic predictions' along an axis: class-1 instances with slice-local ids and a
    little per-axis boundary jitter, the way an independent 2-D model would produce them."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(vol.shape[axis]):
        sl = np.take(vol, i, axis=axis).copy()
        drop = rng.random(sl.shape) < 0.03
        sl[drop] = 0
        ids = np.unique(sl)
        ids = ids[ids > 0]
        perm = rng.permutation(len(ids)) + 1
        pan = np.zeros_like(sl)
        for k, v in zip(perm, ids):
            pan[sl == v] = divisor + k
        out.append(pan)
    return out


def flatten_instances(inst):
    keys = np.array([int(k) for k in inst.keys()], dtype=np.int64)
    boxes = np.array([list(inst[k]['box']) for k in inst.keys()], dtype=np.int64).reshape(len(keys), -1)
    starts = [np.asarray(inst[k]['starts'], dtype=np.int64) for k in inst.keys()]
    runs = [np.asarray(inst[k]['runs'], dtype=np.int64) for k in inst.keys()]
    off = np.cumsum([0] + [len(s) for s in starts]).astype(np.int64)
    cat = lambda l: np.concatenate(l) if l else np.zeros(0, np.int64)
    return {'keys': keys, 'boxes': boxes, 'off': off, 'starts': cat(starts), 'runs': cat(runs)}


def gen_sparse():
    _stub_numba_skimage()
    from empanada import array_utils as au
    from empanada.consensus import merge_objects_from_trackers, merge_semantic_from_trackers
    from empanada.inference.matcher import RLEMatcher
    from empanada.inference.tracker import InstanceTracker
    from oracle import sparse as osp
    out = {}
    rng = np.random.default_rng(11)

    # --- range primitives on random inputs ---
    for t in range(6):
        n = int(rng.integers(2, 40))
        lists = []
        for _ in range(int(rng.integers(2, 4))):
            s = np.sort(rng.choice(400, size=n, replace=False))
            r = rng.integers(1, 6, size=n)
            e = np.minimum(s + r, np.append(s[1:], 10 ** 6))   # non-overlapping within one source
            lists.append(np.stack([s, e], axis=1).astype(np.int64))
        out[f'rng{t}_n'] = np.int64(len(lists))
        for j, l in enumerate(lists):
            out[f'rng{t}_in{j}'] = l
        for thr in (1, 2, 3):
            out[f'rng{t}_vote{thr}'] = np.asarray(au.vote_by_ranges([l.copy() for l in lists], thr)).reshape(-1, 2)
        a, b = lists[0], lists[1]
        out[f'rng{t}_inter'] = np.int64(au.rle_intersection(a[:, 0], a[:, 1] - a[:, 0], b[:, 0], b[:, 1] - b[:, 0]))
        out[f'rng{t}_iou'] = np.float64(au.rle_iou(a[:, 0], a[:, 1] - a[:, 0], b[:, 0], b[:, 1] - b[:, 0]))
        ms, mr = au.merge_rles(a[:, 0].copy(), (a[:, 1] - a[:, 0]).copy(), b[:, 0].copy(), (b[:, 1] - b[:, 0]).copy())
        out[f'rng{t}_merge'] = np.stack([ms, mr], axis=1)
        out[f'rng{t}_invert'] = np.asarray(au.invert_ranges(au.join_ranges([a.copy(), b.copy()]), 500)).reshape(-1, 2)

    # --- matcher -> tracker -> consensus on a synthetic volume ---
    shape = (24, 28, 32)
    divisor = 1000
    vol = synth_label_volume(shape, 7, seed=5)
    trackers = []
    for axis, name in enumerate(('xy', 'xz', 'yz')):
        slices = axis_pan_slices(vol, axis, divisor, seed=100 + axis)
        rle_stack = []
        matcher = RLEMatcher(1, divisor, 0.25, 0.25)
        for pan in slices:                                      # forward_matching, patterns.py:68-100
            seg = osp.pan_seg_to_rle_seg(pan, [1], divisor, [1], force_connected=True)
            if matcher.target_rle is None:
                matcher.initialize_target(seg[1])
            else:
                seg[1] = matcher(seg[1])
            rle_stack.append(seg)
        matcher.target_rle = None                               # backward_matching, patterns.py:102-121
        matcher.assign_new = False
        tr = InstanceTracker(1, divisor, shape, name)
        for idx in range(len(slices) - 1, -1, -1):
            seg = rle_stack[idx]
            if matcher.target_rle is None:
                matcher.initialize_target(seg[1])
            else:
                seg[1] = matcher(seg[1])
            tr.update(seg[1], idx)
        tr.finish()
        trackers.append(tr)
        for k, v in flatten_instances(tr.instances).items():
            out[f'trk_{name}_{k}'] = v
    for thr, ciou, bypass in ((2, 0.75, False), (1, 0.75, True), (3, 0.5, False)):
        inst = merge_objects_from_trackers(trackers, thr, ciou, bypass)
        for k, v in flatten_instances(inst).items():
            out[f'cons_{thr}_{int(bypass)}_{k}'] = v
    # semantic consensus: one instance per tracker
    sem_tr = []
    for tr in trackers:
        t2 = InstanceTracker(2, divisor, shape, tr.axis)
        allr = au.join_ranges([np.stack([a['starts'], a['starts'] + a['runs']], axis=1) for a in tr.instances.values()])
        t2.instances = {2000: {'box': (0, 0, 0) + shape, 'starts': allr[:, 0], 'runs': allr[:, 1] - allr[:, 0]}}
        sem_tr.append(t2)
    for k, v in flatten_instances(merge_semantic_from_trackers(sem_tr, 2)).items():
        out[f'semcons_{k}'] = v
    out['volume_shape'] = np.array(shape, dtype=np.int64)
    save('sparse', **out)


if __name__ == '__main__' and 'sparse' in (sys.argv[1:] or ['sparse']):
    gen_sparse()


def _stub_cztile():
    """inference/tile.py imports cztile (un-vendored) at module level; only Tiler.__init__ uses it.  The golden
    vectors below never call __init__: tile rectangles are inputs."""
    import types
    if 'cztile' not in sys.modules:
        cz = types.ModuleType('cztile')
        a = types.ModuleType('cztile.fixed_total_area_strategy_2d')
        a.AlmostEqualBorderFixedTotalAreaStrategy2D = object
        b = types.ModuleType('cztile.tiling_strategy')
        b.Region2D = object
        sys.modules['cztile'] = cz
        sys.modules['cztile.fixed_total_area_strategy_2d'] = a
        sys.modules['cztile.tiling_strategy'] = b


def synth_tile_pan_segs(shape, yranges, xranges, divisor, seed):
    """A global instance map (class 1) + one semantic region (class 2), cropped per tile with per-tile changes:
    tile-local instance ids, one object removed from a single tile inside the overlap band, one-pixel erosions."""
    rng = np.random.default_rng(seed)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    inst = np.zeros(shape, np.int32)
    for i in range(1, 15):
        cy, cx = rng.uniform(4, h - 4), rng.uniform(4, w - 4)
        ry, rx = rng.uniform(3, 9), rng.uniform(3, 9)
        inst[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] = i
    sem = (((yy - h * 0.5) / (h * 0.3)) ** 2 + ((xx - w * 0.45) / (w * 0.42)) ** 2 <= 1) & (inst == 0)
    tiles = []
    for ti, ((y0, y1), (x0, x1)) in enumerate(zip(yranges, xranges)):
        crop = inst[y0:y1, x0:x1].copy()
        pan = np.zeros(crop.shape, np.int32)
        ids = [v for v in np.unique(crop) if v > 0]
        drop = ids[ti % len(ids)] if ids and ti % 2 == 1 else -1      # seen by the other tiles only
        for k, v in enumerate(ids):
            if v == drop:
                continue
            m = crop == v
            if (v + ti) % 3 == 0:                                       # tile-dependent boundary
                m[:, :-1] &= m[:, 1:]
            pan[m] = divisor + 1 + k
        pan[sem[y0:y1, x0:x1] & (pan == 0)] = 2 * divisor
        tiles.append(pan)
    return tiles


def gen_tiles():
    _stub_numba_skimage()
    _stub_cztile()
    from empanada.consensus import merge_objects_from_tiles, merge_semantic_from_tiles
    from empanada.inference.tile import Tiler, calculate_overlap_rle
    from oracle import sparse as osp
    out = {}
    divisor = 1000
    for case, (shape, tile, ov) in enumerate((((72, 100), 48, 8), ((64, 64), (40, 36), 6), ((50, 90), 32, 10))):
        yr, xr = osp.tile_ranges_2d(shape, tile, ov)
        tiler = object.__new__(Tiler)
        tiler.image_shape, tiler.yranges, tiler.xranges = shape, yr, xr
        ovs, ovr = calculate_overlap_rle(yr, xr, shape)
        pans = synth_tile_pan_segs(shape, yr, xr, divisor, seed=40 + case)
        out[f'c{case}_shape'] = np.array(shape, np.int64)
        out[f'c{case}_yranges'] = np.array(yr, np.int64)
        out[f'c{case}_xranges'] = np.array(xr, np.int64)
        out[f'c{case}_overlap'] = np.stack([np.asarray(ovs, np.int64), np.asarray(ovr, np.int64)], axis=1).reshape(-1, 2)
        segs = []
        for i, pan in enumerate(pans):
            out[f'c{case}_pan{i}'] = pan
            seg = osp.pan_seg_to_rle_seg(pan, [1, 2], divisor, [1], force_connected=True)   # skimage-backed: input only
            # the reference's _join_ranges reads an unbound variable for a single-range input: keep such
            # slivers out of the vectors
            seg[1] = {k: v for k, v in seg[1].items() if len(v['starts']) > 1}
            seg = tiler.translate_rle_seg(seg, i)
            for lab in (1, 2):
                for k, v in flatten_instances(seg[lab]).items():
                    out[f'c{case}_t{i}_l{lab}_{k}'] = v
            segs.append(seg)
        for tag, ovl in (('ov', (ovs, ovr)), ('noov', None)):
            for k, v in flatten_instances(merge_objects_from_tiles([sg[1] for sg in segs], ovl)).items():
                out[f'c{case}_obj_{tag}_{k}'] = v
        for k, v in flatten_instances(merge_semantic_from_tiles([sg[2] for sg in segs])).items():
            out[f'c{case}_sem_{k}'] = v
    out['n_cases'] = np.int64(3)
    save('tiles', **out)


if __name__ == '__main__' and 'regnet' in (sys.argv[1:] or ['regnet']):
    gen_regnet()
if __name__ == '__main__' and 'tiles' in (sys.argv[1:] or ['tiles']):
    gen_tiles()
