"""Upper boundary (Engine2d / Engine3d / tracker_consensus) on the MI355X vs the oracle's sparse
pipeline fed with the SAME per-slice panoptic maps: trackers, consensus instances and the filled
volumes must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIV = 1000


@pytest.fixture(scope='module')
def model_config():
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    # tiny test volumes: lift the centre head's bias so that every slice has centres (the seeded head gives
    # ~90 centres per 1024^2 tile, i.e. none on a 48x40 slice) and make the semantic head mostly foreground
    w, b = P['ins_center.head.1']
    P['ins_center.head.1'] = (w, b + np.float32(0.75))
    w, b = P['semantic_head.head.1']
    P['semantic_head.head.1'] = (w, b + np.float32(2.5))
    w, b = P['semantic_pr.point_head.predictor']
    P['semantic_pr.point_head.predictor'] = (w, b + np.float32(2.5))
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    return {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
            'norms': {'mean': 0.57571, 'std': 0.12765}}


def _oracle_axis(pan_segs, shape, axis_name, min_size, min_extent):
    from oracle import sparse as osp
    m = osp.RLEMatcher(1, DIV, 0.25, 0.25)
    stack = []
    for pan in pan_segs:
        seg = osp.pan_seg_to_rle_seg(pan, [1], DIV, [1], force_connected=True)
        stack.append(osp.apply_matchers(seg, [m]))
    m.target_rle = None
    m.assign_new = False
    tr = osp.InstanceTracker(1, DIV, shape, axis_name)
    for idx in range(len(pan_segs) - 1, -1, -1):
        seg = osp.apply_matchers(stack[idx], [m])
        tr.update(seg[1], idx)
    tr.finish()
    osp.remove_small_objects(tr, min_size)
    osp.remove_pancakes(tr, min_extent)
    return tr


def _same_instances(a, b):
    assert [int(k) for k in a] == [int(k) for k in b]
    for ka, kb in zip(a, b):
        assert tuple(int(v) for v in a[ka]['box']) == tuple(int(v) for v in b[kb]['box'])
        np.testing.assert_array_equal(np.asarray(a[ka]['starts']), np.asarray(b[kb]['starts']))
        np.testing.assert_array_equal(np.asarray(a[ka]['runs']), np.asarray(b[kb]['runs']))


def test_engine3d_orthoplane_consensus(model_config):
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    from oracle import sparse as osp
    vol = synth.blob_volume(20, 48, 40, seed=3, n_blobs=5)
    eng = Engine3d(model_config, label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.5,
                   min_size=30, min_extent=2, save_panoptic=True, batch_size=7)
    trackers, otrackers = {}, {}
    for axis_name, axis in (('xy', 0), ('xz', 1), ('yz', 2)):
        pans = [p.cpu().numpy() for p in eng.predict_slices(vol, axis)]
        assert len(pans) == vol.shape[axis] and pans[0].shape == tuple(s for i, s in enumerate(vol.shape) if i != axis)
        stack, trs = eng.infer_on_axis(vol, axis_name)
        otr = _oracle_axis(pans, vol.shape, axis_name, 30, 2)
        _same_instances(trs[0].instances, otr.instances)
        want = osp.numpy_fill_instances(np.zeros(vol.shape, np.int32), otr.instances)
        np.testing.assert_array_equal(stack, want)
        assert stack.dtype == np.int32
        trackers[axis_name], otrackers[axis_name] = trs, otr
    assert sum(len(t[0].instances) for t in trackers.values()) > 0
    out = list(tracker_consensus(trackers, None, model_config, label_divisor=DIV, pixel_vote_thr=2,
                                 cluster_iou_thr=0.75, allow_one_view=False, min_size=30, min_extent=2,
                                 dtype=np.uint32))
    assert len(out) == 1
    cvol, name, inst = out[0]
    ocons = osp.InstanceTracker(1, DIV, vol.shape, 'xy')
    ocons.instances = osp.merge_objects_from_trackers(list(otrackers.values()), 2, 0.75, False)
    osp.remove_small_objects(ocons, 30)
    osp.remove_pancakes(ocons, 2)
    _same_instances(inst, ocons.instances)
    np.testing.assert_array_equal(cvol, osp.numpy_fill_instances(np.zeros(vol.shape, np.uint32), ocons.instances))
    assert name == 'mito' and cvol.dtype == np.uint32


def test_engine2d_force_connected(model_config):
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine2d
    from oracle import sparse as osp
    img = synth.em_tiles(1, 200, seed=9)[0][:150, :170]      # needs factor padding (150x170 -> 160x176)
    eng = Engine2d(model_config, label_divisor=DIV, nms_kernel=3, confidence_thr=0.5)
    x = eng.preprocessor(img)['image'].unsqueeze(0)
    raw = eng.engine(x, img.shape, 1).squeeze(0).cpu().numpy().astype(np.int32)
    got = eng.infer(img)
    assert got.shape == img.shape and got.dtype == np.int32
    np.testing.assert_array_equal(got, osp.force_connected_pan(raw.copy(), [1], DIV))
    with pytest.raises(Exception, match='float'):
        eng.infer(img.astype(np.float32))                     # Preprocessor contract, utils.py:196-197
    # uint16 input is normalised by 65535 (utils.py:199-200); the raw-integer upload (normalisation fused into the
    # stem) must give the same labels as the host Preprocessor + factor_pad route
    img16 = img.astype(np.uint16) * 257
    x16 = eng.preprocessor(img16)['image'].unsqueeze(0)
    raw16 = eng.engine(x16, img16.shape, 1).squeeze(0).cpu().numpy().astype(np.int32)
    np.testing.assert_array_equal(eng.infer(img16), osp.force_connected_pan(raw16.copy(), [1], DIV))


def test_batched_slices_equal_single_slice_calls(model_config):
    """predict_slices (batched forward) == calling the 3d engine slice by slice like the reference loop"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d
    vol = synth.blob_volume(9, 32, 48, seed=4)
    eng = Engine3d(model_config, label_divisor=DIV, median_kernel_size=3, confidence_thr=0.5, batch_size=4)
    a = [p.cpu().numpy() for p in eng.predict_slices(vol, 0)]
    e = eng.engine
    b = []
    for i in range(vol.shape[0]):
        x = eng.preprocessor(vol[i])['image'].unsqueeze(0)
        r = e(x, vol[i].shape, 1)
        if r is not None:
            b.append(r[0].cpu().numpy())
    b += [s[0].cpu().numpy() for s in e.end(1)]
    e.reset()
    assert len(a) == len(b) == 9
    for p, q in zip(a, b):
        np.testing.assert_array_equal(p, q)


def test_engine2d_inference_scale_2(model_config):
    """inference_scale = 2: input down-scaled by resize_by_factor, PointRend renders one extra step
    (render_steps = 3) and cells are up-sampled x8, so the label map comes back at the original size."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine2d
    from empanada_napari_amd.preprocess import resize_by_factor
    from oracle import sparse as osp
    img = synth.em_tiles(1, 256, seed=12)[0][:250, :230]
    eng = Engine2d(model_config, inference_scale=2, label_divisor=DIV, nms_kernel=3, confidence_thr=0.5)
    got = eng.infer(img)
    assert got.shape == img.shape and got.dtype == np.int32
    small = resize_by_factor(img, 2)
    x = eng.preprocessor(small)['image'].unsqueeze(0)
    raw = eng.engine(x, img.shape, 2).squeeze(0).cpu().numpy().astype(np.int32)
    np.testing.assert_array_equal(got, osp.force_connected_pan(raw.copy(), [1], DIV))


def test_engine2d_tiled_inference_equals_oracle_pipeline(model_config):
    """Engine2d(tile_size>0): per-tile engine output -> RLE -> translation -> tile consensus on the product path
    equals the oracle's restatement of inference.py:283-318 fed with the same per-tile panoptic maps."""
    import numpy as np
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine2d
    from oracle import sparse as osp
    eng = Engine2d(model_config, label_divisor=1000, nms_kernel=3, confidence_thr=0.3, tile_size=128)
    img = synth.blob_image(208, 288, seed=77)
    out = eng.infer(img)
    assert out.shape == img.shape and out.dtype == np.uint32
    tiler = eng.last_tiler
    assert len(tiler) == 2 * 3
    segs = []
    for i in range(len(tiler)):
        tile = tiler(img, i)
        x = eng.preprocessor(tile)['image'].unsqueeze(0)
        pan = eng.engine(x, tile.shape, upsampling=1).squeeze(0).cpu().numpy().astype(np.int32)
        seg = osp.pan_seg_to_rle_seg(pan, eng.labels, 1000, eng.engine.thing_list, force_connected=True)
        segs.append(osp.translate_rle_seg(seg, tiler.yranges[i], tiler.xranges[i], img.shape))
    ref = {}
    ov = osp.calculate_overlap_rle(tiler.yranges, tiler.xranges, img.shape)
    for label in eng.labels:
        if label in eng.engine.thing_list:
            ref[label] = osp.merge_objects_from_tiles([s[label] for s in segs], ov)
        else:
            ref[label] = osp.merge_semantic_from_tiles([s[label] for s in segs])
    want = osp.rle_seg_to_pan_seg(ref, img.shape)
    assert np.array_equal(out, want)
    assert len(np.unique(out)) > 2, 'degenerate test image: no instances'
    # the batched tile route (raw integers, one forward per group of tiles) == the per-tile route
    assert np.array_equal(eng._infer_tiled(img, batched=False), out)


@pytest.mark.parametrize('dtype', [np.uint8, np.int16])
def test_multigpu_engine_single_rank_equals_engine3d(model_config, monkeypatch, dtype):
    """MultiGPUEngine3d on a one-rank group (the slab driver with no neighbours) must give Engine3d's trackers: same
    forward (raw-integer upload for uint8; the host Preprocessor route otherwise), per-slice recursive median instead
    of the run kernel, batched voting / merge, run lists through gather_object, C++ matcher on rank 0.
    The N > 1 exchange logic is covered on CPU with gloo (tests/test_multigpu_cpu.py)."""
    import os
    import socket
    import torch.distributed as dist
    from empanada_napari_amd import multigpu, synth
    from empanada_napari_amd.inference import Engine3d
    vol = synth.blob_volume(10, 40, 56, seed=11)
    if dtype is not np.uint8:
        vol = vol.astype(np.int16) * 128           # an integer type that goes through the host Preprocessor route
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        monkeypatch.setattr(multigpu.MultiGPUEngine3d, 'MIN_WORLD', 1)
        kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2)
        mg = multigpu.MultiGPUEngine3d(model_config, **kw)
        e3 = Engine3d(model_config, **kw)
        for axis in ('xy', 'yz'):
            _, ta = mg.infer_on_axis(vol, axis)
            _, tb = e3.infer_on_axis(vol, axis)
            assert len(tb[0].instances) > 0
            _same_instances(ta[0].instances, tb[0].instances)
        with pytest.raises(Exception, match='2 or more'):
            monkeypatch.setattr(multigpu.MultiGPUEngine3d, 'MIN_WORLD', 2)
            multigpu.MultiGPUEngine3d(model_config, **kw)
    finally:
        dist.destroy_process_group()


def test_engine3d_morphology_options(model_config):
    """label_erosion / label_dilation / fill_holes_in_segmentation run after the size filters, in the reference's order
    (empanada_napari/inference.py:560-570): same trackers as the oracle's filters applied to the plain engine's output"""
    import copy
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d
    from oracle import sparse as osp
    vol = synth.blob_volume(10, 48, 56, seed=21)
    kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2)
    _, plain = Engine3d(model_config, **kw).infer_on_axis(vol, 'xy')
    _, got = Engine3d(model_config, label_erosion=1, label_dilation=2, fill_holes_in_segmentation=True,
                      **kw).infer_on_axis(vol, 'xy')
    want = osp.InstanceTracker(1, DIV, vol.shape, 'xy')
    want.instances = copy.deepcopy(plain[0].instances)
    assert len(want.instances) > 0
    osp.erode(want, vol.shape, [1], DIV, [1], 1)
    osp.dilate(want, vol.shape, [1], DIV, [1], 2)
    osp.fill_holes_in_segmentation(want, vol.shape, [1], DIV, [1])
    _same_instances(got[0].instances, want.instances)


def test_multiclass_bifpn_orthoplane_consensus():
    """BASELINE configs[4] in small: PanopticBiFPN with 4 outputs (background + two instance classes + one semantic
    class), softmax / argmax hardening, one matcher + tracker per class, instance consensus for the thing classes and
    the pixel vote for the semantic class -- all against the oracle pipeline fed with the same per-slice label maps."""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    from oracle import sparse as osp
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=4)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    w, b = P['ins_center.head.1']
    P['ins_center.head.1'] = (w, b + np.float32(0.75))         # centres on small slices (see model_config above)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    labels, things = [1, 2, 3], [1, 2]
    mc = {'model': model, 'thing_list': things, 'labels': labels, 'class_names': {1: 'mito', 2: 'nucleus', 3: 'droplet'},
          'padding_factor': 128, 'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.blob_volume(12, 40, 36, seed=8, n_blobs=6)
    eng = Engine3d(mc, label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.3, min_size=8, min_extent=1)
    trackers, otrackers = {}, {}
    seen = set()
    for axis_name, axis in (('xy', 0), ('xz', 1), ('yz', 2)):
        pans = [p.cpu().numpy() for p in eng.predict_slices(vol, axis)]
        seen |= set(int(v) // DIV for v in np.unique(np.stack(pans)) if v > 0)
        _, trs = eng.infer_on_axis(vol, axis_name)
        ms = [osp.RLEMatcher(c, DIV, 0.25, 0.25) for c in things]
        stack = [osp.apply_matchers(osp.pan_seg_to_rle_seg(p, labels, DIV, things, force_connected=True), ms) for p in pans]
        for m in ms:
            m.target_rle, m.assign_new = None, False
        otrs = [osp.InstanceTracker(c, DIV, vol.shape, axis_name) for c in labels]
        for idx in range(len(pans) - 1, -1, -1):
            seg = osp.apply_matchers(stack[idx], ms)
            for t in otrs:
                t.update(seg[t.class_id], idx)
        for t, got in zip(otrs, trs):
            t.finish()
            osp.remove_small_objects(t, 8)      # the size filters apply to every tracker (inference.py:556-558)
            osp.remove_pancakes(t, 1)
            assert got.class_id == t.class_id
            _same_instances(got.instances, t.instances)
        trackers[axis_name], otrackers[axis_name] = trs, otrs
    assert len(seen) >= 2, f'the seeded network should produce several classes on this volume, got {seen}'
    out = list(tracker_consensus(trackers, None, mc, label_divisor=DIV, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                 allow_one_view=False, min_size=8, min_extent=1, dtype=np.uint32))
    assert [name for _, name, _ in out] == ['mito', 'nucleus', 'droplet']
    for (cvol, name, inst), c in zip(out, labels):
        cls = [t for ts in otrackers.values() for t in ts if t.class_id == c]
        want = osp.InstanceTracker(c, DIV, vol.shape, 'xy')
        if c in things:
            want.instances = osp.merge_objects_from_trackers(cls, 2, 0.75, False)
            osp.remove_small_objects(want, 8)
            osp.remove_pancakes(want, 1)
        else:
            want.instances = osp.merge_semantic_from_trackers(cls, 2)
        _same_instances(inst, want.instances)
        np.testing.assert_array_equal(cvol, osp.numpy_fill_instances(np.zeros(vol.shape, np.uint32), want.instances))


def test_config0_bifpn_single_512_tile_through_engine2d():
    """BASELINE configs[0]: MitoNet_mini-class (PanopticBiFPN) 2-D inference on one 512 x 512 tile through Engine2d.
    The label map must equal the oracle's post-processing (instance voting, merge, force_connected) of the engine's own
    head outputs bit for bit; the heads themselves are checked against the reference goldens in test_gpu_model.py."""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab, logits_to_prob
    from empanada_napari_amd.inference import Engine2d
    from oracle import postprocess as opp
    from oracle import sparse as osp
    cfg = dict(weights.MITONET_MINI_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    w, b = P['ins_center.head.1']
    P['ins_center.head.1'] = (w, b + np.float32(0.5))
    w, b = P['semantic_head.head.1']
    P['semantic_head.head.1'] = (w, b + np.float32(1.0))
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 128,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    img = synth.em_tiles(1, 512, seed=40)[0]
    eng = Engine2d(mc, label_divisor=DIV, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5)
    got = eng.infer(img)
    assert got.shape == (512, 512) and got.dtype == np.int32 and got.max() > DIV
    x = eng.preprocessor(img)['image'].unsqueeze(0).cuda()
    out = model(x, 2, interpolate_ins=False)
    sem = logits_to_prob(out['sem_logits']).cpu().numpy()
    oeng = opp.RenderEngine(None, [1], label_divisor=DIV, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                            coarse_boundaries=True)
    cells = oeng.cells(out['ctr_hmp'].cpu().numpy(), out['offsets'].cpu().numpy(), 1)
    pan = oeng.postprocess(sem, cells)[0].astype(np.int32)
    np.testing.assert_array_equal(got, osp.force_connected_pan(pan.copy(), [1], DIV))


def test_stack_postprocessing_matches_oracle(model_config):
    """inference.py:56-109: relabel 1..n (stable sort of runs, inference.py:31-54), size / span filters, dense fill."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d, stack_postprocessing
    from oracle import sparse as osp
    vol = synth.blob_volume(18, 40, 48, seed=8, n_blobs=5)
    eng = Engine3d(model_config, label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.5,
                   min_size=0, min_extent=0, batch_size=5)
    pans = [p.cpu().numpy() for p in eng.predict_slices(vol, 0)]
    _, trs = eng.infer_on_axis(vol, 'xy')
    out = list(stack_postprocessing({'xy': trs}, None, model_config, label_divisor=DIV, min_size=40, min_extent=3,
                                    dtype=np.uint32))
    assert len(out) == 1
    svol, name, inst = out[0]
    otr = _oracle_axis(pans, vol.shape, 'xy', 0, 0)
    ost = osp.InstanceTracker(1, DIV, vol.shape, 'xy')
    ost.instances = osp.instance_relabel(otr)
    osp.remove_small_objects(ost, 40)
    osp.remove_pancakes(ost, 3)
    assert len(ost.instances) > 0
    _same_instances(inst, ost.instances)
    np.testing.assert_array_equal(svol, osp.numpy_fill_instances(np.zeros(vol.shape, np.uint32), ost.instances))
    assert name == 'mito' and svol.dtype == np.uint32


@pytest.mark.parametrize('fmt', [2, 3])
def test_zarr_volume_in_zarr_store_out(model_config, tmp_path, fmt, monkeypatch):
    """BASELINE configs[2] says 'zarr volume': a zarr directory store (v2 uncompressed / v3 with gzip chunks) as the INPUT volume and ``store_url`` outputs
    (panoptic stack per axis, consensus / stack volumes; inference.py:100-103,404,474-489) give what the numpy route
    gives; the store is then read back through the files alone.  Median kernel 11 (the widget's maximum,
    _volume_inference.py:388) exercises the wide recursive-median instantiation."""
    import json
    from empanada_napari_amd import synth, zstore
    from empanada_napari_amd.inference import Engine3d, tracker_consensus, stack_postprocessing
    vol = synth.blob_volume(24, 40, 32, seed=12, n_blobs=6)
    monkeypatch.setenv('EMP_ZARR_FORMAT', str(fmt))      # the layout of the stores the engine creates
    src = zstore.open_store(str(tmp_path / 'in.zarr'), mode='w', zarr_format=fmt).create_array(
        'em', shape=vol.shape, dtype=np.uint8, chunks=(8, 16, 16), compressor='gzip' if fmt == 3 else None)
    src[...] = vol
    zvol = zstore.open_store(str(tmp_path / 'in.zarr'), mode='r')['em']
    kw = dict(label_divisor=DIV, median_kernel_size=11, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2,
              save_panoptic=True, batch_size=6)
    ref_eng = Engine3d(model_config, **kw)
    url = str(tmp_path / 'out.zarr')
    eng = Engine3d(model_config, store_url=url, chunk_size=(8, 16, 16), **kw)
    trackers, ref_trackers = {}, {}
    for axis_name in ('xy', 'xz', 'yz'):
        ref_stack, ref_trackers[axis_name] = ref_eng.infer_on_axis(vol, axis_name)
        stack, trackers[axis_name] = eng.infer_on_axis(zvol, axis_name)
        _same_instances(trackers[axis_name][0].instances, ref_trackers[axis_name][0].instances)
        np.testing.assert_array_equal(np.asarray(stack[...]), ref_stack)
        assert tuple(stack.chunks) == (8, 16, 16) and stack.dtype == np.int32
    ref = list(tracker_consensus(ref_trackers, None, model_config, label_divisor=DIV, pixel_vote_thr=2, min_size=20,
                                 min_extent=2, dtype=np.uint32))
    out = list(tracker_consensus(trackers, url, model_config, label_divisor=DIV, pixel_vote_thr=2, min_size=20,
                                 min_extent=2, dtype=np.uint32, chunk_size=(8, 16, 16)))
    np.testing.assert_array_equal(np.asarray(out[0][0][...]), ref[0][0])
    _same_instances(out[0][2], ref[0][2])
    # the files on disk are a zarr store any reader opens
    meta = json.load(open(tmp_path / 'out.zarr' / 'mito' / ('.zarray' if fmt == 2 else 'zarr.json')))
    assert meta['zarr_format'] == fmt and meta['shape'] == list(vol.shape)
    assert (meta['compressor'] is None) if fmt == 2 else (meta['codecs'] == [{'name': 'bytes', 'configuration': {'endian': 'little'}}])
    back = zstore.open_store(url, mode='r')
    assert set(back.array_keys()) >= {'mito', 'panoptic_xy', 'panoptic_xz', 'panoptic_yz'}
    np.testing.assert_array_equal(back['mito'][...], ref[0][0])
    sp = list(stack_postprocessing({'xy': trackers['xy']}, url, model_config, label_divisor=DIV, min_size=20,
                                   min_extent=2, dtype=np.uint32, chunk_size=(8, 16, 16)))
    sr = list(stack_postprocessing({'xy': ref_trackers['xy']}, None, model_config, label_divisor=DIV, min_size=20,
                                   min_extent=2, dtype=np.uint32))
    np.testing.assert_array_equal(np.asarray(sp[0][0][...]), sr[0][0])


def test_engine2d_infer_batch_equals_per_image_calls(model_config):
    """the pipelined batch API (uploads / downloads on side streams, fused force_connected) returns exactly what
    Engine2d.infer returns image by image, incl. a ragged last batch, non-multiple-of-16 sizes and uint16 input"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine2d
    from oracle import sparse as osp
    eng = Engine2d(model_config, label_divisor=DIV, nms_kernel=3, confidence_thr=0.5)
    imgs = [np.ascontiguousarray(t[:150, :170]) for t in synth.em_tiles(7, 200, seed=40)]
    want = [eng.infer(im) for im in imgs]
    got = eng.infer_batch(imgs, batch=3)
    assert len(got) == 7
    for a, b in zip(got, want):
        assert a.dtype == np.int32 and a.shape == (150, 170)
        np.testing.assert_array_equal(a, b)
    assert any(len(np.unique(a)) > 2 for a in got)
    imgs16 = [im.astype(np.uint16) * 257 for im in imgs[:2]]
    for a, im in zip(eng.infer_batch(imgs16, batch=8), imgs16):
        np.testing.assert_array_equal(a, eng.infer(im))
    assert eng.infer_batch([]) == []
    # multi-class ranges: the fused force_connected against the oracle on a synthetic two-class map
    from empanada_napari_amd import sparse as ps
    rng = np.random.default_rng(3)
    pan = np.zeros((2, 64, 80), np.int64)
    m = rng.random(pan.shape)
    pan[m > 0.55] = DIV + rng.integers(1, 4, size=int((m > 0.55).sum()))
    pan[m < 0.2] = 2 * DIV + rng.integers(1, 3, size=int((m < 0.2).sum()))
    pan[:, :8, :8] = 3 * DIV                                   # a stuff class stays as it is
    out = ps.force_connected(torch.from_numpy(pan).cuda(), [1, 2], DIV).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(out[i], osp.force_connected_pan(pan[i].copy(), [1, 2], DIV).astype(np.int32))


def _picklable_model_config():
    """model_config whose 'model' is a state dict (the spawned rank processes rebuild the engine from it)"""
    from empanada_napari_amd import weights
    cfg = dict(weights.MITONET_PDL_CFG)
    sd = weights.seeded_state_dict(cfg, seed=0)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 2.5), ('semantic_pr.point_head.predictor', 2.5)):
        sd[name + '.bias'] = sd[name + '.bias'] + np.float32(shift)
    return {'model': sd, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
            'norms': {'mean': 0.57571, 'std': 0.12765}}


@pytest.mark.parametrize('world', [1, 2])
def test_multigpu_engine_spawns_rccl_ranks(monkeypatch, world):
    """The public MultiGPUEngine3d API in its own launch mode (multigpu.py:214-238): the engine starts ``world`` rank
    processes (one per GPU, RCCL process group + gloo host group), each builds its engine from the pickled
    model_config; same trackers and panoptic stack as Engine3d in this process.  world = 2 needs two GPUs; world = 1
    runs the same machinery (spawn, RCCL init, shared-memory volume, procedural volume, persistent ranks) on one."""
    from empanada_napari_amd import multigpu, synth
    from empanada_napari_amd.inference import Engine3d
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs')
    mc = _picklable_model_config()
    monkeypatch.setattr(multigpu.MultiGPUEngine3d, 'MIN_WORLD', 1)
    kw = dict(label_divisor=DIV, median_kernel_size=5, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2)
    pv = synth.ProceduralVolume((14, 40, 56), seed=5, cell=16)
    vol = pv.block(0, 0, 14, 'cuda').cpu().numpy()           # the ranks synthesise their blocks on the GPU as well
    mg = multigpu.MultiGPUEngine3d(mc, world_size=world, save_panoptic=True, **kw)
    try:
        e3 = Engine3d(mc, stuff_area=32, save_panoptic=True, **kw)      # the multi-GPU flavour's stuff_area (Q11)
        for axis, v in (('xy', vol), ('xz', vol), ('xy', pv)):
            sa, ta = mg.infer_on_axis(v, axis)
            sb, tb = e3.infer_on_axis(vol, axis)
            assert len(tb[0].instances) > 0
            _same_instances(ta[0].instances, tb[0].instances)
            np.testing.assert_array_equal(sa, sb)
        assert all(p.is_alive() for p in mg._procs) and len(mg._procs) == world
    finally:
        mg.close()
    with pytest.raises(Exception, match='inference_scale'):
        bad = multigpu.MultiGPUEngine3d(mc, world_size=1, inference_scale=2, **kw)
        try:
            bad.infer_on_axis(vol, 'xy')
        finally:
            bad.close()


def _picklable_bifpn4_config():
    """BASELINE configs[4] in small: PanopticBiFPN with 4 outputs (background, two instance classes, one semantic class) as a
    state dict; the architecture is read from the state dict itself (weights.infer_cfg), as for an exported model"""
    from empanada_napari_amd import weights
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=4)
    sd = weights.seeded_state_dict(cfg, seed=3)
    sd['ins_center.head.1.bias'] = sd['ins_center.head.1.bias'] + np.float32(0.75)
    return {'model': sd, 'thing_list': [1, 2], 'labels': [1, 2, 3], 'class_names': {1: 'mito', 2: 'nucleus', 3: 'droplet'},
            'padding_factor': 128, 'norms': {'mean': 0.57571, 'std': 0.12765}}


def test_multigpu_two_ranks_multiclass_bifpn(monkeypatch):
    """BASELINE configs[4]'s multi-GPU leg in small (VERDICT r02 'configs untested'): the 4-class PanopticBiFPN through TWO
    real ranks of the slab pipeline on this box's GPU -- softmax / argmax hardening, one slab matcher per class on every
    rank (two thing classes chained down and up the ranks, the semantic class tracked only), per-slab tracks concatenated
    by the caller -- against Engine3d on the whole stack, every class, xy and yz."""
    from empanada_napari_amd import multigpu, synth
    from empanada_napari_amd.inference import Engine3d
    mc = _picklable_bifpn4_config()
    kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.3, min_size=8, min_extent=1)
    vol = synth.blob_volume(12, 40, 36, seed=8, n_blobs=6)
    mg = multigpu.MultiGPUEngine3d(mc, world_size=2, dist_backend='gloo', devices=[0, 0], **kw)
    try:
        e3 = Engine3d(mc, stuff_area=32, **kw)
        seen = 0
        for axis in ('xy', 'yz'):
            _, ta = mg.infer_on_axis(vol, axis)
            _, tb = e3.infer_on_axis(vol, axis)
            assert [t.class_id for t in ta] == [1, 2, 3]
            for a, b in zip(ta, tb):
                _same_instances(a.instances, b.instances)
                seen += len(b.instances)
        assert seen > 0
        assert len(mg.last_host_s) == 2
    finally:
        mg.close()


@pytest.mark.parametrize('ks', [3, 7])
def test_multigpu_two_ranks_on_one_gpu(monkeypatch, ks):
    """Two REAL ranks of the HIP slab pipeline (each its own process, engine and arena) sharing the one GPU of this box,
    gloo as the transport (device maps staged through the host): raw look-ahead to the previous rank, filtered carry to
    the next, in-place recursive median per slab, run lists gathered to the caller -- against Engine3d on the whole
    stack.  Everything of the N > 1 path except the RCCL transport itself (covered at world 1 above and on gloo in
    tests/test_multigpu_cpu.py)."""
    from empanada_napari_amd import multigpu, synth
    from empanada_napari_amd.inference import Engine3d
    mc = _picklable_model_config()
    kw = dict(label_divisor=DIV, median_kernel_size=ks, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2)
    vol = synth.ProceduralVolume((13, 40, 56), seed=9, cell=16).numpy()
    mg = multigpu.MultiGPUEngine3d(mc, world_size=2, dist_backend='gloo', devices=[0, 0], **kw)
    try:
        e3 = Engine3d(mc, stuff_area=32, **kw)
        for axis in ('xy', 'yz'):
            _, ta = mg.infer_on_axis(vol, axis)
            _, tb = e3.infer_on_axis(vol, axis)
            assert len(tb[0].instances) > 0
            _same_instances(ta[0].instances, tb[0].instances)
    finally:
        mg.close()


@pytest.mark.parametrize('batch', [None, 5])
def test_two_stream_axis_pipeline_equals_single_stream(model_config, monkeypatch, batch):
    """Engine3d.infer_on_axis keeps the forwards on the caller's stream and runs median / voting / merge / dense -> runs of
    a batch one batch behind on a second stream (EMP_STACK_TWO_STREAMS): the trackers must equal the single-stream run,
    also with several batches per axis (the group handed out lags the forward by one batch) and with a median window
    that spans batches."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d
    vol = synth.blob_volume(37, 48, 40, seed=3, n_blobs=24, fast=True)
    res = {}
    for two in ('0', '1'):
        monkeypatch.setenv('EMP_STACK_TWO_STREAMS', two)
        eng = Engine3d(model_config, label_divisor=DIV, median_kernel_size=5, nms_kernel=3, nms_threshold=0.1,
                       confidence_thr=0.5, min_size=20, min_extent=2, batch_size=batch)
        res[two] = {}
        for axis in ('xy', 'yz'):
            _, trs = eng.infer_on_axis(vol, axis)
            res[two][axis] = trs[0].instances
    for axis in ('xy', 'yz'):
        assert len(res['0'][axis]) > 0
        _same_instances(res['0'][axis], res['1'][axis])


@pytest.mark.parametrize('axis_name', ['xy', 'yz'])
def test_staged_host_volume_equals_whole_volume_upload(model_config, monkeypatch, axis_name):
    """A volume larger than EMP_VOLUME_ON_DEVICE_GIB is not uploaded whole: its batches go up through two pinned staging
    buffers on a side stream, one batch ahead (Engine3d._staged_batches).  Forced here with a limit
    of 0: trackers and the panoptic stack equal the whole-volume-upload path bit for bit, uint8 and uint16."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d
    vol = synth.blob_volume(23, 40, 56, seed=12)
    for v in (vol, vol.astype(np.uint16) * 257):
        kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, confidence_thr=0.5, min_size=20, min_extent=2,
                  save_panoptic=True, batch_size=5)
        monkeypatch.delenv('EMP_VOLUME_ON_DEVICE_GIB', raising=False)
        s0, t0 = Engine3d(model_config, **kw).infer_on_axis(v, axis_name)
        monkeypatch.setenv('EMP_VOLUME_ON_DEVICE_GIB', '0')
        s1, t1 = Engine3d(model_config, **kw).infer_on_axis(v, axis_name)
        assert len(t0[0].instances) > 0
        _same_instances(t0[0].instances, t1[0].instances)
        np.testing.assert_array_equal(s0, s1)
