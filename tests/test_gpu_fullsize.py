"""Properties at BASELINE.json's full sizes (1024^2 tiles in a batch, a 512 x 512 slab of a volume), where the oracle
is too slow to be the checker: a batch equals N sequential batch-1 calls (the reference asserts batch 1,
engines.py:306), runs are repeatable bit for bit, label maps obey the reference's numbering, dense -> runs -> dense is
the identity, and a small oracle-checked crop agrees with the corresponding full-size computation where the algorithm
is local."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DIV = 10000


@pytest.fixture(scope='module')
def engine():
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    return PanopticDeepLabRenderEngine(model, thing_list=[1], label_divisor=DIV, nms_threshold=0.1, nms_kernel=3,
                                       confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)


def test_batch_of_1024_tiles_equals_sequential_calls_and_is_repeatable(engine):
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize_params
    tiles = torch.from_numpy(synth.em_tiles(6, 1024, seed=77))[:, None].cuda()
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    a = engine.infer_batch(tiles, sub=float(sub), mul=float(mul)).clone()
    b = engine.infer_batch(tiles, sub=float(sub), mul=float(mul)).clone()
    assert torch.equal(a, b), 'two runs of the same batch differ'
    assert a.shape == (6, 1024, 1024) and a.dtype == torch.int64
    # head outputs too: a batch of six and a batch of one pick different conv tiles for the deep layers (256x256 needs
    # enough tiles to fill the chip), and every kernel variant keeps the same K order -- bit-identical
    full = {k: v.clone() for k, v in engine.model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul)).items()}
    single = engine.model(tiles[4:5], 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
    for k in full:
        assert torch.equal(single[k][0], full[k][4]), f'{k}: batch-1 forward differs from the batched one'
    for i in range(6):
        one = engine.call_raw(tiles[i:i + 1], sub, mul)           # the reference-style batch-1 call
        assert torch.equal(one[0], a[i]), f'tile {i}: batched result differs from the single call'
        ids = torch.unique(a[i])
        ids = ids[ids > 0]
        assert len(ids) > 0 and int(ids.min()) == DIV + 1 and int(ids.max()) == DIV + len(ids), \
            'instance ids of a class are 1..K without gaps (postprocess.py:260-281)'


def test_dense_runs_dense_round_trip_on_a_512_slab():
    """pan_stack_to_runs (CCL + run extraction) -> StackMatcher-free fill: rle_seg_to_pan_seg(pan_seg_to_rle_seg(x)) == CC(x)"""
    from empanada_napari_amd import sparse as ps
    rng = np.random.default_rng(5)
    D, H, W = 16, 512, 512
    zz, yy, xx = np.mgrid[0:D, 0:H, 0:W]
    vol = np.zeros((D, H, W), np.int32)
    for k in range(1, 41):
        c = rng.uniform(0, 1, 3) * (D, H, W)
        r = rng.uniform(0.05, 0.15, 3) * (D * 4, H, W)
        vol[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1] = 1000 + k
    vol[rng.random(vol.shape) < 0.02] = 0
    dv = torch.from_numpy(vol).cuda()
    cc, _ = ps.ccl8(dv)                                                      # per-slice components
    segs = ps.pan_stack_to_rle_segs(dv, [1], 1000, [1], force_connected=True)
    assert len(segs) == D
    for z in (0, D // 2, D - 1):
        back = ps.rle_seg_to_pan_seg(segs[z], (H, W)).astype(np.int64)
        want = cc[z].cpu().numpy().astype(np.int64)
        want[want > 0] += 1000
        np.testing.assert_array_equal(back, want)
        n_runs = sum(len(a['starts']) for a in segs[z][1].values())
        n_px = sum(int(a['runs'].sum()) for a in segs[z][1].values())
        assert n_px == int((vol[z] > 0).sum()) and n_runs < n_px


def test_volume_morphology_round_trip_at_512():
    """erode then dilate never grows an object beyond its original voxels' 1-neighbourhood and keeps runs sorted,
    boxes tight: checked on a 24 x 512 x 512 volume through tracker -> dense -> filter -> 26-CC -> runs"""
    from empanada_napari_amd import sparse as ps
    rng = np.random.default_rng(9)
    shape = (24, 512, 512)
    zz, yy, xx = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]]
    vol = np.zeros(shape, np.int32)
    for k in range(1, 25):
        c = rng.uniform(0.1, 0.9, 3) * shape
        r = rng.uniform(0.04, 0.1, 3) * (shape[0] * 6, shape[1], shape[2])
        vol[((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1] = 1000 + k
    tr = ps.InstanceTracker(1, 1000, shape, 'xy')
    tr.instances = ps.volume_to_instances(vol, [1], 1000, [1], force_connected=False)
    before = {k: int(v['runs'].sum()) for k, v in tr.instances.items()}
    assert sum(before.values()) == int((vol > 0).sum())
    ps.erode(tr, shape, [1], 1000, [1], iterations=1)
    eroded = sum(int(v['runs'].sum()) for v in tr.instances.values())
    assert 0 < eroded < sum(before.values())
    ps.dilate(tr, shape, [1], 1000, [1], iterations=1)
    dense = ps.tracker_to_volume(tr, shape).cpu().numpy()
    assert ((dense > 0) & (vol == 0)).sum() == 0, 'opening (erode, dilate) must stay inside the original objects'
    for a in tr.instances.values():
        assert np.all(np.diff(a['starts']) > 0)
        z, y, x = np.unravel_index(a['starts'], shape)
        assert a['box'][0] == z.min() and a['box'][1] == y.min() and a['box'][3] == z.max() + 1


def test_bifpn_batch_of_1024_tiles_is_repeatable_and_image_independent():
    """PanopticBiFPN at 1024^2: every fused path runs at a size where it is chosen (register-weight 3x3 convs, shortcut
    GEMMs, 3x3 and 5x5 separable-conv kernels with several tiles per workgroup); two runs agree bit for bit and an image's
    outputs do not depend on its neighbours in the batch"""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize_params
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=4)
    model = HipPanopticDeepLab(weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg), cfg, folded=True, precision='fp16')
    tiles = torch.from_numpy(synth.em_tiles(4, 1024, seed=5))[:, None].cuda()
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    a = {k: v.clone() for k, v in model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul)).items()}
    b = {k: v.clone() for k, v in model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul)).items()}
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert torch.equal(a[k], b[k]), f'{k}: two runs differ'
    one = model(tiles[2:3], 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
    for k in ('ctr_hmp', 'offsets'):
        # a batch of one takes the small-problem kernels for the deepest maps (fewer tiles than workgroups): same
        # arithmetic and K order, so still identical
        assert torch.equal(one[k][0], a[k][2]), k


def test_config2_batch_of_32_tiles_properties(engine):
    """BASELINE configs[1] at its own batch size (32 x 1024^2): repeatable bit for bit, three scattered tiles equal the
    reference-style single calls, ids of the class are 1..K without gaps on every tile, and the Engine2d pipeline
    (host tiles in, int32 maps out, force_connected) agrees with per-image Engine2d.infer on them."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine2d
    from empanada_napari_amd.preprocess import normalize_params
    host = synth.em_tiles(32, 1024, seed=99)
    tiles = torch.from_numpy(host)[:, None].cuda()
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    a = engine.infer_batch(tiles, sub=float(sub), mul=float(mul)).clone()
    b = engine.infer_batch(tiles, sub=float(sub), mul=float(mul))
    assert torch.equal(a, b) and a.shape == (32, 1024, 1024)
    for i in (0, 13, 31):
        assert torch.equal(engine.call_raw(tiles[i:i + 1], sub, mul)[0], a[i]), f'tile {i}'
    for i in range(32):
        ids = torch.unique(a[i])
        ids = ids[ids > 0]
        assert len(ids) > 0 and int(ids.min()) == DIV + 1 and int(ids.max()) == DIV + len(ids)
    mc = {'model': engine.model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    e2 = Engine2d(mc, label_divisor=DIV, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5)
    outs = e2.infer_batch(list(host), batch=32)
    for i in (0, 13, 31):
        np.testing.assert_array_equal(outs[i], e2.infer(host[i]))
        # force_connected only renumbers: same foreground as the engine-level map
        np.testing.assert_array_equal(outs[i] > 0, a[i].cpu().numpy() > 0)


def test_config3_512_cube_orthoplane_job_properties(engine, tmp_path):
    """BASELINE configs[2] at its own size (512^3, three axes + consensus), where no oracle can follow: the job is
    repeatable, the zarr-store route equals the numpy route, the consensus volume is exactly the fill of its instances,
    ids ascend, every instance passes the size / span filters and its runs are sorted, disjoint and inside its box."""
    from empanada_napari_amd import synth, zstore
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    mc = {'model': engine.model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.blob_volume(512, 512, 512, seed=0, n_blobs=256, fast=True)
    kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500,
              min_extent=5)
    e3 = Engine3d(mc, **kw)

    def job(v, url=None):
        trs = {name: e3.infer_on_axis(v, name)[1] for name in ('xy', 'xz', 'yz')}
        return list(tracker_consensus(trs, url, mc, label_divisor=DIV, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                      allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32,
                                      chunk_size=(256, 256, 256)))[0]
    cvol, name, inst = job(vol)
    cvol2, _, inst2 = job(vol)
    np.testing.assert_array_equal(cvol, cvol2)
    # consensus ids count the clusters from 1 in component order; the size / span filters then drop some (inference.py:139-147)
    assert list(inst) == list(inst2) and len(inst) > 0 and list(inst) == sorted(inst) and min(inst) >= 1
    src = zstore.open_store(str(tmp_path / 'em.zarr'), mode='w').create_array('em', shape=vol.shape, dtype=np.uint8,
                                                                               chunks=(256, 256, 256))
    src[...] = vol
    zvol, _, zinst = job(zstore.open_store(str(tmp_path / 'em.zarr'), mode='r')['em'], str(tmp_path / 'out.zarr'))
    np.testing.assert_array_equal(np.asarray(zvol[...]), cvol)
    want = np.zeros(vol.size, np.uint32)
    for k, a in inst.items():
        st, rn = np.asarray(a['starts']), np.asarray(a['runs'])
        assert int(rn.sum()) >= 500 and np.all(np.diff(st) > 0) and np.all(st[1:] >= (st + rn)[:-1])
        z, y, x = np.unravel_index(np.concatenate([st, st + rn - 1]), vol.shape)
        box = a['box']
        assert z.min() >= box[0] and y.min() >= box[1] and x.min() >= box[2] and z.max() < box[3] and y.max() < box[4] and x.max() < box[5]
        assert min(box[3] - box[0], box[4] - box[1], box[5] - box[2]) >= 5
        for s, r in zip(st, rn):
            want[s:s + r] = k
    np.testing.assert_array_equal(cvol.ravel(), want)


def test_config4_slices_of_4096_squared(engine, monkeypatch):
    """BASELINE configs[3] at its SLICE size (4096^2; the 4096-slice depth and the eight ranks are the driver's to run):
    one slice per forward batch, 64 MiB probability maps under the recursive median, thousands of objects per matcher
    step -- none of which a 1024^2 test touches.  Eight slices of a procedural volume:

      * batch invariance: two slices per forward == the reference-style loop of single-slice calls of the 3-D engine;
      * the xy job is repeatable, its panoptic stack is exactly the fill of its trackers, ids are unique and >= DIV + 1, every
        instance passes the filters, runs are disjoint and inside the box (the properties of the 512^3 test);
      * the sparse assignment solver (what the matcher runs) and the dense whole-matrix solve (scipy's algorithm as the
        reference calls it) give identical trackers at this object count;
      * one RCCL rank of the multi-GPU slab pipeline (block schedule) gives Engine3d's trackers."""
    import os
    import socket
    import torch.distributed as dist
    from empanada_napari_amd import multigpu, synth
    from empanada_napari_amd.inference import Engine3d
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    D, S = 8, 4096
    pv = synth.ProceduralVolume((D, S, S), seed=7, cell=48)
    vol = pv.block(0, 0, D, torch.device('cuda')).cpu().numpy()
    assert vol.shape == (D, S, S) and vol.dtype == np.uint8
    # the seeded network finds ~900 objects per 4096^2 slice; with the centre / semantic head biases lifted (as the 3-D
    # tests of test_gpu_engine3d.py do) it finds thousands -- the regime of configs[3]'s matcher steps
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
        w, b = P[name]
        P[name] = (w, b + np.float32(shift))
    del engine
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    kw = dict(label_divisor=DIV, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=200,
              min_extent=2)

    # ---- batch invariance at this size ----
    e3 = Engine3d(mc, batch_size=2, **kw)
    a = [p.cpu() for p in e3.predict_slices(vol[:4], 0)]
    e = e3.engine
    b = []
    for i in range(4):
        r = e(e3.preprocessor(vol[i])['image'].unsqueeze(0), vol[i].shape, 1)
        if r is not None:
            b.append(r[0].cpu())
    b += [s[0].cpu() for s in e.end(1)]
    e.reset()
    assert len(a) == len(b) == 4
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.equal(p, q), f'slice {i}: batched forward differs from the single-slice call'
    per_slice = [int((torch.unique(p) > 0).sum()) for p in a]
    print('objects per 4096^2 slice:', per_slice)
    assert min(per_slice) > 2000, f'the test needs thousands of objects per slice, got {per_slice}'
    del a, b, e3

    # ---- the xy job: repeatable, stack == fill of the trackers, instance properties ----
    e3 = Engine3d(mc, save_panoptic=True, **kw)
    stack, trs = e3.infer_on_axis(vol, 'xy')
    inst = trs[0].instances
    stack2, trs2 = e3.infer_on_axis(vol, 'xy')
    np.testing.assert_array_equal(stack, stack2)
    assert list(inst) == list(trs2[0].instances) and len(inst) > 1000
    ids = np.array(list(inst))
    assert ids.min() >= DIV + 1 and len(np.unique(ids)) == len(ids)      # tracker order = first seen by the backward pass
    want = np.zeros(vol.size, np.int32)
    for k, o in inst.items():
        st, rn = np.asarray(o['starts']), np.asarray(o['runs'])
        order = np.argsort(st, kind='stable')      # a tracker appends slice by slice in the backward pass's (descending) order
        st, rn = st[order], rn[order]
        assert int(rn.sum()) >= 200 and np.all(rn > 0) and np.all(st[1:] >= (st + rn)[:-1])      # disjoint runs
        z, y, x = np.unravel_index(np.concatenate([st, st + rn - 1]), vol.shape)
        box = o['box']
        assert z.min() >= box[0] and y.min() >= box[1] and x.min() >= box[2] and z.max() < box[3] and y.max() < box[4] and x.max() < box[5]
        assert min(box[3] - box[0], box[4] - box[1], box[5] - box[2]) >= 2
    # an independent fill (numpy): every run of every instance written once
    st = np.concatenate([np.asarray(o['starts']) for o in inst.values()])
    rn = np.concatenate([np.asarray(o['runs']) for o in inst.values()])
    vals = np.repeat(np.array(list(inst), np.int32), [len(o['starts']) for o in inst.values()])
    first = np.cumsum(rn) - rn
    idx = np.repeat(st - first, rn) + np.arange(int(rn.sum()))
    want[idx] = np.repeat(vals, rn)
    np.testing.assert_array_equal(stack.ravel(), want)
    del stack2, trs2, idx, want

    def same(x, y):
        assert [int(k) for k in x] == [int(k) for k in y]
        for k in x:
            assert tuple(int(v) for v in x[k]['box']) == tuple(int(v) for v in y[k]['box'])
            np.testing.assert_array_equal(np.asarray(x[k]['starts']), np.asarray(y[k]['starts']))
            np.testing.assert_array_equal(np.asarray(x[k]['runs']), np.asarray(y[k]['runs']))

    # ---- sparse vs dense assignment at thousands of objects per step ----
    monkeypatch.setenv('EMP_SM_FULL_LSA', '1')
    _, trd = Engine3d(mc, **kw).infer_on_axis(vol, 'xy')
    dense = trd[0].instances
    monkeypatch.delenv('EMP_SM_FULL_LSA')
    same(inst, dense)

    # ---- one RCCL rank of the slab pipeline ----
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        monkeypatch.setattr(multigpu.MultiGPUEngine3d, 'MIN_WORLD', 1)
        mg = multigpu.MultiGPUEngine3d(mc, **kw)
        _, tm = mg.infer_on_axis(vol, 'xy')
        same(tm[0].instances, inst)
        _, tp = mg.infer_on_axis(pv, 'xy')        # the procedural description (each rank synthesises its own slab)
        same(tp[0].instances, inst)
    finally:
        dist.destroy_process_group()
