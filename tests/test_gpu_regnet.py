"""RegNet encoders on the device (reference: empanada/models/encoders/regnet.py:38-316).  By default
they run in the library's fp32 mode (csrc/ref32.hip): the grouped 3x3 on the exact fp32 matrix pipe (one workgroup column
per group), the reference's per-pixel squeeze-excite gate, the 3x3 stride-2 stem with the normalisation fused -- checked
against the reference's own outputs (tests/golden/regnet_forward.npz: PanopticBiFPNPR / regnety_6p4gf and
PanopticDeepLabPR / regnetx_6p4gf) and, at a larger size, against the oracle's fp32 forward in the max norm.  On request
(precision='fp16') they run on the fp16 engine's generic convolutions: every layer teacher-forced to one fp16 ulp, the heads
against the fp32 forward at what they measure."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GROUPED = [
    # N, H, W, groups, group width in, group width out, stride, act
    (2, 20, 28, 3, 56, 56, 1, 1),       # regnetx stage 1: the group width is not a multiple of 16 (padded to 64)
    (1, 17, 33, 2, 72, 72, 2, 1),       # regnety stage 1, strided, odd sizes
    (1, 8, 8, 29, 56, 56, 1, 0),        # regnetx stage 4: 29 groups
    (2, 12, 12, 4, 16, 80, 1, 2),       # more outputs than one 64-column tile per group
]


@pytest.mark.parametrize('case', GROUPED)
def test_grouped_conv32_equals_torch(case):
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, G, ci, co, stride, act = case
    g = torch.Generator().manual_seed(sum(case))
    C = G * ci
    x = torch.randn((N, H, W, C), generator=g)
    w = torch.randn((G * co, ci, 3, 3), generator=g) / np.sqrt(ci * 9)
    b = torch.randn((G * co,), generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride, 1, 1, G)
    ref = torch.relu(ref) if act == 1 else ref * torch.sigmoid(ref) if act == 2 else ref
    ci16 = -(-ci // 16) * 16
    ld = -(-C // 16) * 16 + 16
    xp = torch.zeros((N, H, W, ld))
    xp[..., :C] = x
    wp = torch.zeros((G * co, 9, ci16))
    wp[..., :ci] = w.permute(0, 2, 3, 1).reshape(G * co, 9, ci)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.full((N, Ho, Wo, G * co + 4), 7.0, device=dev())
    xd, wd, bd = xp.to(dev()), wp.to(dev()), b.to(dev())
    _abi.check(lib.emp_conv2d_grouped_nhwc_f32(_abi.ptr(xd), N, H, W, G, ci, ci16, ld, _abi.ptr(wd), _abi.ptr(bd), _abi.ptr(out),
                                               G * co + 4, co, 3, 3, stride, 1, 1, act, _abi.stream_ptr(dev())), 'grouped conv32')
    torch.cuda.synchronize()
    got = out[..., :G * co].cpu().permute(0, 3, 1, 2).double()
    assert torch.all(out[..., G * co:] == 7.0)
    assert float((got - ref).abs().max()) < 4e-6 * float(ref.abs().max()) * np.sqrt(ci * 9 / 64.0 + 1.0)
    # a row too short for the last group's padded read is refused, not read past
    assert lib.emp_conv2d_grouped_nhwc_f32(_abi.ptr(xd), N, H, W, G, ci, ci16, (G - 1) * ci + ci16 - 4, _abi.ptr(wd), _abi.ptr(bd),
                                           _abi.ptr(out), G * co + 4, co, 3, 3, stride, 1, 1, act, _abi.stream_ptr(dev())) != 0 or ci16 == ci


def _model(tag, precision='fp32'):
    from test_regnet import regnet_model
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg, P = regnet_model(tag)
    return cfg, P, HipPanopticDeepLab(P, cfg, folded=True, precision=precision)


@pytest.mark.parametrize('tag', ['y', 'x'])
def test_regnet_forward_matches_the_reference_goldens(golden_dir, tag):
    from empanada_napari_amd.preprocess import normalize
    g = np.load(os.path.join(golden_dir, 'regnet_forward.npz'))
    cfg, P, model = _model(tag)
    assert model.precision == 'fp32'          # the exact comparator mode (the default is 'fp16x3', precision='fp16' the throughput opt-in)
    for case in 'ab':
        img = g[f'{tag}{case}_image']
        rs, interp = int(g[f'{tag}{case}_render_steps']), bool(g[f'{tag}{case}_interpolate_ins'])
        x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
        out = model(x.cuda(), rs, interp)
        for name in ('ctr_hmp', 'offsets'):
            ref = g[f'{tag}{case}_{name}']
            got = out[name].cpu().numpy()
            assert got.shape == ref.shape
            scale = max(1.0, float(np.sqrt((ref.astype(np.float64) ** 2).mean())))
            assert float(np.abs(got - ref).max()) < 2e-4 * scale, (tag, case, name, float(np.abs(got - ref).max()), scale)
        ref = g[f'{tag}{case}_sem_logits']
        got = out['sem_logits'].cpu().numpy()
        assert got.shape == ref.shape
        # PointRend refines the most uncertain cells: two fp32 forwards pick the same ones up to near-ties of the uncertainty
        assert float((np.abs(got - ref) > 1e-3 * max(1.0, float(np.abs(ref).max()))).mean()) < 2e-3
        # raw uint8 in, normalisation inside the 3x3 stem
        raw = model(torch.from_numpy(img)[:, None].cuda(), rs, interp, sub=0.57571 * 255, mul=1.0 / (0.12765 * 255))
        for name in ('ctr_hmp', 'offsets'):
            assert float((raw[name] - out[name]).abs().max()) < 1e-3 * max(1.0, float(out[name].abs().max()))


@pytest.mark.parametrize('tag,size', [('y', 256), ('x', 224)])
def test_regnet_heads_within_1e3_of_the_fp32_oracle(tag, size):
    """the north star's gate in the max norm, at a size with ragged tiles (224 = 7 x 32: stage 4 is 7 x 7)"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg, P, model = _model(tag)
    img = synth.em_tiles(2, size, seed=5)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu().numpy() for k, v in model(x.cuda(), 2, False).items()}
    taps = {}
    ref = pdl_model.model_forward(P, x, cfg, 2, False, taps)
    for k in ('ctr_hmp', 'offsets'):
        scale = max(1.0, float(ref[k].pow(2).mean().sqrt()))
        assert float(np.abs(out[k] - ref[k].numpy()).max()) / scale < 1e-4, k
    ncls = cfg['num_classes']
    coarse = model.tap_raw('semantic_head.out', (2, ncls, size // 4, size // 4)).cpu()
    if ncls == 1:
        e = (torch.sigmoid(coarse) - torch.sigmoid(taps['sem_coarse'])).abs()
    else:
        e = (torch.softmax(coarse, 1) - torch.softmax(taps['sem_coarse'], 1)).abs()
    assert float(e.max()) < 1e-4
    # the last encoder map itself
    name = 'encoder.stage4.block%d' % cfg['regnet']['depths'][3]
    w4 = cfg['regnet']['widths'][3]
    h4 = -(-size // 32)
    ld = -(-w4 // 16) * 16 + 16
    p5 = model.tap_raw(name, (2, h4, h4, ld)).cpu()
    want = pdl_model.regnet_forward(P, x, cfg['regnet'])[4].permute(0, 2, 3, 1)
    assert float((p5[..., :w4] - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    assert float(p5[..., w4:].abs().max()) == 0.0          # the row tails other kernels rely on stay zero


@pytest.mark.parametrize('tag,size', [('x', 256), ('y', 256)])
def test_regnet_on_the_fp16_engine(tag, size):
    """precision='fp16': the RegNet encoder on the fp16 engine's generic convolutions (one launch per group for the 3x3).
    Against the fp32 forward its heads carry the fp16 engine's usual storage error; gated at 1.3x what was measured (the
    fp32 mode, the default for a RegNet, is the one that meets 1e-3 in the max norm)"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from test_regnet import regnet_model
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg, P = regnet_model(tag)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    assert model.precision == 'fp16'
    img = synth.em_tiles(2, size, seed=5)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu() for k, v in model(x.cuda(), 2, False).items()}
    ref = pdl_model.model_forward(P, x, cfg, 2, False)
    rep = {}
    for k in ('ctr_hmp', 'offsets'):
        scale = max(1.0, float(ref[k].pow(2).mean().sqrt()))
        d = (out[k] - ref[k]).abs()
        rep[k] = (float(d.pow(2).mean().sqrt()) / scale, float(d.max()) / scale)
    print('fp16 RegNet', tag, rep)
    assert rep['ctr_hmp'][0] < GATE[tag][0] and rep['offsets'][0] < GATE[tag][1], rep
    # the last encoder map against the oracle's, and its zero row tail
    w4 = cfg['regnet']['widths'][3]
    p5 = model.tap('encoder.stage4.block%d' % cfg['regnet']['depths'][3]).float().cpu()
    want = pdl_model.regnet_forward(P, x, cfg['regnet'])[4].permute(0, 2, 3, 1)
    assert p5.shape == want.shape
    assert float((p5 - want).pow(2).mean().sqrt()) < 4e-3 * float(want.pow(2).mean().sqrt())
    # batch invariance: one image alone == the same image in the batch, bit for bit
    one = model(x[:1].cuda(), 2, False)
    for k in ('ctr_hmp', 'offsets'):
        assert torch.equal(one[k].cpu(), out[k][:1]), k
    # raw uint8 in
    raw = model(torch.from_numpy(img)[:, None].cuda(), 2, False, sub=0.57571 * 255, mul=1.0 / (0.12765 * 255))
    assert float((raw['ctr_hmp'].cpu() - out['ctr_hmp']).abs().max()) < 2e-2


# ctr rms, offsets rms relative to the map's scale: 1.3 x the measured 0.58e-3 / 0.73e-3 (PDL-PR on regnetx) and 1.38e-3 / 0.84e-3
# (BiFPN-PR on regnety)
GATE = {'x': (0.75e-3, 0.95e-3), 'y': (1.8e-3, 1.1e-3)}


def test_regnet_pdl_at_a_size_the_stride_does_not_divide():
    """PanopticDeepLab pads to multiples of 16 (configs/*.yaml: padding_factor) while a RegNet's last stage sits at stride
    32: 80 x 112 gives a 3 x 4 stage-4 map (3x3 stride-2 pad-1 conv and 1x1 stride-2 shortcut both round up), which the
    decoder resamples to the stride-4 map's 20 x 28 -- same as the reference's convolution arithmetic (the oracle's)"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    cfg, P, model = _model('x')
    img = synth.em_tiles(1, 112, seed=9)[:, :80]
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu() for k, v in model(x.cuda(), 2, True).items()}
    ref = pdl_model.model_forward(P, x, cfg, 2, True)
    for k in ('ctr_hmp', 'offsets'):
        assert out[k].shape == ref[k].shape == (1, 1 if k == 'ctr_hmp' else 2, 80, 112)
        assert float((out[k] - ref[k]).abs().max()) < 1e-4 * max(1.0, float(ref[k].pow(2).mean().sqrt())), k
    # the same on the fp16 engine (its planner rounds the strided sizes up the same way)
    from empanada_napari_amd.engines import HipPanopticDeepLab
    m16 = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    o16 = {k: v.cpu() for k, v in m16(x.cuda(), 2, True).items()}
    for k in ('ctr_hmp', 'offsets'):
        assert o16[k].shape == ref[k].shape
        assert float((o16[k] - ref[k]).pow(2).mean().sqrt()) < 2e-3 * max(1.0, float(ref[k].pow(2).mean().sqrt())), k


def test_regnet_state_dict_through_the_public_engines():
    """a RegNet model handed to the engines the way the widgets do (model_config['model'] = the unfused state dict: the
    architecture is inferred, the fp32 mode chosen): Engine2d label map == the oracle's post-processing of the oracle's
    fp32 forward up to near-ties; Engine3d runs a small stack through the same model"""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.inference import Engine2d, Engine3d
    from oracle import pdl_model, postprocess as opp, sparse as osp
    cfg0 = dict(weights.MITONET_PDL_CFG, encoder='regnetx_6p4gf')
    sd = weights.seeded_state_dict(cfg0, seed=12)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.5), ('semantic_pr.point_head.predictor', 1.5)):
        sd[name + '.bias'] = sd[name + '.bias'] + np.float32(shift)
    mc = {'model': sd, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    eng = Engine2d(mc, label_divisor=1000, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5)
    assert eng.engine.model.precision == 'fp16x3' and eng.engine.model.cfg['encoder'] == 'regnetx_6p4gf'
    img = synth.em_tiles(1, 256, seed=21)[0][:200, :232]
    got = eng.infer(img)
    assert got.shape == img.shape and got.dtype == np.int32
    cfg = dict(eng.engine.model.cfg)
    P = weights.fold_state_dict(sd, cfg)
    x = eng.preprocessor(img)['image'].unsqueeze(0)
    assert x.ndim == 4
    xp = torch.nn.functional.pad(x, (0, 240 - 232, 0, 208 - 200))
    r = {k: v.numpy() for k, v in pdl_model.model_forward(P, xp, cfg, 2, False).items()}
    r['sem'] = opp.logits_to_prob(r['sem_logits'])
    oeng = opp.RenderEngine(lambda *_: r, [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                            coarse_boundaries=True)
    want = oeng.postprocess(r['sem'], oeng.cells(r['ctr_hmp'], r['offsets'], 1))[0][:200, :232]
    want = osp.force_connected_pan(want.astype(np.int32).copy(), [1], 1000)
    assert len(np.unique(want)) > 5
    assert float(((got > 0) != (want > 0)).mean()) < 1e-3 and abs(len(np.unique(got)) - len(np.unique(want))) <= 2
    vol = synth.blob_volume(6, 64, 96, seed=4)
    e3 = Engine3d(dict(mc, model=eng.engine.model), label_divisor=1000, median_kernel_size=3, confidence_thr=0.5)
    stack, trackers = e3.infer_on_axis(vol, 'xy')
    assert len(trackers) == 1 and trackers[0].class_id == 1


def test_regnet_buffers_are_clean_after_a_forward_of_another_shape():
    """RegNet maps live in rows whose tails (channels padded to 16, plus the grouped conv's read-ahead) must read as
    zeros; a buffer kept from a larger forward is cleared when the geometry changes -- same bits as a fresh network"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    cfg, P, model = _model('x')
    big = torch.from_numpy(normalize(synth.em_tiles(2, 192, seed=1), 0.57571, 0.12765))[:, None].cuda()
    small = torch.from_numpy(normalize(synth.em_tiles(1, 128, seed=2), 0.57571, 0.12765))[:, None].cuda()
    model(big, 2, False)
    got = {k: v.clone() for k, v in model(small, 2, False).items()}
    w1 = cfg['regnet']['widths'][0]
    ld = -(-w1 // 16) * 16 + 16
    a = model.tap_raw('encoder.stage1.block1.a', (1, 64, 64, ld))
    assert float(a[..., w1:].abs().max()) == 0.0 and float(a[..., :w1].abs().max()) > 0.0
    _, _, fresh = _model('x')
    want = fresh(small, 2, False)
    for k in got:
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize('tag', ['x', 'y'])
def test_regnet_fp16_layers_teacher_forced(tag):
    """Every layer of the RegNet encoder on the fp16 engine, fed the ENGINE'S OWN input map (teacher forcing): the
    reference's arithmetic (regnet.py:51-97, blocks.py:35-50) in fp32 on those inputs with the weights as the engine
    stores them (fp16), rounded once to fp16, equals the engine's output map to one fp16 ulp (+1e-4 of the map's scale: a
    sum added in another order can land on the other side of a rounding boundary) -- layer by layer, so no error of one
    layer can hide behind the next: a (1x1), b (grouped 3x3, per-group launches on channel slices) [+ the per-pixel gate],
    the shortcut, c (1x1 + shortcut + ReLU)."""
    import torch.nn.functional as F
    from empanada_napari_amd import synth
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from test_regnet import regnet_model
    cfg, P = regnet_model(tag)
    r = cfg['regnet']
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    size = 128 if tag == 'y' else 96
    x = torch.from_numpy(normalize(synth.em_tiles(2, size, seed=8), 0.57571, 0.12765))[:, None]
    model(x.cuda(), 2, False)
    tap = lambda name: model.tap(name).float().cpu().permute(0, 3, 1, 2).contiguous()
    w16 = lambda name: (torch.from_numpy(P[name][0]).half().float(), torch.from_numpy(P[name][1]))
    r16 = lambda t: t.half().float()

    def check(name, got, want):
        d = (got - want).abs()
        rms = float(want.pow(2).mean().sqrt())
        ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
        excess = float((d - ulp - 1e-4 * max(1.0, rms)).max())
        assert excess <= 0, f'{name}: off by more than one fp16 ulp (max |d| {float(d.max()):.3e}, rms {rms:.3f})'

    w, b = P['encoder.stem.cbr.0']
    check('stem', tap('stem'), r16(F.relu(F.conv2d(x, torch.from_numpy(w), torch.from_numpy(b), 2, 1))))      # fp32 weights
    xin_name, n_checked = 'stem', 1
    for si, (d, g) in enumerate(zip(r['depths'], r['groups']), start=1):
        for bi in range(1, d + 1):
            p = f'encoder.stage{si}.block{bi}'
            s = 2 if bi == 1 else 1
            xin = tap(xin_name)
            wa, ba = w16(f'{p}.bottleneck.a.0')
            check(p + '.a', tap(p + '.a'), r16(F.relu(F.conv2d(xin, wa, ba))))
            wb, bb = w16(f'{p}.bottleneck.b.0')
            check(p + '.b', tap(p + '.b'), r16(F.relu(F.conv2d(tap(p + '.a'), wb, bb, s, 1, 1, g))))
            bname = p + '.b'
            if r['use_se']:      # the gated map is written over the gate (.se2); .b keeps the pre-gate map
                w0, b0 = w16(f'{p}.bottleneck.se.se.0')
                w2, b2 = w16(f'{p}.bottleneck.se.se.2')
                ns = w0.shape[0]
                check(p + '.se1', tap(p + '.se1')[:, :ns], r16(F.relu(F.conv2d(tap(p + '.b'), w0, b0))))
                assert tap(p + '.se1').shape[1] == -(-ns // 8) * 8 and not bool(tap(p + '.se1')[:, ns:].any())      # the squeeze width's padding to 8
                # (the gate logits themselves are gone -- overwritten by the gated map: two steps in one check)
                gate = r16(F.conv2d(tap(p + '.se1')[:, :ns], w2, b2))
                want = r16(tap(p + '.b') * torch.sigmoid(gate))
                dg = (tap(p + '.se2') - want).abs()
                # one ulp of the gate logit moves sigmoid by <= 2^-12 relative: two ulps of the product
                assert float((dg - 2 * torch.maximum(want.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10 - 1e-4).max()) <= 0, p
                bname = p + '.se2'
            wc, bc = w16(f'{p}.bottleneck.c.0')
            if f'{p}.downsample.conv.0' in P:
                wd, bd = w16(f'{p}.downsample.conv.0')
                short = r16(F.conv2d(xin, wd, bd, s))
                check(p + '.ds', tap(p + '.ds'), short)
                short = tap(p + '.ds')
            else:
                short = xin
            check(p, tap(p), r16(F.relu(F.conv2d(tap(bname), wc, bc) + short)))
            xin_name = p
            n_checked += 4
    assert n_checked > 60
