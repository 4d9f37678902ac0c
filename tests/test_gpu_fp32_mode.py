"""The fp32 REFERENCE MODE of the library (csrc/ref32.hip, emp_pdl_set_precision / HipPanopticDeepLab(precision='fp32');
VERDICT r03 item 7).  The reference computes this path in fp32 (empanada/inference/engines.py:248-255); the fp16 engine meets
the north star's "within 1e-3 on the float semantic / center heatmaps" in rms only.  In this mode every map and weight is
fp32 and every product runs on the exact fp32 matrix pipe, so the gate holds in the MAX norm -- asserted here against the
oracle's fp32 forward (itself pinned by the reference goldens), for both network families, up to BASELINE's tile size."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _sig(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, act, res
    (2, 20, 28, 64, 64, 1, 1, 0, 1, 1, False),
    (1, 33, 17, 48, 96, 3, 1, 1, 1, 1, True),       # odd sizes, ragged 64-pixel / 64-cout tiles, residual
    (2, 16, 16, 32, 40, 3, 2, 1, 1, 0, False),      # stride 2, Cout not a multiple of 32
    (1, 24, 24, 64, 128, 3, 1, 4, 4, 2, False),     # dilation 4, SiLU
    (3, 8, 8, 256, 16, 1, 1, 0, 1, 0, False),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv32_equals_torch_fp32(case):
    """the generic conv of the fp32 mode against torch's fp32 conv on the same operands (fp64 reference for the scale):
    an fp32 fmaf chain per output in another order -- agreement to a few fp32 ulps of the accumulated magnitude"""
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    N, H, W, Cin, Cout, k, stride, pad, dil, act, res = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g)
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    r = torch.randn((N, Ho, Wo, Cout), generator=g) if res else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride, pad, dil)
    if res:
        ref = ref + r.permute(0, 3, 1, 2).double()
    if act == 1:
        ref = torch.relu(ref)
    elif act == 2:
        ref = ref * torch.sigmoid(ref)
    xd, bd = x.to(dev()), b.to(dev())
    wd = w.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin).contiguous().to(dev())
    rd = r.to(dev()) if res else None
    out = torch.full((N, Ho, Wo, Cout + 8), 7.0, device=dev())       # a channel slice of a wider buffer
    _abi.check(lib.emp_conv2d_nhwc_f32(_abi.ptr(xd), N, H, W, Cin, Cin, _abi.ptr(wd), _abi.ptr(bd), None,
                                       _abi.ptr(rd) if res else None, Cout, _abi.ptr(out), Cout + 8, Cout, k, k, stride, pad,
                                       dil, act, _abi.stream_ptr(dev())), 'conv32')
    torch.cuda.synchronize()
    got = out[..., :Cout].cpu().permute(0, 3, 1, 2).double()
    assert torch.all(out[..., Cout:] == 7.0), 'wrote outside its channel slice'
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 2e-6 * scale * np.sqrt(Cin * k * k / 64.0 + 1.0), float((got - ref).abs().max())


def test_conv32_refuses_unpadded_channels():
    from gpu_common import dev
    from empanada_napari_amd import _abi
    lib = _abi.load()
    x = torch.zeros((1, 4, 4, 24), device=dev())
    w = torch.zeros((8, 1, 24), device=dev())
    o = torch.zeros((1, 4, 4, 8), device=dev())
    assert lib.emp_conv2d_nhwc_f32(_abi.ptr(x), 1, 4, 4, 24, 24, _abi.ptr(w), None, None, None, 0, _abi.ptr(o), 8, 8, 1, 1, 1, 0,
                                   1, 0, _abi.stream_ptr(dev())) != 0


def _models(family, ncls=1):
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    if family == 'pdl':
        cfg = dict(weights.MITONET_PDL_CFG)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    else:
        cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
        w, b = P[name]
        P[name] = (w, b + np.float32(shift))
    return cfg, P, HipPanopticDeepLab(P, cfg, folded=True, precision='fp32')


def _check_heads(out, ref, taps, ncls, coarse):
    rep = {}
    for k in ('ctr_hmp', 'offsets'):
        d = np.abs(out[k] - ref[k].numpy())
        scale = max(1.0, float(ref[k].pow(2).mean().sqrt()))
        rep[k] = (float(d.max()) / scale, float(np.sqrt((d ** 2).mean())) / scale)
        assert rep[k][0] < TOL, (k, rep[k])                      # the north star's gate, in the MAX norm
        assert rep[k][0] < 1e-4, (k, rep[k])                     # ... and what an all-fp32 forward actually measures
    if ncls == 1:
        e = np.abs(_sig(coarse) - _sig(taps['sem_coarse'].numpy()))
    else:
        e = np.abs(torch.softmax(torch.from_numpy(coarse), 1).numpy() - torch.softmax(taps['sem_coarse'], 1).numpy())
    rep['sem_coarse_prob_max'] = float(e.max())
    assert rep['sem_coarse_prob_max'] < 1e-4, rep
    # the final map: PointRend picks the 8192 most uncertain cells per step; with all-fp32 inputs the two sides pick the
    # same cells except at fp32 near-ties of the uncertainty
    if ncls == 1:
        pe = np.abs(_sig(out['sem_logits']) - _sig(ref['sem_logits'].numpy()))
    else:
        pe = np.abs(torch.softmax(torch.from_numpy(out['sem_logits']), 1).numpy() - torch.softmax(ref['sem_logits'], 1).numpy())
    rep['final_prob_frac_over_1e3'] = float((pe > TOL).mean())
    assert rep['final_prob_frac_over_1e3'] < 2e-4, rep
    return rep


@pytest.mark.parametrize('family,ncls,size', [('pdl', 1, 256), ('bifpn', 1, 256), ('bifpn', 4, 384)])
def test_fp32_mode_heads_within_1e3_max_norm(family, ncls, size):
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    cfg, P, model = _models(family, ncls)
    assert model.precision == 'fp32'
    img = synth.em_tiles(2, size, seed=11)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu().numpy() for k, v in model(x.cuda(), 2, False).items()}
    coarse = model.tap_raw('semantic_head.out', (2, ncls, size // 4, size // 4)).cpu().numpy()
    taps = {}
    ref = pdl_model.model_forward(P, x, cfg, 2, False, taps)
    rep = _check_heads(out, ref, taps, ncls, coarse)
    print(family, ncls, size, rep)
    # raw uint8 input (normalisation inside the stem) == the normalised float input, and interpolate_ins works
    a = model(torch.from_numpy(img)[:, None].cuda(), 2, True, sub=0.57571 * 255, mul=1.0 / (0.12765 * 255))
    b = model(x.cuda(), 2, True)
    for k in a:
        assert a[k].shape == b[k].shape
        assert float((a[k] - b[k]).abs().max()) < 1e-3 * max(1.0, float(b[k].abs().max())), k
    with pytest.raises(Exception):
        model.tap('encoder.layer1.0')          # fp16 taps do not exist in this mode


def test_fp32_mode_at_baseline_tile_size_and_label_maps():
    """BASELINE configs[1]'s tile (1024^2): ctr / semantic within 1e-3 of the fp32 oracle in the max norm, and the label
    map of the fp32-mode heads through the HIP post-processing == the oracle pipeline's label map up to fp32 near-ties."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model, postprocess as opp
    torch.set_num_threads(min(__import__('os').cpu_count() or 1, 32))
    cfg, P, model = _models('pdl')
    img = synth.em_tiles(1, 1024, seed=2024)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu().numpy() for k, v in model(x.cuda(), 2, False).items()}
    coarse = model.tap_raw('semantic_head.out', (1, 1, 256, 256)).cpu().numpy()
    taps = {}
    ref = pdl_model.pdl_forward(P, x, cfg, 2, False, taps)
    rep = _check_heads(out, ref, taps, 1, coarse)
    print('fp32 mode @1024^2:', rep)
    eng = PanopticDeepLabRenderEngine(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                                      padding_factor=16, coarse_boundaries=True)
    pan = eng(x, img.shape[-2:], 1).cpu().numpy()[0]
    r = {k: v.numpy() for k, v in ref.items()}
    r['sem'] = opp.logits_to_prob(r['sem_logits'])
    oeng = opp.RenderEngine(lambda *_: r, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                            coarse_boundaries=True)
    want = oeng.postprocess(r['sem'], oeng.cells(r['ctr_hmp'], r['offsets'], 1))[0]
    n_hip, n_ref = len(np.unique(pan)) - 1, len(np.unique(want)) - 1
    fg = float(((pan > 0) != (want > 0)).mean())
    print(f'fp32 mode label maps: {n_hip} vs {n_ref} instances, foreground flips {fg:.2e}')
    assert n_ref > 100 and abs(n_hip - n_ref) <= 2
    assert fg < 2e-4          # the fp16 engine: 1.3e-3 of the pixels (tests/test_gpu_parity_fullsize.py)


def test_strip_depthwise_kernel_is_the_plain_kernels_fmaf_chain(monkeypatch):
    """the fp32 mode's depthwise strip kernel (8 outputs per thread, W % 8 == 0) against the one-output kernel it
    replaced (EMP_DW32_STRIP=0), through the whole fp32-mode forward: the same fmaf chain (ky-major, kx-minor, from 0) -> the same
    bits in every head on finite inputs (they differ only for -0.0 / inf / NaN taps, which no network produces)"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    cfg, P, model = _models('pdl')
    x = torch.from_numpy(normalize(synth.em_tiles(2, 256, seed=23), 0.57571, 0.12765))[:, None].cuda()
    monkeypatch.setenv('EMP_DW32_STRIP', '1')
    a = {k: v.clone() for k, v in model(x, 2, False).items()}
    monkeypatch.setenv('EMP_DW32_STRIP', '0')
    b = model(x, 2, False)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('arch', ['pdl', 'bifpn'])
def test_four_column_bilinear_kernel_is_the_plain_kernels_arithmetic(monkeypatch, arch):
    """Round 6: the fp32 / fp16x3 graph's up-sampler computes four output columns per thread from at most three input columns
    (bilinear32x4_kernel, csrc/ref32.hip) -- the same loads and the same expression per output as the one-output kernel
    (EMP_BILINEAR32_X4=0), so every head is bit-identical, in both precisions that use it and on an odd-ratio size"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg, P, _ = _models(arch)
    for prec, size in (('fp32', 256), ('fp16x3', 384)):
        model = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
        x = torch.from_numpy(normalize(synth.em_tiles(2, size, seed=29), 0.57571, 0.12765))[:, None].cuda()
        monkeypatch.setenv('EMP_BILINEAR32_X4', '1')
        a = {k: v.clone() for k, v in model(x, 2, False).items()}
        monkeypatch.setenv('EMP_BILINEAR32_X4', '0')
        b = model(x, 2, False)
        for k in a:
            assert torch.equal(a[k], b[k]), (prec, k)


@pytest.mark.parametrize('arch,ncls', [('pdl', 1), ('bifpn', 4)])
def test_vector_point_sampling_is_the_scalar_kernels_arithmetic(monkeypatch, arch, ncls):
    """Round 6 (late): PointRend's point sampling of the fp32 / fp16x3 graph with 16 lanes per point and four channels per lane
    (point_features32v_kernel) -- per channel the one-channel-per-lane kernel's fmaf chain over the four corners (EMP_PF32_VEC=0),
    so the refined semantic logits are bit-identical, for one and four classes, in both precisions"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg, P, _ = _models(arch, ncls)
    for prec, size in (('fp32', 256), ('fp16x3', 384)):
        model = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
        x = torch.from_numpy(normalize(synth.em_tiles(2, size, seed=31), 0.57571, 0.12765))[:, None].cuda()
        monkeypatch.setenv('EMP_PF32_VEC', '1')
        a = {k: v.clone() for k, v in model(x, 3, False).items()}
        monkeypatch.setenv('EMP_PF32_VEC', '0')
        b = model(x, 3, False)
        assert float(a['sem_logits'].abs().max()) > 0
        for k in a:
            assert torch.equal(a[k], b[k]), (prec, k)
