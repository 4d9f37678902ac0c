"""RegNet encoders (reference: empanada/models/encoders/regnet.py:38-316, the two the reference can
export: quantization/encoders/__init__.py) -- host side: the layout derived from the generating parameters, the layer
spec in the reference's key layout, the architecture read back from an export's keys + shapes, and the oracle's restated
forward against outputs of the imported reference (tests/golden/regnet_forward.npz, oracle/gen_golden.py::gen_regnet):
PanopticBiFPNPR on regnety_6p4gf (grouped 3x3 + the reference's per-pixel squeeze-excite gate) and PanopticDeepLabPR on
regnetx_6p4gf (whose stage 4 stays at stride 2 whatever ``stage4_stride`` says)."""
import os

import numpy as np
import pytest
import torch

from empanada_napari_amd import weights
from empanada_napari_amd.preprocess import normalize
from oracle import pdl_model

MODELS = {
    'y': (dict(weights.MITONET_MINI_CFG, encoder='regnety_6p4gf', num_classes=2), 11),
    'x': (dict(weights.MITONET_PDL_CFG, encoder='regnetx_6p4gf'), 12),
}


def regnet_model(tag):
    cfg, seed = MODELS[tag]
    cfg = dict(cfg, regnet=weights.regnet_cfg(cfg))
    return cfg, weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=seed), cfg)


def test_layouts_of_the_exportable_regnets():
    """regnet.py:262-271,296-301 through RegNetConfig (:168-260): widths / depths / groups as the reference derives them"""
    x = weights.regnet_layout(*weights.REGNET_PARAMS['regnetx_6p4gf'])
    y = weights.regnet_layout(*weights.REGNET_PARAMS['regnety_6p4gf'])
    assert (x['widths'], x['depths'], x['groups'], x['use_se']) == ([168, 392, 784, 1624], [2, 4, 10, 1], [3, 7, 14, 29], False)
    assert (y['widths'], y['depths'], y['groups'], y['use_se']) == ([144, 288, 576, 1296], [2, 7, 14, 2], [2, 4, 8, 18], True)
    assert x['w_stem'] == y['w_stem'] == 32
    assert weights.encoder_widths({'encoder': 'regnety_6p4gf'}) == [144, 288, 576, 1296]
    assert weights.encoder_widths({'encoder': 'resnet50'}) == [256, 512, 1024, 2048]
    with pytest.raises(NotImplementedError):
        weights.regnet_cfg({'encoder': 'regnety_16gf'})


@pytest.mark.parametrize('tag', ['y', 'x'])
def test_architecture_is_read_back_from_the_exports_keys(golden_dir, tag):
    """an export carries no architecture (empanada_napari/configs/*.yaml): widths, depths, group width and the gate come
    from the fused state dict's shapes -- here the reference's own fused key layout with the shapes behind it"""
    g = np.load(os.path.join(golden_dir, 'regnet_forward.npz'))
    sd = {str(k): np.empty(tuple(int(d) for d in str(s).split(',')) if str(s) else (), np.float32)
          for k, s in zip(g[f'{tag}_fused_keys'], g[f'{tag}_fused_shapes'])}
    cfg = weights.infer_cfg(sd)
    want, _ = MODELS[tag]
    assert cfg['encoder'] == want['encoder'] and cfg['arch'] == want['arch']
    assert cfg['regnet'] == weights.regnet_cfg(want) and cfg['num_classes'] == want['num_classes']
    spec = weights.model_spec(cfg)
    names = {L['name'] for L in spec}
    assert 'encoder.stem.cbr.0' in names and 'encoder.stage4.block1.downsample.conv.0' in names
    assert ('encoder.stage1.block1.bottleneck.se.se.2' in names) == (tag == 'y')
    assert 'encoder.stage1.block2.downsample.conv.0' not in names          # same width, stride 1: identity shortcut
    for L in spec:      # every layer of the spec is in the export, under the plain or the fused (".0") key, with its shape
        if L['kind'] == 'fw':
            assert sd[L['name']].shape == L['shape']
            continue
        key = L['name'] + '.weight' if L['name'] + '.weight' in sd else L['name'] + '.0.weight'
        assert sd[key].shape == L['shape'], L['name']


@pytest.mark.parametrize('tag,case', [('y', 'a'), ('y', 'b'), ('x', 'a'), ('x', 'b')])
def test_oracle_forward_matches_the_reference(golden_dir, tag, case):
    g = np.load(os.path.join(golden_dir, 'regnet_forward.npz'))
    cfg, P = regnet_model(tag)
    x = torch.from_numpy(normalize(g[f'{tag}{case}_image'], 0.57571, 0.12765))[:, None]
    out = pdl_model.model_forward(P, x, cfg, int(g[f'{tag}{case}_render_steps']), bool(g[f'{tag}{case}_interpolate_ins']))
    for name in ('sem_logits', 'ctr_hmp', 'offsets'):
        ref = g[f'{tag}{case}_{name}']
        assert out[name].shape == ref.shape
        np.testing.assert_allclose(out[name].numpy(), ref, atol=3e-4, rtol=3e-4, err_msg=f'{tag}{case}/{name}')
    assert np.abs(g[f'{tag}{case}_sem_logits']).max() > 0.5


@pytest.mark.parametrize('tag', ['y', 'x'])
def test_oracle_pyramid_matches_the_reference(golden_dir, tag):
    """the encoder alone: stem + four stages, sampled -- pins the grouped convolutions, the gate and the strides level by
    level (a decoder could hide a wrong level behind its own normalisation)"""
    g = np.load(os.path.join(golden_dir, 'regnet_forward.npz'))
    cfg, P = regnet_model(tag)
    x = torch.from_numpy(normalize(g[f'{tag}a_image'], 0.57571, 0.12765))[:, None]
    pyr = pdl_model.regnet_forward(P, x, cfg['regnet'])
    assert len(pyr) == 5
    for i, f in enumerate(pyr):
        assert tuple(f.shape) == tuple(int(v) for v in g[f'{tag}_pyr{i}_shape'])
        assert abs(float(f.abs().mean()) - float(g[f'{tag}_pyr{i}_absmean'])) < 1e-4 * max(1.0, float(g[f'{tag}_pyr{i}_absmean']))
        np.testing.assert_allclose(f[:, ::7, ::3, ::3].numpy(), g[f'{tag}_pyr{i}_sample'], atol=2e-4, rtol=2e-4)
    assert float(pyr[4].abs().mean()) > 0.05          # the seeded net keeps its residual stream alive to the last stage


def test_regnet_has_no_fp16_emulation():
    cfg, P = regnet_model('x')
    with pytest.raises(NotImplementedError):
        pdl_model.model_forward(P, torch.zeros(1, 1, 64, 64), cfg, 2, False, emu=pdl_model.Fp16Emu())
