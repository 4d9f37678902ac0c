"""Oracle (torch-CPU restatement) vs golden vectors produced by the imported
reference (oracle/gen_golden.py -> tests/golden/pdl_forward.npz)."""
import os

import numpy as np
import pytest
import torch

from empanada_napari_amd import weights
from empanada_napari_amd.preprocess import normalize
from oracle import pdl_model


@pytest.fixture(scope='module')
def folded():
    cfg = dict(weights.MITONET_PDL_CFG)
    return cfg, weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)


@pytest.mark.parametrize('case', ['a', 'b', 'c', 'd'])
def test_forward_matches_reference(golden_dir, folded, case):
    g = np.load(os.path.join(golden_dir, 'pdl_forward.npz'))
    cfg, P = folded
    x = torch.from_numpy(normalize(g[f'{case}_image'], 0.57571, 0.12765))[:, None]
    out = pdl_model.pdl_forward(P, x, cfg, int(g[f'{case}_render_steps']), bool(g[f'{case}_interpolate_ins']))
    for name in ('sem_logits', 'ctr_hmp', 'offsets'):
        ref = g[f'{case}_{name}']
        got = out[name].numpy()
        assert got.shape == ref.shape
        # tolerance: BN folding re-associates fp32 products (fold vs unfused BN)
        np.testing.assert_allclose(got, ref, atol=2e-4, rtol=2e-4, err_msg=f'{case}/{name}')
    # the synthetic weights give heads a useful dynamic range (not ~0 everywhere)
    assert np.abs(g[f'{case}_sem_logits']).max() > 0.5


def test_fold_accepts_fused_layout(golden_dir, folded):
    """fold_state_dict must understand the exported (fuse_model) key layout."""
    g = np.load(os.path.join(golden_dir, 'pdl_forward.npz'))
    cfg, P = folded
    keys = set(str(k) for k in g['fused_keys'])
    # build a fused-layout dict from the folded params and fold it again
    fsd = {}
    for L in weights.pdl_spec(cfg):
        n = L['name']
        w, b = P[n]
        if n + '.0.weight' in keys:
            fsd[n + '.0.weight'] = w
            if n + '.0.bias' in keys:
                fsd[n + '.0.bias'] = b
        elif n + '.weight' in keys:
            fsd[n + '.weight'] = w
            if n + '.bias' in keys:
                fsd[n + '.bias'] = b
        else:
            raise AssertionError(n)
    # un-fused BN leftovers (separable conv blocks keep their BatchNorm)
    leftover = [k for k in keys if k.endswith('running_mean')]
    assert leftover, 'exported layout keeps BN after separable convs'
    for k in leftover:
        base = k[:-len('.running_mean')]
        c = [L for L in weights.pdl_spec(cfg) if L['bn'] == base][0]['shape'][0]
        fsd[base + '.weight'] = np.ones(c, np.float32)
        fsd[base + '.bias'] = np.zeros(c, np.float32)
        fsd[base + '.running_mean'] = np.zeros(c, np.float32)
        fsd[base + '.running_var'] = np.ones(c, np.float32) - np.float32(weights.BN_EPS)
    Q = weights.fold_state_dict(fsd, cfg)
    for n in P:
        np.testing.assert_allclose(Q[n][0], P[n][0], atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize('tag,ncls', [('m1', 1), ('m4', 4)])
@pytest.mark.parametrize('case', ['a', 'b'])
def test_bifpn_forward_matches_reference(golden_dir, tag, ncls, case):
    """PanopticBiFPNPR (MitoNet_v1_mini-class, and the 4-class variant of BASELINE configs[4])."""
    g = np.load(os.path.join(golden_dir, 'bifpn_forward.npz'))
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    x = torch.from_numpy(normalize(g[f'{tag}{case}_image'], 0.57571, 0.12765))[:, None]
    out = pdl_model.model_forward(P, x, cfg, int(g[f'{tag}{case}_render_steps']), bool(g[f'{tag}{case}_interpolate_ins']))
    for name in ('sem_logits', 'ctr_hmp', 'offsets'):
        ref = g[f'{tag}{case}_{name}']
        got = out[name].numpy()
        assert got.shape == ref.shape
        tol = 5e-4 * max(1.0, float(np.abs(ref).max()))
        bad = np.abs(got - ref) > tol
        # multi-class PointRend: a top-2 tie in the uncertainty ranking may pick a different cell
        assert bad.mean() < (2e-3 if name == 'sem_logits' else 1e-12), (name, bad.mean(), np.abs(got - ref).max())


def test_fp16_format_emulation_brackets_the_fp32_forward():
    """oracle.pdl_model.Fp16Emu (the checker of tests/test_gpu_parity_fullsize.py): with every rounding site off it
    is the fp32 forward (up to the re-associated ASPP projection); with the engine's sites on it moves the heads by
    the fp16-format amount (1e-4 .. 1e-2), never more."""
    import torch
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    x = torch.from_numpy(normalize(synth.em_tiles(1, 64, seed=4), 0.57571, 0.12765))[:, None]
    ref = pdl_model.pdl_forward(P, x, cfg, 2, False)
    off = pdl_model.pdl_forward(P, x, cfg, 2, False, emu=pdl_model.Fp16Emu(False, False))
    emu = pdl_model.Fp16Emu()
    on = pdl_model.pdl_forward(P, x, cfg, 2, False, emu=emu)
    assert pdl_model._EMU is None
    assert float((off['ctr_hmp'] - ref['ctr_hmp']).abs().max()) < 2e-5
    d = float((on['ctr_hmp'] - ref['ctr_hmp']).abs().max())
    assert 1e-4 < d < 1e-2, d
    # every conv of the spec except the fp32-kept ones is a weight site; every block output an activation site
    assert 'encoder.layer3.2.conv2' in emu.sites_w and 'encoder.conv1' not in emu.sites_w
    assert 'semantic_head.head.1' not in emu.sites_w and 'encoder.layer4.2' in emu.sites_a
    split = pdl_model.pdl_forward(P, x, cfg, 2, False, emu=pdl_model.Fp16Emu(True, False, split_weights=emu.sites_w))
    assert float((split['ctr_hmp'] - ref['ctr_hmp']).abs().max()) < 2e-5     # hi + lo pairs carry 22 bits
