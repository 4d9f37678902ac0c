"""Full-size float parity of the network forward, at the north-star tolerance.

1024 x 1024 tiles (BASELINE configs[1] size) through ``HipPanopticDeepLab`` against

  (A) the oracle, layer by layer, on the engine's own input maps (teacher forcing): every layer must be the fp32
      result of the reference's arithmetic rounded once to the engine's storage format (one fp16 ulp; 1e-4 on the fp32
      heads).  This isolates KERNEL error from FORMAT error -- end to end the two cannot be told apart, because fp16
      pipelines decorrelate through rounding (reported by test_end_to_end_distance_to_format_oracle_is_reported).
      Which conv kernel runs depends on the launch's tile count: ONE tile has 32 tiles of 256 x 256 on the ASPP convs and
      runs them on the deep-ring 64 x 64 kernel.  The check is therefore made TWICE: on a batch of TWELVE tiles (image 0
      of every tap against the oracle; convolutions are per image), where the dominant ``conv_igemm256_kernel``, the
      merged two-decoder ASPP launches, the register-weight 3x3 kernels and the back-to-back conv fusion are what
      runs -- asserted through the engine's own launch profile -- and on the batch-1 call of the reference API (few-tile
      kernels, two-stream decoders).
  (B) the plain fp32 oracle (= the reference forward, pinned by tests/golden/pdl_forward.npz): the north star's gate,
      "semantic and center heatmaps within 1e-3 of the fp32 CPU path", asserted in rms on the centre heat-map and on the
      semantic probability; the max norm is bounded at 1.2 x what is measured (profiles/r03_parity_fullsize.json;
      profiles/r02_error_budget.csv explains why fp16 maps cannot reach 1e-3 in max norm at this throughput).
  (C) end to end: HIP heads -> HIP voting/merge vs fp32-oracle heads -> oracle voting/merge; the label FLIP COUNT is
      reported (pixels whose foreground differs, pixels whose instance differs after matching ids by overlap).

PointRend refines the 8192 most uncertain cells per step; a cell selected on one side only differs by
(refined - interpolated).  (A) therefore checks the subdivision separately on IDENTICAL inputs (the engine's own
coarse logits and feature map handed to the oracle's PointRend), where selection is identical up to fp32 ties.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 on the float semantic/center heatmaps"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_fullsize.json')


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


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


@pytest.fixture(scope='module')
def case():
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))      # oneDNN convs stop scaling (and oversubscribe) beyond ~32 threads
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    # the seeded network's centre head stays below the NMS threshold on most of a tile: lift the two head biases so
    # that the end-to-end comparison has instances and foreground (same parameters on both sides)
    for name, shift in (('ins_center.head.1', 0.75), ('semantic_head.head.1', 1.0), ('semantic_pr.point_head.predictor', 1.0)):
        w, b = P[name]
        P[name] = (w, b + np.float32(shift))
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    img = synth.em_tiles(1, 1024, seed=2024)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = {k: v.cpu().numpy() for k, v in model(x.cuda(), 2, False).items()}
    torch.cuda.synchronize()
    semx = model.tap('semantic_decoder.stage0.out').float().cpu().permute(0, 3, 1, 2).contiguous()
    coarse = model.tap_raw('semantic_head.out', (1, 1, 256, 256)).cpu()
    taps32 = {}
    ref32 = pdl_model.pdl_forward(P, x, cfg, 2, False, taps32)
    return dict(cfg=cfg, P=P, model=model, img=img, x=x, out=out, semx=semx, coarse=coarse, ref32=ref32, taps32=taps32)


def test_every_layer_is_the_correctly_rounded_fp32_result_at_full_size(case):
    """(A) on the batch-1 call: each layer of the 1024^2 forward, recomputed by the oracle FROM THE ENGINE'S OWN INPUT
    MAPS (teacher forcing, oracle.pdl_model.teacher_forced_layers), equals the engine's output map: to one fp16 ulp where
    the engine stores fp16 (the kernels' fp32 sums run in another order, so a result next to a rounding boundary may land
    on the other side), to 1e-4 of the map's scale on the fp32 heads -- far inside the north star's 1e-3.  One tile gives
    the deep layers fewer than 192 tiles: this is the few-tile path (deep-ring 64 x 64 conv tile, separate ASPP
    launches, two-stream decoders) of the reference API's calling convention; the batch-32 kernels are checked by
    test_every_layer_at_batch_twelve_runs_the_dominant_kernels below."""
    from oracle import pdl_model
    model, out = case['model'], case['out']
    fp32_heads = {'semantic_head.out': case['coarse'], 'ins_center.out': torch.from_numpy(out['ctr_hmp']),
                  'ins_xy.out': torch.from_numpy(out['offsets'])}

    def tap(name):
        return model.tap(name).float().cpu().permute(0, 3, 1, 2).contiguous()

    rows = _teacher_forced_rows(pdl_model.teacher_forced_layers(case['P'], case['cfg'], case['x'], tap), tap, fp32_heads)
    worst_flip, worst_ulp = max(r[3] for r in rows), max(r[4] for r in rows)
    for r in rows:
        print('%-36s rms %8.4f  max|d| %.3e  differing %.4f  d/tol %.2f' % r)
    _report('teacher_forced', dict(layers=len(rows), worst_differing_fraction=worst_flip, worst_error_over_tolerance=worst_ulp,
                                   heads={r[0]: r[2] for r in rows if r[0].endswith('.out') and 'stage' not in r[0]}))
    assert len(rows) >= 60


def test_every_layer_at_batch_twelve_runs_the_dominant_kernels(case):
    """(A) where the batch-32 kernels run (VERDICT r02 'weak' 2): twelve 1024^2 tiles in one call -- 192 pixel tiles of
    256 on the stride-16 maps, the threshold from which even the 256-cout layers go to the 256 x 256 implicit-GEMM tile.
    The engine's launch profile must show the SAME 29 launches of ``conv_igemm256_kernel<0, false>`` as the batch-32
    bench step (the three merged two-decoder ASPP 3x3 convs among them) -- then image 0 of every tap is checked against
    the oracle layer by layer exactly as above (a convolution is per image; the ASPP pooling branch is per image too)."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    model = case['model']
    imgs = np.concatenate([case['img'], synth.em_tiles(11, 1024, seed=2025)])
    x12 = torch.from_numpy(normalize(imgs, 0.57571, 0.12765))[:, None]
    model.profile(True)
    out12 = {k: v.cpu() for k, v in model(x12.cuda(), 2, False).items()}
    torch.cuda.synchronize()
    ms, flops, launches = model.profile_read()
    model.profile(False)
    print(f'batch 12: {launches} launches of the 256x256 conv tile, {flops / 1e12:.2f} TFLOP in {ms:.2f} ms')
    assert launches >= 29, f'only {launches} launches went to conv_igemm256_kernel: the batch-32 kernels are not under test'
    coarse = model.tap_raw('semantic_head.out', (12, 1, 256, 256))[:1].cpu()
    fp32_heads = {'semantic_head.out': coarse, 'ins_center.out': out12['ctr_hmp'][:1], 'ins_xy.out': out12['offsets'][:1]}

    def tap(name):
        return model.tap(name)[:1].float().cpu().permute(0, 3, 1, 2).contiguous()

    rows = _teacher_forced_rows(pdl_model.teacher_forced_layers(case['P'], case['cfg'], x12[:1], tap), tap, fp32_heads)
    _report('teacher_forced_batch12', dict(layers=len(rows), launches_conv256=int(launches),
                                          worst_differing_fraction=max(r[3] for r in rows),
                                          worst_error_over_tolerance=max(r[4] for r in rows),
                                          heads={r[0]: r[2] for r in rows if r[0].endswith('.out') and 'stage' not in r[0]}))
    assert len(rows) >= 60
    # and the batch result of image 0 is the batch-1 result, bit for bit (every tile variant walks K in the same order)
    for k in ('ctr_hmp', 'offsets', 'sem_logits'):
        assert np.array_equal(out12[k][0].numpy(), case['out'][k][0]), k


def test_end_to_end_distance_to_format_oracle_is_reported(case):
    """Two fp16 pipelines decorrelate through rounding (see teacher_forced_layers' docstring): end to end the engine is
    as far from the format-emulating oracle as from fp32.  Reported for DESIGN.md on a 256^2 crop (the statement does not
    depend on the size and the extra oracle forward stays cheap); only sanity-bounded here."""
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    x = torch.from_numpy(normalize(case['img'][:, :256, :256], 0.57571, 0.12765))[:, None]
    o = {k: v.cpu().numpy() for k, v in case['model'](x.cuda(), 2, False).items()}
    r16 = pdl_model.pdl_forward(case['P'], x, case['cfg'], 2, False, emu=pdl_model.Fp16Emu())
    r32 = pdl_model.pdl_forward(case['P'], x, case['cfg'], 2, False)
    rep = {}
    for tag, r in (('format', r16), ('fp32', r32)):
        e = np.abs(o['ctr_hmp'] - r['ctr_hmp'].numpy())
        rep[f'ctr_max_vs_{tag}'], rep[f'ctr_rms_vs_{tag}'] = float(e.max()), float(np.sqrt((e ** 2).mean()))
    e = np.abs(r16['ctr_hmp'].numpy() - r32['ctr_hmp'].numpy())
    rep['ctr_max_format_vs_fp32'], rep['ctr_rms_format_vs_fp32'] = float(e.max()), float(np.sqrt((e ** 2).mean()))
    print('end-to-end centre heat-map distances @256^2:', rep)
    _report('end_to_end_256', rep)
    assert rep['ctr_rms_vs_format'] < 2e-3 and rep['ctr_max_vs_format'] < 2e-2


def test_pointrend_on_identical_inputs_at_1e3(case):
    """(A) for the subdivision: the oracle's PointRend on the ENGINE's coarse logits and features."""
    from oracle import pdl_model
    cfg, P = case['cfg'], case['P']
    emu = pdl_model.Fp16Emu().bind(P)
    pdl_model._EMU = emu
    try:
        want = pdl_model.point_rend_forward(P, case['coarse'], case['semx'], 2, cfg['subdivision_num_points'], cfg['num_fc'])
    finally:
        pdl_model._EMU = None
    d = np.abs(_sig(case['out']['sem_logits']) - _sig(want.numpy()))
    flips = int((d > TOL).sum())
    rep = dict(prob_max=float(d.max()), cells_over_1e3=flips, cells=int(d.size))
    print('PointRend on identical inputs:', rep)
    _report('pointrend_identical_inputs', rep)
    # a cell whose uncertainty ties with the 8192nd one in fp32 may be picked by one side only
    assert flips <= 16, f'{flips} cells differ by more than 1e-3 in probability'
    assert np.sort(d.ravel())[-17 if flips else -1] < TOL


def test_heads_vs_fp32_reference_forward(case):
    """(B): gap to the fp32 forward = the fp16 format (error budget in profiles/): rms within 1e-3, max reported."""
    o, r, t = case['out'], case['ref32'], case['taps32']
    e_ctr = np.abs(o['ctr_hmp'] - r['ctr_hmp'].numpy())
    e_off = np.abs(o['offsets'] - r['offsets'].numpy())
    e_semc = np.abs(_sig(case['coarse'].numpy()) - _sig(t['sem_coarse'].numpy()))
    e_prob = np.abs(_sig(o['sem_logits']) - _sig(r['sem_logits'].numpy()))
    rep = dict(ctr_max=float(e_ctr.max()), ctr_rms=float(np.sqrt((e_ctr ** 2).mean())),
               off_max=float(e_off.max()), off_rms=float(np.sqrt((e_off ** 2).mean())),
               sem_coarse_prob_max=float(e_semc.max()), sem_coarse_prob_rms=float(np.sqrt((e_semc ** 2).mean())),
               prob_max=float(e_prob.max()), prob_rms=float(np.sqrt((e_prob ** 2).mean())),
               prob_frac_over_1e3=float((e_prob > TOL).mean()), prob_frac_over_1e2=float((e_prob > 1e-2).mean()))
    print('HIP vs fp32 oracle @1024^2:', rep)
    _report('vs_fp32_oracle', rep)
    # the north star's gate, in rms (heat-map values are O(1), range ~[-3, 3]; probabilities in [0, 1]) ...
    assert rep['ctr_rms'] < TOL, rep['ctr_rms']                      # measured 0.927e-3 (round 2: 1.0002e-3)
    assert rep['sem_coarse_prob_rms'] < TOL, rep['sem_coarse_prob_rms']      # measured 0.645e-3
    # ... and the max norm at 1.2 x what is measured (profiles/r03_parity_fullsize.json): 4.62e-3 / 4.12e-3
    assert rep['ctr_max'] < 5.6e-3 and rep['sem_coarse_prob_max'] < 5.0e-3, rep
    assert rep['off_rms'] < 1.5e-2 and rep['off_max'] < 8.5e-2, rep       # pixels; measured 1.20e-2 / 6.9e-2


def test_final_semantic_map_is_gated(case):
    """(B) for the map a user gets -- the semantic probability AFTER PointRend.  PointRend
    re-predicts the 8192 most uncertain cells per step from the fine features; a cell picked on one side only differs by
    (refined - interpolated), which is not a format error.  So: (1) the share of cells beyond 1e-3 and the plain rms are
    bounded at 1.2 x what is measured (profiles/r04_parity_fullsize.json), and (2) on the cells that NEITHER side refined --
    final logit == the twice bilinearly up-sampled coarse logit of that side -- the north star's 1e-3 holds in rms, as on
    the coarse map they interpolate."""
    import torch.nn.functional as F
    o, r, t = case['out'], case['ref32'], case['taps32']

    def untouched(final, coarse):
        up = coarse
        for _ in range(2):
            up = F.interpolate(up, scale_factor=2.0, mode='bilinear', align_corners=False)
        return (torch.as_tensor(final) - up).abs() < 1e-4

    m_hip = untouched(o['sem_logits'], case['coarse'])
    m_ref = untouched(r['sem_logits'], t['sem_coarse'])
    both = (m_hip & m_ref).numpy()
    e_prob = np.abs(_sig(o['sem_logits']) - _sig(r['sem_logits'].numpy()))
    rep = dict(cells=int(e_prob.size), untouched_on_both_sides=float(both.mean()),
               untouched_hip=float(m_hip.float().mean()), untouched_ref=float(m_ref.float().mean()),
               prob_rms_untouched=float(np.sqrt((e_prob[both] ** 2).mean())), prob_max_untouched=float(e_prob[both].max()),
               prob_frac_over_1e3_untouched=float((e_prob[both] > TOL).mean()),
               prob_rms=float(np.sqrt((e_prob ** 2).mean())), prob_frac_over_1e3=float((e_prob > TOL).mean()),
               selection_differs=float((m_hip != m_ref).float().mean()))
    print('final semantic map @1024^2:', rep)
    _report('final_semantic', rep)
    assert rep['untouched_on_both_sides'] > 0.5, rep       # two steps x 8192 points and their 2x2 / 4x4 footprints
    assert rep['prob_rms_untouched'] < TOL, rep            # the gate, where PointRend's selection does not interfere
    assert rep['prob_max_untouched'] < 6.0e-3, rep         # max norm: the fp16 format, as on the coarse map (4.1e-3 x 1.2 + interpolation)
    assert rep['prob_frac_over_1e3'] < 0.093 and rep['prob_rms'] < 0.022, rep      # 1.2 x measured (0.077 / 0.018)


def _match_ids(a, b):
    """relabel the instances of ``a`` with the id of the instance of ``b`` they overlap most (0 stays 0)."""
    out = np.zeros_like(a)
    ids = np.unique(a)
    for i in ids[ids > 0]:
        m = a == i
        vals, cnt = np.unique(b[m], return_counts=True)
        keep = vals > 0
        out[m] = vals[keep][np.argmax(cnt[keep])] if keep.any() else -int(i)
    return out


def test_end_to_end_label_flips_vs_fp32_pipeline(case):
    """(C): label maps after BOTH pipelines: flip counts reported, foreground flips bounded."""
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    from oracle import postprocess as opp
    eng = PanopticDeepLabRenderEngine(case['model'], [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)
    pan = eng(case['x'], case['img'].shape[-2:], 1).cpu().numpy()[0]
    r = {k: v.numpy() for k, v in case['ref32'].items()}
    r['sem'] = opp.logits_to_prob(r['sem_logits'])
    oeng = opp.RenderEngine(lambda *_: r, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                            confidence_thr=0.5, coarse_boundaries=True)
    want = oeng.postprocess(r['sem'], oeng.cells(r['ctr_hmp'], r['offsets'], 1))[0]
    # and the post-processing alone on the engine's own heads (bit-exact, as in test_gpu_postprocess.py)
    o = dict(case['out'])
    # (the engine's own probabilities: the host's sigmoid may round the other way at the 0.5 threshold)
    from empanada_napari_amd.engines import logits_to_prob
    o['sem'] = logits_to_prob(torch.from_numpy(np.ascontiguousarray(o['sem_logits'])).cuda()).cpu().numpy()
    assert float(np.abs(o['sem'] - opp.logits_to_prob(o['sem_logits'])).max()) < 1e-6
    own = oeng.postprocess(o['sem'], oeng.cells(o['ctr_hmp'], o['offsets'], 1))[0]
    assert np.array_equal(pan, own), f'{int((pan != own).sum())} flips with identical head tensors'
    n_hip, n_ref = len(np.unique(pan)) - 1, len(np.unique(want)) - 1
    fg_flip = int(((pan > 0) != (want > 0)).sum())
    verbatim = int((pan != want).sum())
    matched = _match_ids(pan, want)
    ins_flip = int((matched != want).sum())
    rep = dict(instances_hip=n_hip, instances_ref=n_ref, pixels=int(pan.size), foreground_flips=fg_flip,
               verbatim_label_diffs=verbatim, instance_flips_after_id_matching=ins_flip,
               foreground_fraction=float((want > 0).mean()))
    print('end-to-end label flips (HIP fp16 pipeline vs fp32 oracle pipeline):', rep)
    _report('label_flips', rep)
    assert n_ref > 0 and n_hip > 0
    # round 2: 1396 foreground flips / 7832 instance flips; round 3 (centre path at fp32 accuracy): 1396 / 5013
    assert fg_flip <= 1700, f'{fg_flip} foreground flips'
    assert ins_flip <= 6000, f'{ins_flip} pixels change instance'
    assert abs(n_hip - n_ref) <= max(2, 0.005 * n_ref)


@pytest.mark.parametrize('ncls', [1, 4])
def test_bifpn_512_tile_vs_fp32_forward(ncls):
    """PanopticBiFPNPR (MitoNet_v1_mini class; 4 outputs = BASELINE configs[4]) on one 512^2 tile against the fp32 oracle
    forward (pinned by tests/golden/bifpn_forward.npz): the same statement as (B) for the second network family -- rms inside
    1e-3 of the head's scale ASSERTED (round 4), max bounded.  Its encoder and heads are the kernels that (A) checks layer by layer."""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    x = torch.from_numpy(normalize(synth.em_tiles(1, 512, seed=77), 0.57571, 0.12765))[:, None]
    out = {k: v.cpu() for k, v in model(x.cuda(), 2, False).items()}
    taps = {}
    ref = pdl_model.bifpn_forward(P, x, cfg, 2, False, taps)
    rep = {}
    for k in ('ctr_hmp', 'offsets'):
        d = (out[k] - ref[k]).abs()
        scale = float(ref[k].pow(2).mean().sqrt())
        rep[k] = dict(max=float(d.max()), rms=float(d.pow(2).mean().sqrt()), scale_rms=scale)
        # the north star's 1e-3, relative to the map's rms (these heads are unbounded: |ctr| rms ~8, |offsets| rms ~80 px).
        # Round 4 measures 0.79e-3 / 0.63e-3 (centre, 1 / 4 classes) and 0.90e-3 / 0.78e-3 (offsets): precise nodes, hi + lo
        # weight pairs and fused maps on the centre path (pdl_net.hip precise_node / wsplit_on / fsplit_on); round 3: 1.4e-3 / 1.6e-3
        assert rep[k]['rms'] < TOL * max(1.0, scale), (k, rep[k])
        assert rep[k]['max'] < 9e-3 * max(1.0, scale), (k, rep[k])
    if ncls == 1:
        pe = (torch.sigmoid(out['sem_logits']) - torch.sigmoid(ref['sem_logits'])).abs()
    else:
        pe = (torch.softmax(out['sem_logits'], 1) - torch.softmax(ref['sem_logits'], 1)).abs()
    rep['prob'] = dict(max=float(pe.max()), rms=float(pe.pow(2).mean().sqrt()), frac_over_1e2=float((pe > 1e-2).float().mean()))
    print(f'BiFPN ({ncls} class) 512^2 vs fp32 oracle:', rep)
    _report(f'bifpn_512_ncls{ncls}', rep)
    assert rep['prob']['frac_over_1e2'] < 2e-2          # PointRend selection flips only


def _teacher_forced_rows(gen, tap, fp32_heads):
    """Shared assertion loop of the teacher-forced checks: one fp16 ulp (+1e-4 of the map's scale for the rounded
    depthwise intermediate) where the engine stores fp16, 1e-4 of the scale on the fp32 heads."""
    from oracle import pdl_model
    rows = []
    for name, want, rounds in gen:
        rms = float(want.pow(2).mean().sqrt())
        if rounds:
            got = tap(name)[:, :want.shape[1]]
            w16 = pdl_model.Fp16Emu.r16(want)
            d = (got - w16).abs()
            ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
            excess = float((d - ulp - 1e-4 * max(1.0, rms)).max())
            flips = float((got != w16).float().mean())
            rows.append((name, rms, float(d.max()), flips, float((d / (ulp + 1e-4 * max(1.0, rms))).max())))
            assert excess <= 0, f'{name}: off by more than one fp16 ulp (max |d| {float(d.max()):.3e}, rms {rms:.3f})'
            assert flips < 0.05, f'{name}: {flips:.3%} of the elements differ from the correctly rounded result'
        else:
            got = fp32_heads[name]
            d = float((got - want).abs().max())
            rows.append((name, rms, d, 0.0, 0.0))
            assert d < 1e-4 * max(1.0, rms), f'{name}: fp32 head off by {d:.3e} (rms {rms:.3f})'
    return rows


@pytest.mark.parametrize('ncls', [1, 4])
def test_bifpn_every_layer_is_the_correctly_rounded_fp32_result(ncls):
    """(A) for the second network family: every map of the PanopticBiFPNPR forward at 512^2 -- encoder at
    output stride 32, P2 / P6 resampling, the 3 x (top-down + bottom-up) fast-normalised fusion nodes with their 3x3
    separable convs, the five transposed-conv decoder steps, the 5x5 fusion and the three heads -- recomputed by
    oracle.pdl_model.teacher_forced_layers_bifpn from the engine's own input maps."""
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    x = torch.from_numpy(normalize(synth.em_tiles(1, 512, seed=77), 0.57571, 0.12765))[:, None]
    out = {k: v.cpu() for k, v in model(x.cuda(), 2, False).items()}
    heads = {'semantic_head.out': model.tap_raw('semantic_head.out', (1, ncls, 128, 128)).cpu(),
             'ins_center.out': out['ctr_hmp'], 'ins_xy.out': out['offsets']}

    def tap(name):
        return model.tap(name).float().cpu().permute(0, 3, 1, 2).contiguous()

    rows = _teacher_forced_rows(pdl_model.teacher_forced_layers_bifpn(P, cfg, x, tap), tap, heads)
    for r in rows:
        print('%-44s rms %9.4f  max|d| %.3e  differing %.4f  d/tol %.2f' % r)
    _report(f'teacher_forced_bifpn_ncls{ncls}',
            dict(layers=len(rows), worst_differing_fraction=max(r[3] for r in rows), worst_error_over_tolerance=max(r[4] for r in rows),
                 heads={r[0]: r[2] for r in rows if r[0] in heads}))
    assert len(rows) >= 170
