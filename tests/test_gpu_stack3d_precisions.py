"""End-to-end 3-D label parity by precision (round 6; VERDICT r05 missing 3 / item 1c).  BASELINE configs[2]: the 512^3
ortho-plane job (Engine3d.infer_on_axis x 3 + tracker_consensus; empanada_napari/inference.py:111-169, 491-578) in all three
precisions of the forward.  Round 5 reported 12 consensus objects on the fp16 engine and 11 in the fp16x3 mode without saying
which one is the reference's answer.  The fp32 mode (heads within 1e-4 of the oracle's fp32 forward,
tests/test_gpu_fp32_mode.py) is the on-device comparator: every object of every precision is matched to the fp32 mode's objects by
voxel overlap, and the report (gpurun_out/stack3d_precisions.json -> profiles/r06_stack3d_precisions.json) names the objects that
exist on one side only."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, 'gpurun_out', 'stack3d_precisions.json')
DIV = 10000


def _job(precision, vol):
    import torch
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)      # bench.py's network
    model = HipPanopticDeepLab(P, cfg, folded=True, precision=precision)
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    e3 = Engine3d(mc, label_divisor=DIV, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500,
                  min_extent=5)

    def run():
        trs = {name: e3.infer_on_axis(vol, name)[1] for name in ('xy', 'xz', 'yz')}
        return list(tracker_consensus(trs, None, mc, label_divisor=DIV, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False,
                                      min_size=500, min_extent=5, dtype=np.uint32, chunk_size=(256, 256, 256)))[0]
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cvol, _, inst = run()
    dt = time.perf_counter() - t0
    cvol = np.asarray(cvol)
    del e3, model
    torch.cuda.empty_cache()
    return cvol, inst, dt


def _objects(v):
    ids, cnt = np.unique(v, return_counts=True)
    return {int(i): int(c) for i, c in zip(ids, cnt) if i > 0}


def _match(a, b):
    """every object of ``a`` -> (best-overlapping object of ``b``, IoU); objects of b nobody maps to are 'missing'"""
    oa, ob = _objects(a), _objects(b)
    pair = a.astype(np.uint64) * np.uint64(1 << 32) | b.astype(np.uint64)
    keys, cnt = np.unique(pair[(a > 0) & (b > 0)], return_counts=True)
    best = {}
    for k, c in zip(keys, cnt):
        ia, ib = int(k >> np.uint64(32)), int(k & np.uint64(0xffffffff))
        iou = c / float(oa[ia] + ob[ib] - c)
        if ia not in best or iou > best[ia][1]:
            best[ia] = (ib, iou)
    rows = [{'id': ia, 'voxels': oa[ia], 'match': best.get(ia, (0, 0.0))[0], 'iou': round(best.get(ia, (0, 0.0))[1], 5)} for ia in sorted(oa)]
    hit = {r['match'] for r in rows if r['iou'] >= 0.5}
    missing = [{'id': ib, 'voxels': ob[ib]} for ib in sorted(ob) if ib not in hit]
    return rows, missing


def test_stack3d_label_volumes_by_precision():
    from empanada_napari_amd import synth
    vol = synth.blob_volume(512, 512, 512, seed=0, n_blobs=256, fast=True)      # bench.py's stack3d volume
    out = {}
    for prec in ('fp32', 'fp16x3', 'fp16'):
        out[prec] = _job(prec, vol)
    ref = out['fp32'][0]
    rep = {'volume': [512] * 3, 'comparator': "precision='fp32' (heads within 1e-4 of the oracle's fp32 forward)"}
    for prec in ('fp32', 'fp16x3', 'fp16'):
        cvol, inst, dt = out[prec]
        rows, missing = _match(cvol, ref)
        extra = [r for r in rows if r['iou'] < 0.5]
        rep[prec] = {
            'seconds': round(dt, 3), 'Mvoxel_per_s': round(vol.size / dt / 1e6, 1), 'consensus_objects': len(inst),
            'foreground_voxels': int((cvol > 0).sum()),
            'foreground_voxels_differing_from_fp32': int(((cvol > 0) != (ref > 0)).sum()),
            'objects_without_an_fp32_counterpart_iou50': extra, 'fp32_objects_without_a_counterpart_iou50': missing,
            'min_iou_of_matched_objects': min([r['iou'] for r in rows if r['iou'] >= 0.5] or [0.0]),
        }
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    json.dump(rep, open(REPORT, 'w'), indent=1)
    print(json.dumps(rep, indent=1))
    n32 = rep['fp32']['consensus_objects']
    assert n32 >= 8
    # the default mode reproduces the fp32 mode's objects: same count, every object matched, voxel flips at the level of
    # fp32 near-ties of a probability at its threshold (the forwards differ by ~2e-5)
    x3 = rep['fp16x3']
    assert x3['consensus_objects'] == n32 and not x3['objects_without_an_fp32_counterpart_iou50'] and not x3['fp32_objects_without_a_counterpart_iou50'], x3
    assert x3['foreground_voxels_differing_from_fp32'] <= 2e-4 * rep['fp32']['foreground_voxels'], x3
    assert x3['min_iou_of_matched_objects'] > 0.99, x3
    # the fp16 engine (throughput opt-in): REPORTED, hardly bounded.  On this volume the seeded (untrained) network's semantic
    # probability sits within a few 1e-3 of its threshold over wide areas, so the engine's ~5e-3 deviations move whole regions
    # across it: round 6 measured 12 objects against the fp32 mode's 11, 8 of them without an fp32 counterpart at IoU 0.5 and
    # more differing foreground voxels than there is foreground -- the object COUNT of round 5's report (12 vs 11) understated
    # the difference; the reference's answer is the fp32 / fp16x3 one
    h = rep['fp16']
    assert abs(h['consensus_objects'] - n32) <= 3, h
