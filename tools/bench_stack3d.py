"""BASELINE configs[2]: ortho-plane 3-D inference + consensus on a synthetic cube, one MI355X.
    python tools/bench_stack3d.py [size=256] [batch=0] [--cpu]
Prints one JSON line: voxels/s over the whole job (3 axes + consensus + fill) and a time breakdown."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab  # noqa: E402
from empanada_napari_amd.inference import Engine3d, tracker_consensus  # noqa: E402


def cpu_baseline(cfg, P, vol, n_slices):
    """The reference's per-axis control flow on the host cores, restated with the oracle (3-D engine with recursive
    median, dense -> RLE, matcher, tracker) on the first ``n_slices`` xy slices: voxels per second of ONE axis pass.
    The full job is three such passes + consensus, so the job-level CPU rate is about a third of this figure."""
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model, postprocess as opp, sparse as osp
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def model(x, rs, interp):
        o = pdl_model.pdl_forward(P, torch.from_numpy(x), cfg, rs, interp)
        return {k: v.numpy() for k, v in o.items()}

    eng = opp.RenderEngine3d(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                             padding_factor=16, coarse_boundaries=True, median_kernel_size=3)
    t0 = time.perf_counter()
    pans = []
    for z in range(n_slices):
        r = eng(normalize(vol[z], 0.57571, 0.12765)[None, None], vol[z].shape, 1)
        if r is not None:
            pans.append(r[0])
    pans += [s[0] for s in eng.end(1)]
    m = osp.RLEMatcher(1, 10000, 0.25, 0.25)
    stack = [osp.apply_matchers(osp.pan_seg_to_rle_seg(p, [1], 10000, [1], force_connected=True), [m]) for p in pans]
    m.target_rle, m.assign_new = None, False
    tr = osp.InstanceTracker(1, 10000, (n_slices,) + vol.shape[1:], 'xy')
    for idx in range(len(pans) - 1, -1, -1):
        tr.update(osp.apply_matchers(stack[idx], [m])[1], idx)
    tr.finish()
    dt = time.perf_counter() - t0
    return {'value': round(n_slices * vol.shape[1] * vol.shape[2] / dt, 1), 'unit': 'voxels/s (one axis pass)', 'kind': 'port',
            'cores': torch.get_num_threads(), 'sample': f'{n_slices} xy slices of {vol.shape[1]}x{vol.shape[2]} ({dt:.1f} s)'}


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 0: the engine picks (about 16 Mpixel per batch)
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.blob_volume(size, size, size, seed=0, n_blobs=max(8, (size // 32) ** 2), fast=True)
    eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
                   min_size=500, min_extent=5, batch_size=batch or None)
    # warm-up: one untimed pass of the whole job (arena, first launches, the caching allocator's first hipMallocs for
    # every axis' block sizes -- ~50 ms on the xz axis otherwise)
    _w = {name: eng.infer_on_axis(vol, name)[1] for name in ('xy', 'xz', 'yz')}
    list(tracker_consensus(_w, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False,
                           min_size=500, min_extent=5, dtype=np.uint32))
    del _w
    torch.cuda.synchronize()
    # ---- the job as a user runs it: infer_on_axis x 3 + tracker_consensus (matching overlaps the GPU inside) ----
    j0 = time.perf_counter()
    job_trackers, job_axes = {}, {}
    for name in ('xy', 'xz', 'yz'):
        ja = time.perf_counter()
        _, job_trackers[name] = eng.infer_on_axis(vol, name)
        job_axes[name] = round(time.perf_counter() - ja, 3)
    jc = time.perf_counter()
    job_out = list(tracker_consensus(job_trackers, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                     allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32))
    job_s = time.perf_counter() - j0
    job = {'seconds': round(job_s, 3), 'axes_s': job_axes, 'consensus_fill_s': round(time.perf_counter() - jc, 3),
           'consensus_objects': len(job_out[0][2])}
    del job_out
    # ---- the same stages one after the other, for the breakdown ----
    t = {}
    t0 = time.perf_counter()
    trackers = {}
    for name, axis in (('xy', 0), ('xz', 1), ('yz', 2)):
        ta = time.perf_counter()
        pans = eng.predict_slices(vol, axis)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        # same as infer_on_axis from here (re-using the predicted slices keeps the breakdown honest)
        from empanada_napari_amd import sparse
        trs = eng.create_trackers(vol.shape, name)
        sms = {label: sparse.StackMatcher(label, eng.label_divisor, eng.merge_iou_thr, eng.merge_ioa_thr,
                                          match=label in eng.thing_list) for label in eng.labels}
        for i0 in range(0, len(pans), 64):
            chunk = torch.stack(pans[i0:i0 + 64])
            for label, (runs_list, off) in sparse.pan_stack_to_runs(chunk, eng.labels, eng.label_divisor, eng.thing_list,
                                                                   True).items():
                for runs in runs_list:
                    sms[label].push_runs(runs, chunk.shape[-1], off)
        tb2 = time.perf_counter()
        for tr in trs:
            sms[tr.class_id].forward()
        tc = time.perf_counter()
        for tr in trs:
            tr.instances = sms[tr.class_id].backward_and_track(name, vol.shape)
            tr.finished = True
        for tr in trs:
            sparse.remove_small_objects(tr, eng.min_size)
            sparse.remove_pancakes(tr, eng.min_extent)
        td = time.perf_counter()
        trackers[name] = trs
        t[name] = {'forward_post_s': round(tb - ta, 3), 'dense_to_runs_s': round(tb2 - tb, 3),
                   'forward_match_s': round(tc - tb2, 3),
                   'backward_track_s': round(td - tc, 3), 'objects': len(trs[0].instances)}
    te = time.perf_counter()
    out = list(tracker_consensus(trackers, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                 allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32))
    tf = time.perf_counter()
    total = tf - t0
    print(json.dumps({'metric': 'voxels/sec, 3-D ortho-plane stack + consensus', 'value': round(vol.size / job_s, 1),
                      'unit': 'voxels/s', 'volume': list(vol.shape), 'job': job,
                      'staged_seconds': round(total, 3), 'axes': t,
                      'consensus_fill_s': round(tf - te, 3), 'consensus_objects': len(out[0][2]), 'batch': batch,
                      'cpu_baseline': cpu_baseline(cfg, P, vol, 12) if '--cpu' in sys.argv else None}))


if __name__ == '__main__':
    main()
