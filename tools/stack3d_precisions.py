import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import __graft_entry__ as g; g.load_package()
from empanada_napari_amd import synth, weights
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.inference import Engine3d, tracker_consensus
size = 512
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
vol = synth.blob_volume(size, size, size, seed=0, n_blobs=(size // 32) ** 2, fast=True)
res = {}
for prec in ('fp16', 'fp16x3'):
    model = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16, 'norms': {'mean': 0.57571, 'std': 0.12765}}
    eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500, min_extent=5)
    kw = dict(label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32)
    def job():
        tr = {name: eng.infer_on_axis(vol, name)[1] for name in ('xy', 'xz', 'yz')}
        return list(tracker_consensus(tr, None, mc, **kw))
    job(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = job(); dt = time.perf_counter() - t0
    res[prec] = {'seconds': round(dt, 3), 'Mvoxel_per_s': round(vol.size / dt / 1e6, 1), 'consensus_objects': len(out[0][2])}
    print(prec, res[prec], flush=True)
    del eng, model; torch.cuda.empty_cache()
json.dump(res, open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'stack3d_precisions.json'), 'w'), indent=1)
