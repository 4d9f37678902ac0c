"""cProfile of one axis (yz) and of tracker_consensus on the 512^3 bench volume (host-side hot spots of the 3-D job).
    python tools/profile_stack3d.py"""
import sys, time, cProfile, pstats, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights, sparse
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.inference import Engine3d, tracker_consensus
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
      'norms': {'mean': 0.57571, 'std': 0.12765}}
vol = synth.blob_volume(512, 512, 512, seed=0, n_blobs=256, fast=True)
eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500, min_extent=5)
eng.infer_on_axis(vol[:64], 'xy')
trs = {}
for name in ('xy', 'xz', 'yz'):
    pr = cProfile.Profile() if name == 'yz' else None
    if pr: pr.enable()
    _, trs[name] = eng.infer_on_axis(vol, name)
    if pr:
        pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
print({k: (len(v[0].instances), sum(len(d['starts']) for d in v[0].instances.values())) for k, v in trs.items()})
pr = cProfile.Profile(); pr.enable()
out = list(tracker_consensus(trs, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32))
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
