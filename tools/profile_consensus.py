"""Where the consensus + fill stage of the 512^3 ortho-plane job spends its time (cProfile of tracker_consensus after the
three axis passes):  python tools/profile_consensus.py [size]"""
import cProfile
import os
import pstats
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

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
      'norms': {'mean': 0.57571, 'std': 0.12765}}
vol = synth.blob_volume(size, size, size, seed=0, n_blobs=max(8, (size // 32) ** 2), fast=True)
eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500,
               min_extent=5)
kw = dict(label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32)
for rep in range(2):
    tr = {name: eng.infer_on_axis(vol, name)[1] for name in ('xy', 'xz', 'yz')}
    for name in tr:
        [len(t.instances) for t in tr[name]]      # join the deferred passes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep == 1:
        pr = cProfile.Profile()
        pr.enable()
    out = list(tracker_consensus(tr, None, mc, **kw))
    if rep == 1:
        pr.disable()
    print(f'consensus + fill: {1e3 * (time.perf_counter() - t0):.1f} ms, {len(out[0][2])} objects, runs per axis: '
          + ', '.join(str(sum(len(o["starts"]) for o in t[0].instances.values())) for t in tr.values()))
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
