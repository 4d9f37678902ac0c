"""Host-side profile of ONE Engine3d.infer_on_axis pass on big slices (the shape of BASELINE configs[3]):
    python tools/profile_axis.py [depth=128] [size=1024]"""
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
from empanada_napari_amd.inference import Engine3d  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
      'norms': {'mean': 0.57571, 'std': 0.12765}}
vol = synth.ProceduralVolume((D, S, S), seed=7, cell=48).block(0, 0, D, 'cuda').cpu().numpy()
eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
               min_size=500, min_extent=5)
_, tr = eng.infer_on_axis(vol, 'xy')
print('objects', len(tr[0].instances))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    _, tr = eng.infer_on_axis(vol, 'xy')
    t1 = time.perf_counter()
    n = len(tr[0].instances)
    t2 = time.perf_counter()
    print(f'infer_on_axis returns after {1e3 * (t1 - t0):.1f} ms, deferred tail joined after {1e3 * (t2 - t0):.1f} ms '
          f'({D * S * S / (t2 - t0) / 1e6:.1f} Mvoxel/s; forward alone would be {D * (S / 1024) ** 2 / 1180 * 1e3:.1f} ms at 1180 tiles/s)')
pr = cProfile.Profile()
pr.enable()
_, tr = eng.infer_on_axis(vol, 'xy')
len(tr[0].instances)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
