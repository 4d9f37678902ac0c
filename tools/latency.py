"""Batch-1 latency of the reference-style call (one tile -> label map on the device): python tools/latency.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine  # noqa: E402
from empanada_napari_amd.preprocess import normalize_params  # noqa: E402

cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
eng = PanopticDeepLabRenderEngine(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                                  padding_factor=16, coarse_boundaries=True)
sub, mul = normalize_params(0.57571, 0.12765, 255)
for size in (1024, 512):
    x = torch.from_numpy(synth.em_tiles(1, size, seed=1))[:, None].cuda()
    for _ in range(5):
        eng.call_raw(x, sub, mul)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.call_raw(x, sub, mul)
    torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 30 * 1e3
    t0 = time.perf_counter()
    for _ in range(30):
        model(x, 2, False, sub=float(sub), mul=float(mul))
    torch.cuda.synchronize()
    fwd = (time.perf_counter() - t0) / 30 * 1e3
    print(f'{size}^2 batch 1: call {lat:.3f} ms, forward only {fwd:.3f} ms  (EMP_CONV_SMALL_TILES_BELOW={os.environ.get("EMP_CONV_SMALL_TILES_BELOW", "0")})')
