"""Engine2d.infer_batch under rocprofv3 --kernel-trace --stats, plus event timings of its stages (GPU tool).
   python tools/engine2d_profile.py [chunks]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from empanada_napari_amd import sparse, synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab  # noqa: E402
from empanada_napari_amd.inference import Engine2d  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
      'norms': {'mean': 0.57571, 'std': 0.12765}}
e2 = Engine2d(mc, label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5)
tiles = synth.em_tiles(32, 1024, seed=5)
imgs = [tiles[i % 32] for i in range(32 * chunks)]
e2.infer_batch(imgs[:64], batch=32)
torch.cuda.synchronize()
for rnd in range(2):
    t0 = time.perf_counter()
    out = e2.infer_batch(imgs, batch=32)
    dt = time.perf_counter() - t0
    print(f'infer_batch: {len(imgs) / dt:.1f} tiles/s ({dt / chunks * 1e3:.1f} ms per 32-tile chunk)')
# stage timings on one resident chunk
dev = torch.device('cuda:0')
x = torch.from_numpy(tiles)[:, None].to(dev)
from empanada_napari_amd.preprocess import normalize_params
from empanada_napari_amd.engines import logits_to_prob
sub, mul = normalize_params(0.57571, 0.12765, 255)
eng = e2.engine


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, r


ms_f, mo = timed(lambda: eng.model(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul)))
ms_p, sem = timed(lambda: logits_to_prob(mo['sem_logits']))
ms_c, cc = timed(lambda: eng.instance_cells_int(mo['ctr_hmp'], mo['offsets'], 1))
ms_m, pan = timed(lambda: eng.panoptic_merge_int(sem, cc[0], cc[3]))
ms_m2, _ = timed(lambda: eng.panoptic_merge_int(sem, cc[0], 4096))
ms_fc, lab = timed(lambda: sparse.force_connected(pan, [1], 10000))
pin = torch.empty((32, 1024, 1024), dtype=torch.int32, pin_memory=True)
ms_d2h, _ = timed(lambda: pin.copy_(lab, non_blocking=True))
pin_in = torch.empty((32, 1, 1024, 1024), dtype=torch.uint8, pin_memory=True)
ms_h2d, _ = timed(lambda: x.copy_(pin_in, non_blocking=True))
t = time.perf_counter()
st = pin_in.numpy()
for j in range(32):
    st[j, 0] = tiles[j]
ms_host = (time.perf_counter() - t) * 1e3
print(f'forward {ms_f:.2f}  prob {ms_p:.2f}  cells {ms_c:.2f} (kmax {cc[3]})  merge {ms_m:.2f} (with max_ids 4096: {ms_m2:.2f})  '
      f'force_connected {ms_fc:.2f}  D2H {ms_d2h:.2f}  H2D {ms_h2d:.2f}  host staging copy {ms_host:.2f}  [ms per 32 tiles]')
