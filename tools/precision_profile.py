"""One precision mode of the library (fp16x3 / fp32) under rocprofv3: forward + voting + merge of B x 1024^2 tiles, R times.
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o px -- python3 tools/precision_profile.py fp16x3 8 3"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine, logits_to_prob  # noqa: E402
from empanada_napari_amd.preprocess import normalize_params  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
arch = sys.argv[4] if len(sys.argv) > 4 else 'pdl'
cfg = dict(weights.MITONET_PDL_CFG if arch == 'pdl' else weights.MITONET_MINI_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
m = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
eng = PanopticDeepLabRenderEngine(m, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                                  padding_factor=16, coarse_boundaries=True)
tiles = torch.from_numpy(synth.em_tiles(B, 1024, seed=1234))[:, None].cuda()
sub, mul = normalize_params(0.57571, 0.12765, 255)


def step():
    o = m(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
    sem = logits_to_prob(o['sem_logits'])
    cells, _, _, kmax = eng.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
    return eng.panoptic_merge_int(sem, cells, kmax)


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f'{prec} {arch} batch {B}: {dt * 1e3:.2f} ms per step = {B / dt:.1f} tiles/s, {m.last_flops() / dt / 1e12:.1f} TFLOP/s fp32-equivalent')
