"""Forward throughput of the BiFPN variant (MitoNet_v1_mini class) on synthetic tiles.
    python tools/bench_bifpn.py [batch=32] [size=1024]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.preprocess import normalize_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = dict(weights.MITONET_MINI_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
base = synth.em_tiles(4, S, seed=1)
x = torch.from_numpy(np.concatenate([base] * (B // 4))[:B])[:, None].cuda()
sub, mul = normalize_params(0.57571, 0.12765, 255)
for _ in range(2):
    model(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    model(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'BiFPN-PR forward: batch {B} x {S}^2: {dt * 1e3:.2f} ms = {B / dt:.0f} tiles/s, {model.last_flops() / dt / 1e12:.0f} TFLOP/s '
      f'({model.last_flops() / B / 1e9:.1f} GFLOP per tile), arena {model.arena_bytes() / 2 ** 30:.1f} GiB')
