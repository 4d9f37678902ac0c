"""Forward time at small batches against the few-tile threshold (EMP_CONV_SMALL_TILES_BELOW): python tools/small_batch.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.preprocess import normalize_params
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
sub, mul = normalize_params(0.57571, 0.12765, 255)
for B, S in ((1, 1024), (2, 1024), (4, 1024), (8, 1024), (16, 1024), (32, 1024), (16, 512), (64, 512)):
    x = torch.from_numpy(synth.em_tiles(B, S, seed=1))[:, None].cuda()
    for _ in range(3):
        model(x, 2, False, sub=float(sub), mul=float(mul))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model(x, 2, False, sub=float(sub), mul=float(mul))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f'B={B:3d} x {S}^2: {ms:7.3f} ms/forward = {ms / (B * S * S / 1024 ** 2):6.3f} ms per 1024^2-equivalent  (SMALL_TILES_BELOW={os.environ.get("EMP_CONV_SMALL_TILES_BELOW", "256")})')
