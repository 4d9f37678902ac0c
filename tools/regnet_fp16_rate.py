import sys, time, json, torch, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tests'))
import __graft_entry__ as ge
ge.load_package()
from empanada_napari_amd import synth
from empanada_napari_amd.engines import HipPanopticDeepLab
from test_regnet import regnet_model
for tag in ('x', 'y'):
    cfg, P = regnet_model(tag)
    m = HipPanopticDeepLab(P, cfg, folded=True, precision='fp16')
    for B in (1, 4, 16):
        x = torch.from_numpy(synth.em_tiles(B, 1024, seed=3))[:, None].cuda()
        sub, mul = 0.57571 * 255, 1 / (0.12765 * 255)
        m(x, 2, False, sub=sub, mul=mul); torch.cuda.synchronize()
        t = time.perf_counter(); R = 5
        for _ in range(R): m(x, 2, False, sub=sub, mul=mul)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
        print(f'{cfg["arch"]}/{cfg["encoder"]} fp16 batch {B}: {dt*1e3:.2f} ms = {B/dt:.1f} tiles/s', flush=True)
