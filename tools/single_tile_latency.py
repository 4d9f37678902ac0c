"""configs[0]: MitoNet_mini (PanopticBiFPN-PR) on ONE 512 x 512 tile: single-call latency of the forward and of Engine2d.infer, by precision"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.inference import Engine2d
from empanada_napari_amd.preprocess import normalize_params
for arch, cfg0 in (('BiFPN-PR mini', weights.MITONET_MINI_CFG), ('PDL-PR', weights.MITONET_PDL_CFG)):
    for S in (512, 1024):
        for prec in ('fp16x3', 'fp16'):
            cfg = dict(cfg0)
            P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
            model = HipPanopticDeepLab(P, cfg, folded=True, precision=prec)
            sub, mul = normalize_params(0.57571, 0.12765, 255)
            img = synth.em_tiles(1, S, seed=1)
            x = torch.from_numpy(img)[:, None].cuda()
            for _ in range(3):
                model(x, 2, False, sub=float(sub), mul=float(mul))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model(x, 2, False, sub=float(sub), mul=float(mul))
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t0) / 20 * 1e3
            mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 128 if 'BiFPN' in arch else 16,
                  'norms': {'mean': 0.57571, 'std': 0.12765}}
            e2 = Engine2d(mc, label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5)
            for _ in range(3):
                e2.infer(img[0])
            t0 = time.perf_counter()
            for _ in range(20):
                e2.infer(img[0])
            inf = (time.perf_counter() - t0) / 20 * 1e3
            print(f'{arch:14s} {S}^2 {prec:7s}: forward {fwd:6.3f} ms, Engine2d.infer (host uint8 -> host label map) {inf:6.3f} ms', flush=True)
            del model, e2
