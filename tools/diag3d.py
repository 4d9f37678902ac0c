import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights
from empanada_napari_amd.engines import HipPanopticDeepLab, factor_pad, logits_to_prob
from empanada_napari_amd.inference import Engine3d, take
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16, 'norms': {'mean': 0.57571, 'std': 0.12765}}
S, B = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 16
vol = synth.blob_volume(64, S, S, seed=0)
eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, confidence_thr=0.5, batch_size=B)
e = eng.engine
def T(): torch.cuda.synchronize(); return time.perf_counter()
eng.predict_slices(vol[:B], 0)
t0 = T()
imgs = [eng.preprocessor(np.asarray(take(vol, i, 0)))['image'] for i in range(B)]
t1 = T(); x = factor_pad(torch.stack(imgs), 16); t2 = T()
xd = e.to_model_device(x); t3 = T()
for _ in range(3): mo = e.model(xd, 2, interpolate_ins=False)
t4 = T()
sem = logits_to_prob(mo['sem_logits']); t5 = T()
cells, _, _, kmax = e.instance_cells_int(mo['ctr_hmp'], mo['offsets'], 1); t6 = T()
pan = e.panoptic_merge_int(sem, cells, kmax); t7 = T()
print(f'B={B}: preprocess {t1-t0:.4f} stack/pad {t2-t1:.4f} h2d {t3-t2:.4f} forward {(t4-t3)/3:.4f} prob {t5-t4:.4f} cells {t6-t5:.4f} merge {t7-t6:.4f}')
ta = T(); out = eng.predict_slices(vol, 0); tb = T()
print(f'predict_slices 64 slices: {tb-ta:.3f} s -> {64/(tb-ta):.1f} slices/s')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); out = eng.predict_slices(vol, 0); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
