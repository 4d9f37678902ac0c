"""Stage timing of the host matcher / tracker on one axis of big slices: python tools/profile_match.py [depth] [size]"""
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
pans = eng.predict_slices(vol, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
runs_all = []
for i0 in range(0, len(pans), 64):
    chunk = torch.stack(pans[i0:i0 + 64])
    per = sparse.pan_stack_to_runs(chunk, [1], 10000, [1], True)
    runs_all += per[1][0]
    off = per[1][1]
t1 = time.perf_counter()
nruns = [len(r) for r in runs_all]
nobj = [len(np.unique(r[:, 2])) for r in runs_all]
print(f'dense->runs {1e3*(t1-t0):.1f} ms; runs per slice mean {np.mean(nruns):.0f} max {max(nruns)}; objects per slice mean {np.mean(nobj):.0f} max {max(nobj)}')
if os.environ.get('EMP_DUMP_RUNS'):      # the run lists as a fixture for profiling the host matcher without a GPU
    np.savez_compressed(os.environ['EMP_DUMP_RUNS'], off=np.int64(off), S=np.int64(S), n=np.array(nruns, dtype=np.int64),
                        runs=np.concatenate(runs_all).astype(np.int32))
sm = sparse.StackMatcher(1, 10000, 0.25, 0.25, match=True)
t0 = time.perf_counter()
for r in runs_all:
    sm.push_runs(r, S, off)
t1 = time.perf_counter()
sm.forward()
t2 = time.perf_counter()
inst = sm.backward_and_track('xy', vol.shape)
t3 = time.perf_counter()
print(f'push {1e3*(t1-t0):.1f} ms  forward {1e3*(t2-t1):.1f} ms  backward+track+export {1e3*(t3-t2):.1f} ms  tracks {len(inst)}')
tr = sparse.InstanceTracker(1, 10000, vol.shape, 'xy')
tr.instances = inst
t4 = time.perf_counter()
sparse.remove_small_objects(tr, 500)
sparse.remove_pancakes(tr, 5)
print(f'filters {1e3*(time.perf_counter()-t4):.1f} ms -> {len(tr.instances)} objects')
