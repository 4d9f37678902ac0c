"""Host-side timeline of the 512^3 ortho-plane job (BASELINE configs[2]): per axis the time to the first chunk, to the last chunk and to
the return of infer_on_axis, then consensus + fill.  python tools/axis_timeline.py   (EMP_TOOL_PRECISION=fp16|fp16x3|fp32)"""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.load_package()
from empanada_napari_amd import synth, weights, inference
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.inference import Engine3d, tracker_consensus
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16x3'))
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16, 'norms': {'mean': 0.57571, 'std': 0.12765}}
e3 = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5, min_size=500, min_extent=5)
vol = synth.blob_volume(512, 512, 512, seed=0, n_blobs=256, fast=True)
T0 = [0.0]
log = []
orig = Engine3d.iter_slice_chunks
def wrapped(self, volume, axis, post_stream=None):
    t_in = time.perf_counter()
    first = True
    for ch in orig(self, volume, axis, post_stream=post_stream):
        if first:
            log.append(('first chunk after', round(1e3 * (time.perf_counter() - t_in), 1)))
            first = False
        yield ch
    log.append(('chunks done after', round(1e3 * (time.perf_counter() - t_in), 1)))
Engine3d.iter_slice_chunks = wrapped
def job():
    trs = {}
    for name in ('xy', 'xz', 'yz'):
        t = time.perf_counter()
        trs[name] = e3.infer_on_axis(vol, name)[1]
        log.append((name + ' infer_on_axis returned after', round(1e3 * (time.perf_counter() - t), 1)))
    t = time.perf_counter()
    out = list(tracker_consensus(trs, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75, allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32, chunk_size=(256, 256, 256)))
    log.append(('consensus + fill', round(1e3 * (time.perf_counter() - t), 1)))
job(); torch.cuda.synchronize(); log.clear()
t = time.perf_counter(); job(); torch.cuda.synchronize()
print('job', round(1e3 * (time.perf_counter() - t), 1))
for l in log: print(l)
