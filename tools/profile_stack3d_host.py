"""cProfile of the host side of one axis of the 3-D pipeline (RLE extraction + forward / backward matching + tracking)."""
import cProfile, os, pstats, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from empanada_napari_amd import synth, weights, sparse
from empanada_napari_amd.engines import HipPanopticDeepLab
from empanada_napari_amd.inference import Engine3d
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(weights.MITONET_PDL_CFG)
P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
model = HipPanopticDeepLab(P, cfg, folded=True)
mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
      'norms': {'mean': 0.57571, 'std': 0.12765}}
vol = synth.blob_volume(size, size, size, seed=0, n_blobs=max(8, (size // 32) ** 2))
eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
               min_size=500, min_extent=5, batch_size=64)
pans = eng.predict_slices(vol, 0)
torch.cuda.synchronize()


def host():
    trs = eng.create_trackers(vol.shape, 'xy')
    matchers = sparse.create_matchers(eng.thing_list, eng.label_divisor, eng.merge_iou_thr, eng.merge_ioa_thr)
    rle_stack = []
    for i0 in range(0, len(pans), 64):
        for seg in sparse.pan_stack_to_rle_segs(torch.stack(pans[i0:i0 + 64]), eng.labels, eng.label_divisor,
                                                eng.thing_list, True):
            rle_stack.append(sparse.apply_matchers(seg, matchers))
    for index, seg in sparse.backward_matching(rle_stack, matchers, vol.shape[0]):
        sparse.update_trackers(seg, index, trs)
    sparse.finish_tracking(trs)


pr = cProfile.Profile()
pr.enable()
host()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
