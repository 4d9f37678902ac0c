"""The synthetic matcher -> tracker case shared by the oracle tests and the product tests
(inputs identical to oracle/gen_golden.py::gen_sparse; deterministic numpy generators)."""
import numpy as np


def synth_label_volume(shape, n_obj, seed):
    rng = np.random.default_rng(seed)
    d, h, w = shape
    zz, yy, xx = np.mgrid[0:d, 0:h, 0:w].astype(np.float32)
    vol = np.zeros(shape, dtype=np.int64)
    for i in range(1, n_obj + 1):
        c = rng.uniform(0.15, 0.85, 3) * np.array(shape)
        r = rng.uniform(0.08, 0.22, 3) * np.array(shape)
        m = ((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 < 1
        vol[m] = i
    return vol


def axis_pan_slices(vol, axis, divisor, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(vol.shape[axis]):
        sl = np.take(vol, i, axis=axis).copy()
        drop = rng.random(sl.shape) < 0.03
        sl[drop] = 0
        ids = np.unique(sl)
        ids = ids[ids > 0]
        perm = rng.permutation(len(ids)) + 1
        pan = np.zeros_like(sl)
        for k, v in zip(perm, ids):
            pan[sl == v] = divisor + k
        out.append(pan)
    return out


SHAPE = (24, 28, 32)
DIVISOR = 1000


def run_axis_pipeline(impl, to_rle=None):
    """impl: module/object providing RLEMatcher, InstanceTracker, pan_seg_to_rle_seg (oracle or product)."""
    vol = synth_label_volume(SHAPE, 7, seed=5)
    to_rle = to_rle or impl.pan_seg_to_rle_seg
    trackers = []
    for axis, name in enumerate(('xy', 'xz', 'yz')):
        slices = axis_pan_slices(vol, axis, DIVISOR, seed=100 + axis)
        stack = []
        m = impl.RLEMatcher(1, DIVISOR, 0.25, 0.25)
        for pan in slices:
            seg = to_rle(pan, [1], DIVISOR, [1], force_connected=True)
            if m.target_rle is None:
                m.initialize_target(seg[1])
            else:
                seg[1] = m(seg[1])
            stack.append(seg)
        m.target_rle = None
        m.assign_new = False
        tr = impl.InstanceTracker(1, DIVISOR, SHAPE, name)
        for idx in range(len(slices) - 1, -1, -1):
            seg = stack[idx]
            if m.target_rle is None:
                m.initialize_target(seg[1])
            else:
                seg[1] = m(seg[1])
            tr.update(seg[1], idx)
        tr.finish()
        trackers.append(tr)
    return trackers
