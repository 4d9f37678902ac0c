"""numpy / pure-Python restatement of the sparse label algebra (oracle, test only).

Run-length ("RLE") objects are ``{'box', 'starts', 'runs'}`` over row-major
raveled indices, exactly the reference's schema.  Every function cites the
reference lines it restates.  Pinned by:
  * the reference's own unit tests (tests/test_array_utils.py, tests/test_zarr_utils.py),
    restated in tests/test_oracle_sparse.py;
  * golden vectors generated from the imported reference (numba / skimage stubs,
    oracle/gen_golden.py -> tests/golden/sparse.npz) for the matcher, tracker,
    voting, joining and the ortho-plane consensus.
PARITY UNPINNED: ``connected_components`` / ``pan_seg_to_rle_seg`` restate
scikit-image semantics (``measure.label`` full connectivity, raster-order label
numbers; ``regionprops`` half-open bbox, row-major coords) that the reference pins
nowhere (scikit-image is absent here); they are checked for self-consistency only.
"""
import math
from itertools import combinations

import numpy as np
from scipy.optimize import linear_sum_assignment

i64 = np.int64


# ----------------------------------------------------------------------------
# boxes (array_utils.py:105-129, 148-211)
# ----------------------------------------------------------------------------
def merge_boxes(b1, b2):
    n = len(b1)
    h = n // 2
    return tuple(min(b1[i], b2[i]) if i < h else max(b1[i], b2[i]) for i in range(n))


def box_iou_pairs(boxes1, boxes2):
    """array_utils.py:148-176 (_box_iou): rows, cols, ious, intersections of the non-zero pairs."""
    boxes1 = np.asarray(boxes1)
    boxes2 = np.asarray(boxes2)
    rows, cols, ious, inters = [], [], [], []
    if len(boxes1) == 0 or len(boxes2) == 0:
        return rows, cols, ious, inters
    nd = boxes1.shape[1] // 2
    lo = np.maximum(boxes1[:, None, :nd], boxes2[None, :, :nd])
    hi = np.minimum(boxes1[:, None, nd:], boxes2[None, :, nd:])
    inter = np.prod(np.maximum(0, hi - lo), axis=2)
    a1 = np.prod(boxes1[:, nd:] - boxes1[:, :nd], axis=1)
    a2 = np.prod(boxes2[:, nd:] - boxes2[:, :nd], axis=1)
    for r, c in zip(*np.nonzero(inter > 0)):
        rows.append(int(r))
        cols.append(int(c))
        ious.append(inter[r, c] / (a1[r] + a2[c] - inter[r, c]))
        inters.append(inter[r, c])
    return rows, cols, ious, inters


# ----------------------------------------------------------------------------
# run-length primitives (array_utils.py:213-256, 344-459)
# ----------------------------------------------------------------------------
def rle_encode(indices):
    indices = np.asarray(indices)
    brk = np.flatnonzero(indices[1:] != indices[:-1] + 1) + 1
    edges = np.concatenate([[0], brk, [len(indices)]])
    return indices[edges[:-1]], np.diff(edges)


def rle_decode(starts, runs):
    return np.concatenate([np.arange(s, s + r) for s, r in zip(starts, runs)])


def intersection_from_ranges(merged, changes):
    """array_utils.py:344-373, same scan."""
    total = 0
    check = None
    for chg, r1, r2 in zip(changes, merged[:-1], merged[1:]):
        if chg:
            check = r1
        elif check is None:
            continue
        if check[1] < r2[0]:
            continue
        total += min(check[1], r2[1]) - max(check[0], r2[0])
    return total


def rle_intersection(sa, ra, sb, rb):
    """array_utils.py:375-407."""
    a = np.stack([sa, sa + ra], axis=1)
    b = np.stack([sb, sb + rb], axis=1)
    merged = np.concatenate([a, b], axis=0)
    ids = np.concatenate([np.zeros(len(a), i64), np.ones(len(b), i64)])
    order = np.argsort(merged, axis=0, kind='stable')[:, 0]
    merged, ids = merged[order], ids[order]
    return intersection_from_ranges(merged, ids[:-1] != ids[1:])


def rle_iou(sa, ra, sb, rb, return_intersection=False):
    inter = rle_intersection(sa, ra, sb, rb)
    union = ra.sum() + rb.sum() - inter
    return (inter / union, inter) if return_intersection else inter / union


def rle_ioa(sa, ra, sb, rb):
    return rle_intersection(sa, ra, sb, rb) / rb.sum()


# ----------------------------------------------------------------------------
# voting / joining (array_utils.py:461-752)
# ----------------------------------------------------------------------------
def split_range_by_votes(running, votes, vote_thr=2):
    """array_utils.py:461-519: maximal sub-ranges whose every index has >= vote_thr votes."""
    ok = np.asarray(votes) >= vote_thr
    out = []
    i, n = 0, len(ok)
    while i < n:
        if ok[i]:
            j = i
            while j + 1 < n and ok[j + 1]:
                j += 1
            out.append([running[0] + i, running[0] + j + 1])
            i = j + 1
        else:
            i += 1
    return np.array(out, dtype=i64).reshape(-1, 2)


def extend_range(r1, r2, votes):
    """array_utils.py:521-561 (mutates r1 like the reference)."""
    first = r2[0] - r1[0]
    last = len(votes)
    off = r2[1] - r1[1]
    if off > 0:
        r1[1] = r2[1]
        votes = np.concatenate([votes, np.ones(off, dtype=i64)])
    elif off < 0:
        last += off
    # the reference loops `for i in range(first, last): votes[i] += 1`: a negative first index wraps
    # around (exercised by its own unit test with unsorted input), so no slice arithmetic here
    np.add.at(votes, np.arange(first, last), 1)
    return r1, votes


def rle_voting(ranges, vote_thr=2):
    """array_utils.py:563-625 with init_index = term_index = None (all call sites)."""
    assert vote_thr > 1
    voted = []
    running = None
    votes = None
    for r1, r2 in zip(ranges[:-1], ranges[1:]):
        if running is None:
            running = r1
            votes = np.ones(r1[1] - r1[0], dtype=i64)
        if running[1] < r2[0]:
            voted.append(split_range_by_votes(running, votes, vote_thr))
            running, votes = None, None
        else:
            running, votes = extend_range(running, r2, votes)
    # the reference finishes with the (possibly empty) running range (:621-623)
    if running is not None:
        voted.append(split_range_by_votes(running, votes, vote_thr))
    if not voted:
        return np.empty((0, 2), dtype=i64)
    return np.concatenate(voted, axis=0)


def concat_sort_ranges(list_of_ranges):
    lst = [r for r in list_of_ranges if len(r) > 0]
    ranges = np.concatenate(lst, axis=0)
    return ranges[np.argsort(ranges[:, 0], kind='stable')]


def _join_ranges(ranges):
    """array_utils.py:658-691 (mutates rows in place; a trailing non-overlapping range is appended)."""
    joined = []
    running = None
    r2 = None
    for r1, r2 in zip(ranges[:-1], ranges[1:]):
        if running is None:
            running = r1
        if running[1] >= r2[0]:
            running[1] = max(running[1], r2[1])
        else:
            joined.append(running.copy())
            running = None
    if running is not None:
        joined.append(running.copy())
    else:
        joined.append((ranges[0] if r2 is None else r2).copy())  # single-range input: unbound in the reference
    return np.array(joined, dtype=i64).reshape(-1, 2)


def join_ranges(list_of_ranges):
    lst = [r for r in list_of_ranges if len(r) > 0]
    return _join_ranges(concat_sort_ranges(lst))


def vote_by_ranges(list_of_ranges, vote_thr=2):
    """array_utils.py:627-639."""
    lst = [r for r in list_of_ranges if len(r) > 0]
    if vote_thr == 1:
        return join_ranges(lst)
    if len(lst) >= vote_thr:
        return np.array(rle_voting(concat_sort_ranges(lst), vote_thr))
    return np.array([])


def invert_ranges(ranges, size):
    """array_utils.py:701-717 (no overlap handling: the reference test expects [6,4])."""
    out = []
    if ranges[0][0] > 0:
        out.append([0, ranges[0][0]])
    for r1, r2 in zip(ranges[:-1], ranges[1:]):
        if r1[1] != r2[0]:
            out.append([r1[1], r2[0]])
    if ranges[-1][1] < size:
        out.append([ranges[-1][1], size])
    return np.array(out, dtype=i64).reshape(-1, 2)


def merge_rles(sa, ra, sb=None, rb=None):
    """array_utils.py:719-752."""
    lst = [np.stack([sa, sa + ra], axis=1)]
    if sb is not None and rb is not None:
        lst.append(np.stack([sb, sb + rb], axis=1))
    j = join_ranges(lst)
    return j[:, 0], j[:, 1] - j[:, 0]


def numpy_fill_instances(volume, instances):
    """array_utils.py:754-766: later instances overwrite earlier ones."""
    flat = volume.reshape(-1)
    for iid, attrs in instances.items():
        for s, r in zip(attrs['starts'], attrs['runs']):
            flat[s:s + r] = iid
    return volume


def chunk_ranges(ranges, modulo, divisor):
    """zarr_utils.py:20-56: split ranges where (index % modulo) // divisor changes."""
    out = []
    for r in np.asarray(ranges):
        cs = (r[0] % modulo) // divisor
        ce = ((r[1] - 1) % modulo) // divisor
        if cs != ce or (r[1] - r[0] > divisor):
            idx = np.arange(r[0], r[1] + 1)
            c = (idx % modulo) // divisor
            split = [0] + [i for i in range(1, len(idx)) if c[i] != c[i - 1]]
            if split[-1] != len(idx) - 1:
                split.append(-1)
            for i, j in zip(split[:-1], split[1:]):
                out.append([int(idx[i]), int(idx[j])])
        else:
            out.append([int(r[0]), int(r[1])])
    return out


def fill_func(seg1d, coords, instance_id):
    """zarr_utils.py:58-67."""
    for s, e in coords:
        seg1d[s:e] = instance_id
    return seg1d


# ----------------------------------------------------------------------------
# dense <-> RLE (inference/rle.py)  -- PARITY UNPINNED (scikit-image semantics)
# ----------------------------------------------------------------------------
def connected_components(seg):
    """rle.py:18-24: skimage.measure.label(seg): 8-connected regions of EQUAL non-zero value,
    numbered 1.. in raster order of each region's first pixel."""
    seg = np.asarray(seg)
    h, w = seg.shape
    parent = np.arange(h * w)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    flat = seg.reshape(-1)
    for y in range(h):
        for x in range(w):
            v = seg[y, x]
            if v == 0:
                continue
            p = y * w + x
            for dy, dx in ((0, -1), (-1, -1), (-1, 0), (-1, 1)):
                yy, xx = y + dy, x + dx
                if 0 <= yy < h and 0 <= xx < w and seg[yy, xx] == v:
                    a, b = find(p), find(yy * w + xx)
                    if a != b:
                        parent[max(a, b)] = min(a, b)
    out = np.zeros(h * w, dtype=np.int64)
    nxt = 0
    names = {}
    for p in range(h * w):
        if flat[p] == 0:
            continue
        r = find(p)
        if r not in names:
            nxt += 1
            names[r] = nxt
        out[p] = names[r]
    return out.reshape(h, w)


def regionprops_rle(instance_seg):
    """regionprops (.label ascending, .bbox half-open, .coords row-major) -> {label: rle attrs} (rle.py:73-81)."""
    attrs = {}
    flat = instance_seg.reshape(-1)
    for lab in np.unique(flat):
        if lab == 0:
            continue
        idx = np.flatnonzero(flat == lab)
        ys, xs = np.unravel_index(idx, instance_seg.shape)
        starts, runs = rle_encode(idx)
        attrs[int(lab)] = {'box': (int(ys.min()), int(xs.min()), int(ys.max()) + 1, int(xs.max()) + 1),
                           'starts': starts.astype(i64), 'runs': runs.astype(i64)}
    return attrs


def pan_seg_to_rle_seg(pan_seg, labels, label_divisor, thing_list, force_connected=True):
    """rle.py:26-86."""
    rle_seg = {}
    for label in labels:
        lo, hi = label * label_divisor, label * label_divisor + label_divisor
        inst = pan_seg.copy()
        inst[(pan_seg < lo) | (pan_seg >= hi)] = 0
        if force_connected and label in thing_list:
            inst = connected_components(inst)
            inst[inst > 0] += lo
        rle_seg[label] = regionprops_rle(inst)
    return rle_seg


def rle_seg_to_pan_seg(rle_seg, shape):
    """rle.py:88-118 (uint32 output, Q13)."""
    pan = np.zeros(shape, dtype=np.uint32).ravel()
    for inst in rle_seg.values():
        for oid, a in inst.items():
            for s, r in zip(a['starts'], a['runs']):
                pan[s:s + r] = oid
    return pan.reshape(shape)


def force_connected_pan(pan_seg, thing_list, label_divisor):
    """Engine2d.force_connected, empanada_napari/inference.py:263-279."""
    for label in thing_list:
        lo, hi = label * label_divisor, label * label_divisor + label_divisor
        inst = pan_seg.copy()
        inst[(pan_seg < lo) | (pan_seg >= hi)] = 0
        inst = connected_components(inst).astype(np.int32)
        inst[inst > 0] += lo
        pan_seg[inst > 0] = inst[inst > 0]
    return pan_seg


# ----------------------------------------------------------------------------
# matcher (inference/matcher.py:136-326)
# ----------------------------------------------------------------------------
def unpack(instance_rles):
    labels = [int(k) for k in instance_rles]
    boxes = [a['box'] for a in instance_rles.values()]
    starts = [a['starts'] for a in instance_rles.values()]
    runs = [a['runs'] for a in instance_rles.values()]
    return np.array(labels), np.array(boxes), starts, runs


def rle_matcher(target, match, iou_thr=0.5):
    """matcher.py:136-232 with return_ioa=True -> (matched, all_labels, matched_ious, ioa_matrix)."""
    tl, tb, ts, tr = unpack(target)
    ml, mb, ms, mr = unpack(match)
    if len(tl) == 0 or len(ml) == 0:
        e = np.array([])
        return (e, e), (tl, ml), e, e
    iou = np.zeros((len(tb), len(mb)), dtype='float')
    ioa = np.zeros((len(tb), len(mb)), dtype=np.float32)
    rows, cols, _, _ = box_iou_pairs(tb, mb)
    for r1, r2 in zip(rows, cols):
        iou[r1, r2] = rle_iou(ts[r1], tr[r1], ms[r2], mr[r2])
        ioa[r1, r2] = rle_ioa(ts[r1], tr[r1], ms[r2], mr[r2])
    mr_, mc_ = linear_sum_assignment(iou, maximize=True)
    keep = iou[mr_, mc_] >= iou_thr
    mr_, mc_ = mr_[keep], mc_[keep]
    return (tl[mr_], ml[mc_]), [tl, ml], iou[(mr_, mc_)], ioa


def merge_attrs(a1, a2):
    """matcher.py:14-28."""
    s, r = merge_rles(a1['starts'], a1['runs'], a2['starts'], a2['runs'])
    return {'box': merge_boxes(a1['box'], a2['box']), 'starts': s, 'runs': r}


class RLEMatcher:
    """matcher.py:234-326."""

    def __init__(self, class_id, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25, assign_new=True):
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.merge_iou_thr = merge_iou_thr
        self.merge_ioa_thr = merge_ioa_thr
        self.assign_new = assign_new
        self.next_label = class_id * label_divisor + 1
        self.target_rle = None

    def initialize_target(self, t):
        self.target_rle = t
        if len(t) > 0:
            self.next_label = max(t.keys()) + 1

    def __call__(self, match_rle, update_target=True):
        matched, all_labels, _, ioa = rle_matcher(self.target_rle, match_rle, self.merge_iou_thr)
        tl, ml = all_labels
        lm = {m: t for t, m in zip(matched[0], matched[1])}
        out = {}
        for i, (m, attrs) in enumerate(match_rle.items()):
            if m in lm:
                new = lm[m]
            else:
                ioa_max = ioa[:, i].max() if len(ioa) > 0 else 0
                if ioa_max >= self.merge_ioa_thr:
                    new = tl[ioa[:, i].argmax()]
                elif self.assign_new:
                    new = self.next_label
                    self.next_label += 1
                else:
                    new = m
            out[new] = attrs if new not in out else merge_attrs(out[new], attrs)
        if update_target:
            self.target_rle = out
        return out


def apply_matchers(rle_seg, matchers):
    """patterns.py:55-66."""
    for m in matchers:
        if m.target_rle is None:
            m.initialize_target(rle_seg[m.class_id])
        else:
            rle_seg[m.class_id] = m(rle_seg[m.class_id])
    return rle_seg


# ----------------------------------------------------------------------------
# tracker (inference/tracker.py) and filters (inference/filters.py:22-56)
# ----------------------------------------------------------------------------
AXIS_NUM = {'xy': 0, 'xz': 1, 'yz': 2}


class InstanceTracker:
    def __init__(self, class_id=None, label_divisor=None, shape3d=None, axis='xy'):
        assert axis in AXIS_NUM
        self.class_id, self.label_divisor, self.shape3d, self.axis = class_id, label_divisor, shape3d, axis
        self.finished = False
        self.instances = {}

    def update(self, instance_rles, index2d):
        """tracker.py:61-100."""
        assert not self.finished
        ax = AXIS_NUM[self.axis]
        shape2d = tuple(s for i, s in enumerate(self.shape3d) if i != ax)
        for label, a in instance_rles.items():
            h1, w1, h2, w2 = a['box']
            if self.axis == 'xy':
                box = (index2d, h1, w1, index2d + 1, h2, w2)
                starts = a['starts'] + index2d * math.prod(shape2d)
                runs = a['runs']
            elif self.axis == 'xz':
                box = (h1, index2d, w1, h2, index2d + 1, w2)
                hc, wc = np.unravel_index(a['starts'], shape2d)
                starts = np.ravel_multi_index((hc, np.full(len(hc), index2d), wc), self.shape3d)
                runs = a['runs']
            else:
                box = (h1, w1, index2d, h2, w2, index2d + 1)
                hc, wc = np.unravel_index(rle_decode(a['starts'], a['runs']), shape2d)
                starts = np.ravel_multi_index((hc, wc, np.full(len(hc), index2d)), self.shape3d)
                runs = np.ones_like(starts)
            if label not in self.instances:
                self.instances[label] = {'box': box, 'starts': [starts], 'runs': [runs]}
            else:
                d = self.instances[label]
                d['box'] = merge_boxes(box, d['box'])
                d['starts'].append(starts)
                d['runs'].append(runs)

    def finish(self):
        """tracker.py:102-123."""
        for d in self.instances.values():
            if isinstance(d['starts'], list):
                starts = np.concatenate(d['starts'])
                if self.axis == 'yz':
                    starts, runs = rle_encode(np.sort(starts, kind='stable'))
                else:
                    runs = np.concatenate(d['runs'])
                d['starts'], d['runs'] = starts, runs
        self.finished = True


def remove_small_objects(tracker, min_size=64):
    for k in list(tracker.instances):
        if tracker.instances[k]['runs'].sum() < min_size:
            del tracker.instances[k]


def remove_pancakes(tracker, min_span=4):
    for k in list(tracker.instances):
        b = tracker.instances[k]['box']
        if any(s < min_span for s in (b[3] - b[0], b[4] - b[1], b[5] - b[2])):
            del tracker.instances[k]


# ----------------------------------------------------------------------------
# optional morphology on a finished tracker (inference/filters.py:14-210).  erosion / dilation / binary_fill_holes
# are the scipy.ndimage calls skimage and the reference make themselves; measure.label / regionprops are restated.
# ----------------------------------------------------------------------------
def label_nd(seg):
    """filters.connected_components (filters.py:14-20) = skimage.measure.label(seg): regions of EQUAL non-zero value under
    full connectivity (8 in 2-D, 26 in 3-D), numbered 1.. in raster order of each region's first element."""
    from scipy import ndimage as ndi
    seg = np.asarray(seg)
    struct = np.ones((3,) * seg.ndim, dtype=bool)
    comps = []
    per_value = {}
    for v in np.unique(seg):
        if v == 0:
            continue
        lab, n = ndi.label(seg == v, structure=struct)
        per_value[v] = lab
        ids, first = np.unique(lab.ravel(), return_index=True)
        comps += [(int(f), v, int(i)) for i, f in zip(ids, first) if i != 0]
    out = np.zeros(seg.shape, dtype=i64)
    for number, (_, v, i) in enumerate(sorted(comps), start=1):
        out[per_value[v] == i] = number
    return out


def regionprops_rle_nd(instance_seg):
    """regionprops on an n-D label array -> {label: {'box' (mins..., maxs+1...), 'starts', 'runs'}}, labels ascending."""
    attrs = {}
    flat = instance_seg.reshape(-1)
    for lab in np.unique(flat):
        if lab == 0:
            continue
        idx = np.flatnonzero(flat == lab)
        coords = np.unravel_index(idx, instance_seg.shape)
        starts, runs = rle_encode(idx)
        attrs[int(lab)] = {'box': tuple(int(c.min()) for c in coords) + tuple(int(c.max()) + 1 for c in coords),
                           'starts': starts.astype(i64), 'runs': runs.astype(i64)}
    return attrs


def filters_pan_seg_to_rle_seg(pan_seg, labels, label_divisor, thing_list, force_connected=True):
    """filters.py:58-116: like rle.pan_seg_to_rle_seg on an n-D array, but ONE flat dict of instance attrs comes back."""
    instance_attrs = {}
    for label in labels:
        lo, hi = label * label_divisor, label * label_divisor + label_divisor
        inst = pan_seg.astype(i64)
        inst[(pan_seg < lo) | (pan_seg >= hi)] = 0
        if force_connected and label in thing_list:
            inst = label_nd(inst)
            inst[inst > 0] += lo
        instance_attrs.update(regionprops_rle_nd(inst))
    return instance_attrs


def filters_rle_seg_to_pan_seg(tracker, shape):
    """filters.py:118-152: tracker instances -> dense uint32 volume (later instances overwrite earlier ones)."""
    pan = np.zeros(shape, dtype=np.uint32).ravel()
    for oid, a in tracker.instances.items():
        for st, r in zip(a['starts'], a['runs']):
            pan[st:st + r] = oid
    return pan.reshape(shape)


def _cross(ndim):
    from scipy import ndimage as ndi
    return ndi.generate_binary_structure(ndim, 1)


def erode(tracker, volume_shape, labels, label_divisor, thing_list, iterations=1):
    """filters.py:154-164; skimage.morphology.erosion(mask) = ndi.grey_erosion(mask, footprint=cross) (mode 'reflect')."""
    from scipy import ndimage as ndi
    mask = filters_rle_seg_to_pan_seg(tracker, volume_shape)
    for _ in range(iterations):
        mask = ndi.grey_erosion(mask, footprint=_cross(mask.ndim))
    tracker.instances = filters_pan_seg_to_rle_seg(mask, labels, label_divisor, thing_list)
    return tracker


def dilate(tracker, volume_shape, labels, label_divisor, thing_list, iterations=1):
    """filters.py:166-176."""
    from scipy import ndimage as ndi
    mask = filters_rle_seg_to_pan_seg(tracker, volume_shape)
    for _ in range(iterations):
        mask = ndi.grey_dilation(mask, footprint=_cross(mask.ndim))
    tracker.instances = filters_pan_seg_to_rle_seg(mask, labels, label_divisor, thing_list)
    return tracker


def fill_holes_in_segmentation(tracker, volume_shape, labels, label_divisor, thing_list):
    """filters.py:178-210 (3-D only; the reference just prints for other ranks)."""
    from scipy.ndimage import binary_fill_holes
    mask_3d = filters_rle_seg_to_pan_seg(tracker, volume_shape)
    assert mask_3d.ndim == 3
    for idx in range(mask_3d.shape[0]):
        mask = mask_3d[idx]
        boxes = {lab: a['box'] for lab, a in regionprops_rle_nd(mask).items()}     # boxes of the slice BEFORE the loop
        for lab in sorted(boxes):
            minr, minc, maxr, maxc = boxes[lab]
            tmp = mask[minr:maxr, minc:maxc]
            tmp = binary_fill_holes(tmp.astype(bool))
            mask[minr:maxr, minc:maxc] = tmp.astype(mask.dtype) * lab
    tracker.instances = filters_pan_seg_to_rle_seg(mask_3d, labels, label_divisor, thing_list)
    return tracker


def instance_relabel(tracker):
    """empanada_napari/inference.py:31-54."""
    out = {}
    for i, a in enumerate(tracker.instances.values(), start=1):
        order = np.argsort(a['starts'], kind='stable')
        out[i] = {'box': a['box'], 'starts': a['starts'][order], 'runs': a['runs'][order]}
    return out


# ----------------------------------------------------------------------------
# ortho-plane consensus (consensus.py:7-469) on a minimal insertion-ordered graph
# that reproduces the networkx iteration orders the reference relies on (Q7, Q8)
# ----------------------------------------------------------------------------
MIN_OVERLAP = 100
MIN_IOU = 1e-2


class _Graph:
    """Undirected graph with networkx's orders: nodes / adjacency in insertion order."""

    def __init__(self):
        self.node = {}
        self.adj = {}

    def add_node(self, n, **attr):
        if n not in self.node:
            self.node[n] = {}
            self.adj[n] = {}
        self.node[n].update(attr)

    def add_edge(self, u, v, **attr):
        self.add_node(u)
        self.add_node(v)
        d = self.adj[u].get(v, {})
        d.update(attr)
        self.adj[u][v] = d
        self.adj[v][u] = d

    def remove_edge(self, u, v):
        del self.adj[u][v]
        if u != v:
            del self.adj[v][u]

    def remove_node(self, n):
        for m in list(self.adj[n]):
            del self.adj[m][n]
        del self.adj[n]
        del self.node[n]

    def has_edge(self, u, v):
        return v in self.adj.get(u, {})

    def edges(self):
        seen = set()
        for u in self.adj:
            for v, d in self.adj[u].items():
                if v not in seen:
                    yield u, v, d
            seen.add(u)

    def n_edges(self):
        return sum(len(a) for a in self.adj.values()) // 2

    def copy(self):
        g = _Graph()
        for n, a in self.node.items():
            g.add_node(n, **a)
        for u, v, d in self.edges():
            g.add_edge(u, v, **d)
        return g

    def subgraph(self, nodes):
        nodes = set(nodes)
        g = _Graph()
        for n in self.node:          # networkx subgraph views iterate in the parent's node order
            if n in nodes:
                g.add_node(n, **self.node[n])
        for u, v, d in self.edges():
            if u in nodes and v in nodes:
                g.add_edge(u, v, **d)
        return g

    def components(self):
        """nx.connected_components: BFS from nodes in insertion order, yields sets."""
        seen = set()
        for s in self.node:
            if s in seen:
                continue
            comp = {s}
            frontier = [s]
            while frontier:
                nxt = []
                for u in frontier:
                    for v in self.adj[u]:
                        if v not in comp:
                            comp.add(v)
                            nxt.append(v)
                frontier = nxt
            seen |= comp
            yield comp


def _avg_edge(G, c1, c2, key):
    w = [G.adj[a][b][key] if G.has_edge(a, b) else 0 for a in c1 for b in c2]
    return sum(w) / len(w)


def create_graph_of_clusters(G, thr):
    """consensus.py:35-75."""
    H = G.copy()
    for u, v, d in list(G.edges()):
        if d['iou'] <= thr:
            H.remove_edge(u, v)
    CG = _Graph()
    for i, cl in enumerate(H.components()):
        CG.add_node(i, cluster=cl)
    for n1, n2 in combinations(list(CG.node), 2):
        c1, c2 = CG.node[n1]['cluster'], CG.node[n2]['cluster']
        iw, ow = _avg_edge(G, c1, c2, 'iou'), _avg_edge(G, c1, c2, 'overlap')
        if iw > MIN_IOU or ow > MIN_OVERLAP:
            CG.add_edge(n1, n2, iou=iw, overlap=ow)
    return CG


def _push(G, src, dst):
    G.node[dst]['cluster'] = G.node[dst]['cluster'].union(G.node[src]['cluster'])
    G.remove_edge(src, dst)


def merge_clusters(G):
    """consensus.py:86-142 (incl. the (most_connected, neighbor) re-added edge, Q8)."""
    H = G.copy()
    while H.n_edges() > 0:
        mc = sorted(H.node, key=lambda x: len(H.adj[x]), reverse=True)[0]
        nbrs = sorted(H.adj[mc], key=lambda x: len(H.node[x]['cluster']), reverse=True)
        if len(H.node[nbrs[0]]['cluster']) > len(H.node[mc]['cluster']):
            for nb in nbrs:
                _push(H, mc, nb)
            H.remove_node(mc)
        else:
            for nb in nbrs:
                _push(H, nb, mc)
                for sn in list(H.adj[nb]):
                    if not H.has_edge(mc, sn):
                        H.add_edge(mc, nb, iou=H.adj[nb][sn]['iou'])
                H.remove_node(nb)
    return H


def merge_instances(d):
    """consensus.py:144-164."""
    vals = list(d.values())
    if len(vals) < 2:
        return vals[0]
    box, s, r = vals[0]['box'], vals[0]['starts'], vals[0]['runs']
    for a in vals[1:]:
        box = merge_boxes(box, a['box'])
        s, r = merge_rles(s, r, a['starts'], a['runs'])
    return dict(box=box, starts=s, runs=r)


def merge_overlapping(ci):
    """consensus.py:166-197."""
    if len(ci) < 2:
        return list(ci.values())
    g = _Graph()
    for k in ci:
        g.add_node(k)
    for a, b in combinations(list(ci), 2):
        iou, inter = rle_iou(ci[a]['starts'], ci[a]['runs'], ci[b]['starts'], ci[b]['runs'], True)
        if iou > MIN_IOU or inter > MIN_OVERLAP:
            g.add_edge(a, b)
    return [merge_instances({k: v for k, v in ci.items() if k in comp}) for comp in g.components()]


def bounding_box_screening(boxes, src):
    """consensus.py:199-236."""
    rows, cols, _, _ = box_iou_pairs(boxes, boxes)
    m = np.array([rows, cols]).T.reshape(-1, 2)
    m = m[src[m[:, 0]] != src[m[:, 1]]]
    m = np.sort(m, axis=-1)
    return np.unique(m, axis=0)


def merge_objects_from_trackers(trackers, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False):
    """consensus.py:348-469."""
    n_votes = len(trackers)
    min_cluster = 1 if bypass else n_votes // 2 + 1
    if pixel_vote_thr < min_cluster:
        cluster_iou_thr = 0
    src, boxes, starts, runs = [], [], [], []
    for ti, tr in enumerate(trackers):
        for a in tr.instances.values():
            src.append(ti)
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    src, boxes = np.array(src), np.array(boxes)
    if len(boxes) == 0:
        return {}
    G = _Graph()
    for n in range(len(src)):
        G.add_node(n, box=boxes[n], starts=starts[n], runs=runs[n])
    for r1, r2 in bounding_box_screening(boxes, src):
        iou, inter = rle_iou(starts[r1], runs[r1], starts[r2], runs[r2], True)
        if iou > 0:
            G.add_edge(int(r1), int(r2), iou=iou, overlap=inter)
    instances, iid = {}, 1
    for comp in G.components():
        if len(comp) < min_cluster:
            continue
        CG = merge_clusters(create_graph_of_clusters(G.subgraph(comp), cluster_iou_thr))
        cid, ci = 1, {}
        for node in CG.node:
            cluster = list(CG.node[node]['cluster'])
            if len(cluster) < min_cluster:
                continue
            box = G.node[cluster[0]]['box']
            for n in cluster[1:]:
                box = merge_boxes(box, G.node[n]['box'])
            ranges = [np.stack([G.node[n]['starts'], G.node[n]['starts'] + G.node[n]['runs']], axis=1) for n in cluster]
            voted = vote_by_ranges(ranges, pixel_vote_thr)
            if len(voted) > 0:
                ci[cid] = {'box': tuple(int(x) for x in box), 'starts': voted[:, 0], 'runs': voted[:, 1] - voted[:, 0]}
                cid += 1
        for a in merge_overlapping(ci):
            instances[iid] = a
            iid += 1
    return instances


def merge_semantic_from_trackers(trackers, pixel_vote_thr=2):
    """consensus.py:289-346."""
    boxes, starts, runs = [], [], []
    for tr in trackers:
        assert len(tr.instances) <= 1
        for a in tr.instances.values():
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    if not boxes:
        return {}
    box = boxes[0]
    for b in boxes[1:]:
        box = merge_boxes(box, b)
    v = vote_by_ranges([np.stack([s, s + r], axis=1) for s, r in zip(starts, runs)], pixel_vote_thr)
    return {1: {'box': box, 'starts': v[:, 0], 'runs': v[:, 1] - v[:, 0]}}


# ----------------------------------------------------------------------------
# 2-D tiled inference (inference/tile.py:8-195, consensus.py:471-625)
# ----------------------------------------------------------------------------
def tile_ranges_1d(length, tile, overlap):
    """Stand-in for cztile's AlmostEqualBorderFixedTotalAreaStrategy2D along one axis (cztile >= 2.0.0 is an
    un-vendored dependency, tile.py:2,88-104): PARITY UNPINNED.  Same contract -- every tile has the full size
    `tile`, neighbours overlap by at least `overlap`, overlaps almost equal: n = ceil((L - ov) / (T - ov)) tiles
    whose starts are floor(i * (L - T) / (n - 1))."""
    tile = min(tile, length)
    if length <= tile:
        return [(0, length)]
    n = -(-(length - overlap) // (tile - overlap))
    return [((i * (length - tile)) // (n - 1), (i * (length - tile)) // (n - 1) + tile) for i in range(n)]


def tile_ranges_2d(shape, tile_size, overlap):
    """Row-major list of (yrange, xrange) tiles (order unpinned, see tile_ranges_1d)."""
    th, tw = (tile_size, tile_size) if isinstance(tile_size, int) else tile_size
    ys, xs = tile_ranges_1d(shape[0], th, overlap), tile_ranges_1d(shape[1], tw, overlap)
    return [y for y in ys for _ in xs], [x for _ in ys for x in xs]


def calculate_overlap_rle(yranges, xranges, image_shape):
    """tile.py:8-52."""
    y = np.array(rle_voting(np.unique(np.stack(yranges, axis=0), axis=0), vote_thr=2))
    x = np.array(rle_voting(np.unique(np.stack(xranges, axis=0), axis=0), vote_thr=2))
    if len(y) > 0:
        row_starts = y[:, 0] * image_shape[1]
        row_runs = y[:, 1] * image_shape[1] - row_starts
    else:
        row_starts, row_runs = [], []
    if len(x) > 0:
        col = np.concatenate([x + r * image_shape[1] for r in range(image_shape[0])], axis=0)
        col_starts, col_runs = col[:, 0], col[:, 1] - col[:, 0]
    else:
        col_starts, col_runs = [], []
    if len(row_starts) > 0 or len(col_starts) > 0:
        return merge_rles(row_starts, row_runs, col_starts, col_runs)
    return [], []


def translate_rle_seg(rle_seg, yrange, xrange, image_shape):
    """Tiler.translate_rle_seg, tile.py:133-172 (in place)."""
    ys, xs = yrange[0], xrange[0]
    w = xrange[1] - xrange[0]
    for labels in rle_seg.values():
        for a in labels.values():
            b = list(a['box'])
            a['box'] = (b[0] + ys, b[1] + xs, b[2] + ys, b[3] + xs)
            st = a['starts']
            a['starts'] = np.ravel_multi_index((st // w + ys, st % w + xs), dims=image_shape)
    return rle_seg


def merge_semantic_from_tiles(tiles):
    """consensus.py:471-521."""
    label_id, boxes, starts, runs = None, [], [], []
    for t in tiles:
        for iid, a in t.items():
            if label_id is None:
                label_id = iid
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    if len(boxes) == 0:
        return {}
    box = np.array(boxes)[0]
    for b in np.array(boxes)[1:]:
        box = merge_boxes(box, b)
    r = join_ranges([np.stack([s, s + u], axis=1) for s, u in zip(starts, runs)])
    return {label_id: {'box': tuple(int(v) for v in box), 'starts': r[:, 0], 'runs': r[:, 1] - r[:, 0]}}


def merge_objects_from_tiles(tiles, overlap_rle=None):
    """consensus.py:523-625."""
    src, labels, boxes, starts, runs = [], [], [], [], []
    for ti, t in enumerate(tiles):
        for iid, a in t.items():
            src.append(ti)
            labels.append(int(iid))
            boxes.append(a['box'])
            starts.append(a['starts'])
            runs.append(a['runs'])
    src, labels, boxes = np.array(src), np.array(labels), np.array(boxes)
    if len(boxes) == 0:
        return {}
    G = _Graph()
    for n in range(len(labels)):
        G.add_node(n, box=boxes[n], starts=starts[n], runs=runs[n])
    for r1, r2 in bounding_box_screening(boxes, src):
        iou, inter = rle_iou(starts[r1], runs[r1], starts[r2], runs[r2], True)
        if iou > 0:
            G.add_edge(int(r1), int(r2), iou=iou, overlap=inter)
    iid = int(np.min(labels))
    out = {}
    for comp in G.components():
        cluster = list(comp)
        box = G.node[cluster[0]]['box']
        for n in cluster[1:]:
            box = merge_boxes(box, G.node[n]['box'])
        voted = join_ranges([np.stack([G.node[n]['starts'], G.node[n]['starts'] + G.node[n]['runs']], axis=1)
                             for n in cluster])
        if overlap_rle is not None and len(cluster) < 2 and np.any(voted):
            ioa = rle_ioa(overlap_rle[0], overlap_rle[1], voted[:, 0], voted[:, 1] - voted[:, 0])
            if ioa > 0.1:
                voted = []
        if np.any(voted):
            out[iid] = {'box': tuple(int(v) for v in box), 'starts': voted[:, 0], 'runs': voted[:, 1] - voted[:, 0]}
            iid += 1
    return out
