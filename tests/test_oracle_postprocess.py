"""numpy oracle of the instance post-processing vs golden vectors produced by
the imported reference (tests/golden/postprocess.npz, median3d.npz).  Integer
outputs must be bit-exact."""
import os

import numpy as np
import pytest

from empanada_napari_amd import synth
from oracle import postprocess as opp

NSPEC = 9


def _case(g, i):
    H, W, n, coarse, ncls, k = [int(v) for v in g[f'{i}_spec']]
    thr = float(g[f'{i}_thr'])
    plateau = i in (3, 4, 6)
    sem, ctr, off = synth.head_outputs(H, W, n, seed=100 + i, coarse=bool(coarse), num_classes=ncls, plateau=plateau)
    return H, W, n, bool(coarse), ncls, k, thr, sem, ctr, off


@pytest.mark.parametrize('i', range(NSPEC))
def test_centers_groups_cells(golden_dir, i):
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W, n, coarse, ncls, k, thr, sem, ctr, off = _case(g, i)
    centers = opp.find_instance_center(ctr, thr, k)
    np.testing.assert_array_equal(centers, g[f'{i}_centers'])
    if centers.shape[0]:
        grp = opp.group_pixels(centers, off, step=4 if coarse else 1)
        np.testing.assert_array_equal(grp, g[f'{i}_groups'])
    cells = opp.get_instance_cells(ctr, off, thr, k, coarse, 1)
    np.testing.assert_array_equal(cells.astype(np.int32), g[f'{i}_cells'])


@pytest.mark.parametrize('i', range(NSPEC))
@pytest.mark.parametrize('divisor,conf', [(1000, 0.5), (10000, 0.3)])
def test_render_engine(golden_dir, i, divisor, conf):
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W, n, coarse, ncls, k, thr, sem, ctr, off = _case(g, i)
    model = lambda x, rs, interp: {'sem_logits': sem, 'ctr_hmp': ctr, 'offsets': off}
    eng = opp.RenderEngine(model, [1] if ncls == 1 else [1, 2], label_divisor=divisor, nms_threshold=thr,
                           nms_kernel=k, confidence_thr=conf, coarse_boundaries=coarse)
    pan = eng(np.zeros((1, 1, H, W), np.float32), (H - 3, W - 5), 1)
    ref = g[f'{i}_pan_{divisor}']
    assert pan.shape == ref.shape
    np.testing.assert_array_equal(pan, ref)


def test_case_coverage(golden_dir):
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    ks = [g[f'{i}_centers'].shape[0] for i in range(NSPEC)]
    assert 0 in ks and max(ks) > 20 and any(0 < k <= 20 for k in ks)  # both grouping branches + empty


@pytest.mark.parametrize('ks', [1, 3, 5, 7])
def test_engine3d_trace(golden_dir, ks):
    g = np.load(os.path.join(golden_dir, 'median3d.npz'))
    n = g['sem_logits'].shape[0]
    it = iter(range(n))

    def model(x, rs, interp):
        z = next(it)
        return {'sem_logits': g['sem_logits'][z], 'ctr_hmp': g['ctr_hmp'][z], 'offsets': g['offsets'][z]}

    eng = opp.RenderEngine3d(model, [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3,
                             confidence_thr=0.5, coarse_boundaries=True, median_kernel_size=ks)
    segs = []
    for z in range(n):
        r = eng(np.zeros((1, 1, 64, 64), np.float32), (64, 64), 1)
        if r is not None:
            segs.append(r)
    segs += eng.end(1)
    got = np.stack(segs).astype(np.int32)
    np.testing.assert_array_equal(got, g[f'pan_ks{ks}'])


def test_recursive_median_scalar(golden_dir):
    g = np.load(os.path.join(golden_dir, 'median3d.npz'))
    q = opp.MedianQueue(3)
    out = []
    for v in g['scalar_in']:
        q.enqueue({'sem': np.array([[v]], np.float32)})
        o = q.get_next(['sem'])
        if o is not None:
            out.append(float(o['sem']))
    out += [float(o['sem']) for o in list(q.median_queue)[q.mid_idx + 1:]]
    np.testing.assert_array_equal(np.array(out, np.float32), g['scalar_out'])
    assert list(g['scalar_out']) == [5, 5, 5, 5, 5, 2]  # recursive, not sliding
