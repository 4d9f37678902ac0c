"""HIP instance post-processing vs the numpy oracle and the reference golden
vectors: label maps must be bit-exact given identical fp32 head tensors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NSPEC = 9


class _Fake:
    """model stand-in returning canned head tensors (like the golden generator)."""

    def __init__(self, outs):
        self.outs = outs if isinstance(outs, list) else [outs]
        self.i = 0
        self._p = torch.zeros(1, device='cuda')

    def eval(self):
        return self

    def parameters(self):
        yield self._p

    def __call__(self, x, render_steps=2, interpolate_ins=True):
        o = self.outs[min(self.i, len(self.outs) - 1)]
        self.i += 1
        return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in o.items()}


def _case(g, i):
    from empanada_napari_amd import synth
    H, W, n, coarse, ncls, k = [int(v) for v in g[f'{i}_spec']]
    thr = float(g[f'{i}_thr'])
    sem, ctr, off = synth.head_outputs(H, W, n, seed=100 + i, coarse=bool(coarse), num_classes=ncls,
                                       plateau=i in (3, 4, 6))
    return H, W, bool(coarse), ncls, k, thr, sem, ctr, off


@pytest.mark.parametrize('i', range(NSPEC))
def test_cells_and_centers_match_reference(golden_dir, i):
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W, coarse, ncls, k, thr, sem, ctr, off = _case(g, i)
    eng = PanopticDeepLabRenderEngine(_Fake({}), [1], nms_threshold=thr, nms_kernel=k, coarse_boundaries=coarse)
    cells, centers, num, kmax = eng.instance_cells_int(torch.from_numpy(ctr).cuda(), torch.from_numpy(off).cuda(), 1)
    K = g[f'{i}_centers'].shape[0]
    assert int(num[0]) == K == kmax
    np.testing.assert_array_equal(centers[0, :K].cpu().numpy(), g[f'{i}_centers'])
    np.testing.assert_array_equal(cells.cpu().numpy()[:, None], g[f'{i}_cells'])
    f = eng.get_instance_cells(torch.from_numpy(ctr).cuda(), torch.from_numpy(off).cuda(), 1)
    assert f.dtype == torch.float32 and tuple(f.shape) == g[f'{i}_cells'].shape


@pytest.mark.parametrize('i', range(NSPEC))
@pytest.mark.parametrize('divisor,conf', [(1000, 0.5), (10000, 0.3)])
def test_render_engine_matches_reference(golden_dir, i, divisor, conf):
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W, coarse, ncls, k, thr, sem, ctr, off = _case(g, i)
    eng = PanopticDeepLabRenderEngine(_Fake({'sem_logits': sem, 'ctr_hmp': ctr, 'offsets': off}),
                                      [1] if ncls == 1 else [1, 2], label_divisor=divisor, nms_threshold=thr,
                                      nms_kernel=k, confidence_thr=conf, coarse_boundaries=coarse)
    pan = eng(torch.zeros(1, 1, H, W), (H - 3, W - 5), 1)
    ref = g[f'{i}_pan_{divisor}']
    assert pan.dtype == torch.int64 and tuple(pan.shape) == ref.shape
    got = pan.cpu().numpy()
    if not np.array_equal(got, ref):
        # the only legitimate source of a flip is sigmoid/softmax rounding exactly at the threshold
        from oracle import postprocess as opp
        p = opp.logits_to_prob(sem)[..., :H - 3, :W - 5]
        near = np.abs(p - conf).min(axis=1) < 1e-6 if ncls == 1 else np.zeros(got.shape, bool)
        assert np.all((got == ref) | near), f'{np.sum(got != ref)} label flips away from the threshold'


def test_batched_postprocess_equals_per_image():
    """N images per launch group == N single-image calls (oracle on each image)."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    from oracle import postprocess as opp
    H = W = 256
    outs = [synth.head_outputs(H, W, n, seed=500 + n, coarse=True) for n in (0, 7, 33, 150)]
    sem = np.concatenate([o[0] for o in outs])
    ctr = np.concatenate([o[1] for o in outs])
    off = np.concatenate([o[2] for o in outs])
    eng = PanopticDeepLabRenderEngine(_Fake({}), [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, coarse_boundaries=True)
    prob = opp.logits_to_prob(sem)
    cells, centers, num, kmax = eng.instance_cells_int(torch.from_numpy(ctr).cuda(), torch.from_numpy(off).cuda(), 1)
    pan = eng.panoptic_merge_int(torch.from_numpy(prob).cuda(), cells, kmax).cpu().numpy()
    for n in range(4):
        c = opp.get_instance_cells(ctr[n:n + 1], off[n:n + 1], 0.1, 3, True, 1)
        np.testing.assert_array_equal(cells[n].cpu().numpy(), c[0, 0].astype(np.int32))
        hard = opp.harden_seg(prob[n:n + 1], 0.5)[0]
        ref = opp.get_panoptic_seg(hard, c, [1], 10000, 64, 0)
        np.testing.assert_array_equal(pan[n], ref[0])


def test_logits_to_prob():
    from empanada_napari_amd.engines import logits_to_prob
    from oracle import postprocess as opp
    rng = np.random.default_rng(0)
    for C in (1, 4):
        x = (rng.standard_normal((2, C, 33, 47)) * 4).astype(np.float32)
        got = logits_to_prob(torch.from_numpy(x).cuda()).cpu().numpy()
        np.testing.assert_allclose(got, opp.logits_to_prob(x), atol=2e-7, rtol=2e-6)  # float domain: ~1 ulp


@pytest.mark.parametrize('ks', [1, 3, 5, 7])
def test_engine3d_trace_matches_reference(golden_dir, ks):
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine3d
    g = np.load(os.path.join(golden_dir, 'median3d.npz'))
    n = g['sem_logits'].shape[0]
    outs = [{'sem_logits': g['sem_logits'][z], 'ctr_hmp': g['ctr_hmp'][z], 'offsets': g['offsets'][z]} for z in range(n)]
    eng = PanopticDeepLabRenderEngine3d(_Fake(outs), [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3,
                                        confidence_thr=0.5, median_kernel_size=ks, coarse_boundaries=True)
    segs = []
    for z in range(n):
        r = eng(torch.zeros(1, 1, 64, 64), (64, 64), 1)
        if r is not None:
            segs.append(r.cpu().numpy())
    segs += [s.cpu().numpy() for s in eng.end(1)]
    got = np.stack(segs).astype(np.int32)
    ref = g[f'pan_ks{ks}']
    if not np.array_equal(got, ref):
        from oracle import postprocess as opp
        frac = np.mean(got != ref)
        assert frac < 1e-5, f'{frac} of labels differ'  # sigmoid rounding at the threshold only


def test_median_is_exact_selection():
    from empanada_napari_amd import _abi
    import ctypes as C
    lib = _abi.load()
    rng = np.random.default_rng(5)
    for ks in (1, 3, 5, 7, 9):
        x = rng.standard_normal((ks, 3, 50, 70)).astype(np.float32)
        x[:, 0, :5] = x[0, 0, :5]  # ties
        t = [torch.from_numpy(x[k]).cuda() for k in range(ks)]
        out = torch.empty_like(t[0])
        ptrs = (C.c_void_p * ks)(*[a.data_ptr() for a in t])
        _abi.check(lib.emp_median_slices(ptrs, ks, _abi.ptr(out), out.numel(), _abi.stream_ptr()), 'median')
        np.testing.assert_array_equal(out.cpu().numpy(), np.sort(x, axis=0)[(ks - 1) // 2])


@pytest.mark.parametrize('i', [0, 5, 8])
def test_harden_and_get_panoptic_seg_api(golden_dir, i):
    """engines.py:114-121,277-292: the two-step API (harden, then get_panoptic_seg on the class map) gives the label
    map of postprocess()."""
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    from oracle import postprocess as opp
    g = np.load(os.path.join(golden_dir, 'postprocess.npz'))
    H, W, coarse, ncls, k, thr, sem, ctr, off = _case(g, i)
    tl = [1] if ncls == 1 else [1, 2]
    eng = PanopticDeepLabRenderEngine(_Fake({}), tl, label_divisor=1000, nms_threshold=thr, nms_kernel=k,
                                      confidence_thr=0.5, coarse_boundaries=coarse)
    prob = torch.from_numpy(opp.logits_to_prob(sem)).cuda()
    hard = eng._harden_seg(prob)
    assert hard.dtype == torch.int64 and tuple(hard.shape) == (1, 1, H, W)
    np.testing.assert_array_equal(hard.cpu().numpy(), opp.harden_seg(opp.logits_to_prob(sem), 0.5))
    cells = eng.get_instance_cells(torch.from_numpy(ctr).cuda(), torch.from_numpy(off).cuda(), 1)
    a = eng.get_panoptic_seg(hard[0], cells)
    b = eng.postprocess(prob, cells)
    assert torch.equal(a, b) and a.dtype == torch.int64
    want = opp.get_panoptic_seg(opp.harden_seg(opp.logits_to_prob(sem), 0.5)[0], cells.cpu().numpy(), tl, 1000, 64, 0)
    np.testing.assert_array_equal(a.cpu().numpy(), want)


@pytest.mark.parametrize('h,w,step,dens', [(256, 256, 4, 0.02), (192, 320, 4, 0.2), (512, 384, 1, 0.004), (1024, 1024, 4, 0.004), (96, 96, 1, 0.9)])
def test_centre_grid_voting_equals_the_scan(h, w, step, dens, monkeypatch):
    """Round 6 (late): from 192 centres per image the nearest-centre vote searches a uniform grid over the centres ring by ring
    (ctr_grid_build_kernel + the grid path of group_pixels_kernel, csrc/postprocess.hip) instead of scanning every centre per pixel.
    It must return the scan's cells bit for bit -- same fp32 expression per candidate, lowest index among the minima of the ROUNDED
    distance, 1e5 start value -- including exact ties (integer votes: a pixel equidistant from several centres), votes far outside
    the image, NaN / inf votes and images of a batch below the threshold (which keep the scan).  EMP_VOTE_GRID=0 is the scan."""
    from gpu_common import dev
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    g = torch.Generator().manual_seed(h * 7 + w + step)
    N = 3
    ctr = torch.rand((N, 1, h, w), generator=g)
    keep = torch.rand((N, 1, h, w), generator=g) < dens      # density of candidate peaks
    ctr = torch.where(keep, ctr, torch.zeros(()))
    ctr[2] = 0.0
    ctr[2, 0, ::h // 9 + 1, ::w // 11 + 1] = 0.9      # image 2: <= 99 centres -> below the grid's threshold
    off = torch.randint(-60, 61, (N, 2, h, w), generator=g).float()      # integer votes: exact ties
    off[0, :, : h // 4] += torch.randn((2, h // 4, w), generator=g) * 3.0      # ... and generic ones
    off[1, 0, 5, 7] = float('nan')
    off[1, 1, 9, 3] = float('inf')
    off[1, 0, 11, 2] = -float('inf')
    off[0, :, -3:, :] = 5.0e4      # far outside the image, still below the 1e5 start value
    off[0, :, -1, :] = 3.0e5       # beyond it: no centre
    eng = PanopticDeepLabRenderEngine(_Fake({}), [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                                      coarse_boundaries=(step == 4))
    eng.MAX_CENTERS = 16384
    ctr, off = ctr.to(dev()), off.to(dev())
    monkeypatch.setenv('EMP_VOTE_GRID', '1')
    a, ca, na, _ = eng.instance_cells_int(ctr, off, 1)
    a, ca, na = a.clone(), ca.clone(), na.clone()
    monkeypatch.setenv('EMP_VOTE_GRID', '0')
    b, cb, nb_, _ = eng.instance_cells_int(ctr, off, 1)
    torch.cuda.synchronize()
    num = na.cpu().tolist()
    assert num == nb_.cpu().tolist() and num[0] >= 192 and num[1] >= 192 and num[2] < 192, num
    assert torch.equal(ca, cb)
    assert torch.equal(a, b), f'{int((a != b).sum())} cells differ (centres per image {num})'
    assert int(a[0].max()) > 100 and int((a[0] == 0).sum()) > 0      # both populated cells and "no centre" cells exist


def test_centre_grid_voting_random_sweep(monkeypatch):
    """24 random geometries (40 .. 700 pixels a side, steps 1 / 4, 0.3 % .. 90 % of the cells candidate peaks, one to three images) x four
    kinds of votes (integer: exact ties; gaussian at 0.5 .. 500 px; none; 3e4 px: mostly outside the 1e5 radius of nothing) -- the grid
    path's cells equal the scan's bit for bit in every one (60 such configurations were run when the kernel was written)"""
    from gpu_common import dev
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine
    rng = np.random.default_rng(123)
    for it in range(24):
        h, w = int(rng.integers(40, 700)), int(rng.integers(40, 700))
        step, dens = int(rng.choice([1, 4])), float(rng.choice([0.003, 0.01, 0.05, 0.3, 0.9]))
        N = int(rng.integers(1, 4))
        g = torch.Generator().manual_seed(it)
        ctr = torch.rand((N, 1, h, w), generator=g)
        ctr = torch.where(torch.rand((N, 1, h, w), generator=g) < dens, ctr, torch.zeros(()))
        mode = it % 4
        if mode == 0:
            off = torch.randint(-40, 41, (N, 2, h, w), generator=g).float()
        elif mode == 1:
            off = torch.randn((N, 2, h, w), generator=g) * float(rng.choice([0.5, 5, 50, 500]))
        elif mode == 2:
            off = torch.zeros((N, 2, h, w))
        else:
            off = torch.randn((N, 2, h, w), generator=g) * 3e4
        eng = PanopticDeepLabRenderEngine(_Fake({}), [1], label_divisor=1000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                                          coarse_boundaries=(step == 4))
        eng.MAX_CENTERS = 1 << 17
        c, o = ctr.to(dev()), off.to(dev())
        monkeypatch.setenv('EMP_VOTE_GRID', '1')
        a = eng.instance_cells_int(c, o, 1)[0].clone()
        monkeypatch.setenv('EMP_VOTE_GRID', '0')
        b = eng.instance_cells_int(c, o, 1)[0]
        assert torch.equal(a, b), (it, h, w, step, dens, mode, int((a != b).sum()))
