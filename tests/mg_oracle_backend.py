"""Per-rank arithmetic for the CPU (gloo) tests of the multi-GPU slab pipeline: the oracle's numpy restatements
behind the backend interface of ``empanada_napari_amd.multigpu.slab_stack_inference`` (test infrastructure: the product's
``HipSlabBackend`` is the only backend the package ships).  Head tensors come from tests/golden/median3d.npz (outputs of
the imported reference model stand-in), so the result is comparable with the reference's 3-D engine trace."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlabBackend:
    def __init__(self, golden, ks_engine_kwargs):
        from oracle import postprocess as opp
        self.g = golden
        self.eng = opp.RenderEngine(None, [1], label_divisor=ks_engine_kwargs['label_divisor'], stuff_area=ks_engine_kwargs['stuff_area'],
                                    nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5, coarse_boundaries=True)
        self.div = ks_engine_kwargs['label_divisor']

    def forward(self, lo, hi, n_ahead):
        from oracle import postprocess as opp
        g = self.g
        rows = [torch.from_numpy(opp.logits_to_prob(g['sem_logits'][z])[0]) for z in range(lo, hi)]
        sem = torch.zeros((hi - lo + n_ahead,) + tuple(rows[0].shape), dtype=torch.float32)
        sem[:hi - lo] = torch.stack(rows)
        stash = [(g['ctr_hmp'][z], g['offsets'][z]) for z in range(lo, hi)]
        return sem, stash

    def median_inplace(self, sem, n_own, hist, n_ahead, first, last, ks):
        from empanada_napari_amd import multigpu
        med = lambda maps: torch.from_numpy(np.sort(np.stack([m.numpy() for m in maps]), axis=0)[(len(maps) - 1) // 2])
        raw = [sem[i].clone() for i in range(n_own)]
        nxt = [sem[n_own + i].clone() for i in range(n_ahead)] or None
        out = multigpu.filtered_stack(raw, ks, med, None if hist is None else list(hist.unbind(0)), nxt, first, last)
        for i, f in enumerate(out):
            sem[i] = f

    def runs(self, sem, stash):
        from oracle import sparse as osp
        out = []
        for i, (ctr, off) in enumerate(stash):
            cells = self.eng.cells(ctr, off, 1)
            pan = self.eng.postprocess(sem[i].numpy()[None], cells)[0]
            out.append(osp.pan_seg_to_rle_seg(pan, [1], self.div, [1], force_connected=True))
        return out


def oracle_backend_factory(model_config, engine_kwargs, rank):
    """``backend_factory`` of MultiGPUEngine3d: runs inside each spawned rank process (no GPU anywhere)."""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    g = np.load(model_config['golden'])
    be = OracleSlabBackend(g, engine_kwargs)
    return lambda volume, axis: be


class LabelStackBackend:
    """A multi-class stack of per-slice label entries (tests/test_slab_matcher._stack) behind the backend interface: the
    network / median stages are dummies, ``runs`` hands out the entries of the rank's slab.  Exercises the slab matcher's
    gloo messages (ghost slices, forward / backward states of several classes, partial trackers)."""

    def __init__(self, shape, axis, seed):
        import test_slab_matcher as tsm
        self.entries = tsm._stack(shape, axis, seed)

    def forward(self, lo, hi, n_ahead):
        # every "probability map" carries its slice index; the rows the exchange has to fill start out as NaN
        sem = torch.full((hi - lo + n_ahead, 1, 4, 4), float('nan'))
        sem[:hi - lo] = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1)
        return sem, (lo, hi)

    def median_inplace(self, sem, n_own, hist, n_ahead, first, last, ks):
        """no filtering -- but a full check of what the halo / carry exchange delivered (slab and block schedules alike):
        the look-ahead rows are the RAW first maps of the slices right behind the block, the carry is the FILTERED tail of
        the slices right before it (a filtered map is marked + 0.5), each exactly once and in order"""
        mid = (ks - 1) // 2
        lo = int(sem[0, 0, 0, 0])
        assert torch.equal(sem[:n_own, 0, 0, 0], torch.arange(lo, lo + n_own, dtype=torch.float32)), 'own rows disturbed'
        assert first == (lo == 0) and (hist is None) == (first or mid == 0), (first, lo, hist is None)
        assert n_ahead == (0 if last else mid)
        if n_ahead:
            want = torch.arange(lo + n_own, lo + n_own + n_ahead, dtype=torch.float32)
            assert torch.equal(sem[n_own:, 0, 0, 0], want), ('look-ahead', lo, n_own, sem[n_own:, 0, 0, 0].tolist())
            assert bool((sem[n_own:] == want.view(-1, 1, 1, 1)).all())
        if hist is not None:
            want = torch.arange(lo - mid, lo, dtype=torch.float32) + 0.5
            assert torch.equal(hist[:, 0, 0, 0], want), ('carry', lo, hist[:, 0, 0, 0].tolist())
        sem[:n_own] += 0.5

    def runs(self, sem, stash):
        lo, hi = stash
        return self.entries[lo:hi]


def label_stack_backend_factory(model_config, engine_kwargs, rank):
    import sys
    for d in (ROOT, os.path.join(ROOT, 'tests')):
        if d not in sys.path:
            sys.path.insert(0, d)
    import __graft_entry__ as graft
    graft.load_package()
    cache = {}

    def make(volume, axis):
        if axis not in cache:
            cache[axis] = LabelStackBackend(tuple(int(v) for v in volume.shape), axis, model_config['seed'] + axis)
        return cache[axis]
    return make
