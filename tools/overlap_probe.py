"""Does the post-processing of batch k hide behind the forward of batch k + 1?  (two streams; the headline step of bench.py
run back to back / pipelined by one batch).  python tools/overlap_probe.py [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine, logits_to_prob  # noqa: E402
from empanada_napari_amd.preprocess import normalize_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    eng = PanopticDeepLabRenderEngine(model, thing_list=[1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)
    B = a.batch
    tiles = torch.from_numpy(synth.em_tiles(B, 1024, seed=1234))[:, None].to(dev)
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    model.reserve(B, 1024, 1024)

    def post(o):
        sem = logits_to_prob(o['sem_logits'])
        cells, _, _, kmax = eng.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
        return eng.panoptic_merge_int(sem, cells, kmax)

    def serial(n):
        outs = None
        for _ in range(n):
            o = model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
            outs = post(o)
        return outs

    def piped(n, fwd_stream, post_stream):
        pend = None
        outs = None
        for _ in range(n):
            with torch.cuda.stream(fwd_stream):
                o = model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
                ev = torch.cuda.Event()
                ev.record(fwd_stream)
            if pend is not None:
                with torch.cuda.stream(post_stream):
                    post_stream.wait_event(pend[1])
                    outs = post(pend[0])
            pend = (o, ev)
        with torch.cuda.stream(post_stream):
            post_stream.wait_event(pend[1])
            outs = post(pend[0])
        return outs

    res = {}
    ref = serial(3)
    torch.cuda.synchronize()
    for name, fn in (('serial', lambda n: serial(n)),
                     ('piped_same_prio', lambda n: piped(n, torch.cuda.Stream(dev), torch.cuda.Stream(dev))),
                     ('piped_fwd_high', lambda n: piped(n, torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev, priority=0))),
                     ('serial_again', lambda n: serial(n))):
        got = fn(3)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), name
        model.profile(True)
        t0 = time.perf_counter()
        fn(a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, fl, nl = model.profile_read()
        model.profile(False)
        res[name] = {'ms_per_step': round(dt * 1e3 / a.steps, 3), 'tiles_per_s': round(B * a.steps / dt, 1),
                     'dominant_TFLOPs': round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None, 'dominant_launches': nl}
        print(name, res[name], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'overlap_probe.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
