"""A/B of EMP_PRECISE_SEPCONV on the PanopticBiFPNPR engine: end-to-end distance of the centre heat-map and
the offsets to the fp32 oracle forward at 512^2 (1 and 4 classes) and to the reference goldens at their sizes, plus the
forward time at batch 1 (1024^2) and batch 32 -- the numbers behind the default in pdl_net.hip (`precise_node`).

    python tools/bifpn_parity_ab.py [--modes 7,1,5,6,2] [--bench 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from empanada_napari_amd import synth, weights  # noqa: E402
from empanada_napari_amd.engines import HipPanopticDeepLab  # noqa: E402
from empanada_napari_amd.preprocess import normalize  # noqa: E402
from oracle import pdl_model  # noqa: E402


def rel(got, ref):
    d = (got - ref).abs()
    scale = float(ref.pow(2).mean().sqrt())
    return float(d.pow(2).mean().sqrt()) / max(1.0, scale), float(d.max()) / max(1.0, scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--modes', default='7,1n,1f,1,6', help="EMP_PRECISE_SEPCONV values; a trailing 'n' switches the hi + lo weight pairs off (EMP_PRECISE_WSPLIT=0), 'f' only the hi + lo fused maps (EMP_PRECISE_FSPLIT=0)")
    ap.add_argument('--bench', type=int, default=1)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'bifpn_parity_ab.json'))
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'bifpn_forward.npz'))
    res = {}
    refs = {}
    for ncls in (1, 4):
        cfg = dict(weights.MITONET_MINI_CFG, num_classes=ncls)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
        x = torch.from_numpy(normalize(synth.em_tiles(1, 512, seed=77), 0.57571, 0.12765))[:, None]
        refs[ncls] = (cfg, P, x, pdl_model.bifpn_forward(P, x, cfg, 2, False))
    for mode in a.modes.split(','):
        os.environ['EMP_PRECISE_SEPCONV'] = mode.rstrip('nf')
        os.environ['EMP_PRECISE_WSPLIT'] = '0' if mode.endswith('n') else '1'
        os.environ['EMP_PRECISE_FSPLIT'] = '0' if mode.endswith('f') else '1'
        row = {}
        for ncls in (1, 4):
            cfg, P, x, ref = refs[ncls]
            model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
            out = {k: v.cpu() for k, v in model(x.cuda(), 2, False).items()}
            for k in ('ctr_hmp', 'offsets'):
                row[f'512_ncls{ncls}_{k}_rms_rel'], row[f'512_ncls{ncls}_{k}_max_rel'] = rel(out[k], ref[k])
            tag = 'm1' if ncls == 1 else 'm4'
            for case in ('a', 'b'):
                xi = torch.from_numpy(normalize(g[f'{tag}{case}_image'], 0.57571, 0.12765))
                xi = xi.reshape((-1, 1) + tuple(xi.shape[-2:]))
                o = model(xi.cuda(), int(g[f'{tag}{case}_render_steps']), bool(g[f'{tag}{case}_interpolate_ins']))
                for k in ('ctr_hmp', 'offsets'):
                    got, want = o[k].cpu().numpy(), g[f'{tag}{case}_{k}']
                    scale = float(np.abs(want).mean()) + 1e-6
                    row[f'golden_{tag}{case}_{k}_rms_over_mean'] = float(np.sqrt(((got - want) ** 2).mean())) / scale
            if a.bench and ncls == 1:
                for B, reps in ((1, 20), (32, 5)):
                    xb = torch.from_numpy(synth.em_tiles(B, 1024, seed=5))[:, None].cuda()
                    sub, mul = 0.57571 * 255, 1.0 / (0.12765 * 255)
                    for _ in range(2):
                        model(xb, 2, False, sub=sub, mul=mul)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        model(xb, 2, False, sub=sub, mul=mul)
                    torch.cuda.synchronize()
                    row[f'forward_ms_b{B}_1024'] = (time.perf_counter() - t0) / reps * 1e3
            del model
        res[mode] = row
        print('mode', mode, json.dumps(row, indent=1), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
