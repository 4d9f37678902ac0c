"""fp16 error budget of the Panoptic-DeepLab forward (CPU, oracle only).

For every place where the HIP engine rounds to fp16 -- the weights of one layer, or the output map of one
layer -- the oracle forward is run with ONLY that rounding switched on (oracle.pdl_model.Fp16Emu) and compared
with the plain fp32 forward: the table says how much of the head error each site explains (independent
roundings add in quadrature), then what the groups and the candidate mitigations (fp16 hi+lo weight pairs on
some layers, VERDICT's proposal) leave.  Writes a CSV (profiles/r02_error_budget.csv).

    python tools/error_budget.py [--size 256] [--out profiles/r02_error_budget.csv]
"""
import argparse
import csv
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
from empanada_napari_amd.preprocess import normalize  # noqa: E402
from oracle import pdl_model  # noqa: E402
from oracle.pdl_model import Fp16Emu  # noqa: E402


def heads(P, x, cfg, emu):
    taps = {}
    o = pdl_model.model_forward(P, x, cfg, 2, False, taps, emu)
    return {'ctr': o['ctr_hmp'], 'off': o['offsets'], 'sem_coarse': taps['sem_coarse'],
            'prob': torch.sigmoid(o['sem_logits']) if o['sem_logits'].shape[1] == 1 else torch.softmax(o['sem_logits'], 1)}


def errs(a, b):
    out = {}
    for k in ('ctr', 'off', 'sem_coarse', 'prob'):
        d = (a[k] - b[k]).abs()
        if k in ('ctr', 'off'):
            out[k + '_rms_rel'] = float(d.pow(2).mean().sqrt()) / max(1.0, float(b[k].pow(2).mean().sqrt()))
        out[k + '_max'] = float(d.max())
        out[k + '_rms'] = float(d.pow(2).mean().sqrt())
    # PointRend refines the 8192 most uncertain cells: probabilities away from selection flips
    d = (a['prob'] - b['prob']).abs()
    out['prob_p999'] = float(torch.quantile(d.flatten()[:: max(1, d.numel() // 1_000_000)], 0.999))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_error_budget.csv'))
    ap.add_argument('--sites', type=int, default=1, help='0: groups and mitigations only')
    ap.add_argument('--arch', default='pdl', help="'pdl' (MitoNet class) or 'bifpn' (MitoNet_mini class, SURVEY row a5)")
    ap.add_argument('--classes', type=int, default=1)
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))      # oneDNN convs stop scaling (and oversubscribe) beyond ~32 threads
    if args.arch == 'bifpn':
        cfg = dict(weights.MITONET_MINI_CFG, num_classes=args.classes)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
    else:
        cfg = dict(weights.MITONET_PDL_CFG)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    img = synth.em_tiles(1, args.size, seed=5)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    t0 = time.time()
    ref = heads(P, x, cfg, None)
    print(f'fp32 forward {time.time() - t0:.1f} s; |ctr| rms {float(ref["ctr"].pow(2).mean().sqrt()):.3f} '
          f'|off| rms {float(ref["off"].pow(2).mean().sqrt()):.3f} |sem_coarse| rms {float(ref["sem_coarse"].pow(2).mean().sqrt()):.3f}')
    probe = pdl_model.engine_emu(P, cfg)      # the engine's formats, hi + lo weight pairs and fused maps included
    full = heads(P, x, cfg, probe)
    rows = []

    def add(kind, site, emu):
        e = errs(heads(P, x, cfg, emu), ref)
        rows.append(dict(kind=kind, site=site, **e))
        print(f'{kind:10s} {site:60s} ctr rel {e["ctr_rms_rel"]:.2e} off rel {e["off_rms_rel"]:.2e} | ctr max {e["ctr_max"]:.2e} rms {e["ctr_rms"]:.2e} | off max {e["off_max"]:.2e} | '
              f'sem max {e["sem_coarse_max"]:.2e} rms {e["sem_coarse_rms"]:.2e} | prob max {e["prob_max"]:.2e} p99.9 {e["prob_p999"]:.2e}',
              flush=True)
        return e

    e = errs(full, ref)
    rows.append(dict(kind='group', site='engine formats (all weights + all activations fp16)', **e))
    print('ALL', e)
    add('group', 'all weights fp16, activations fp32', Fp16Emu(True, False))
    add('group', 'all activations fp16, weights fp32', Fp16Emu(False, True))
    W, A = list(probe.sites_w), list(probe.sites_a)
    enc_w = {w for w in W if w.startswith('encoder.')}
    enc_a = {a for a in A if a.startswith('encoder.')}
    add('group', 'encoder weights fp16 only', Fp16Emu(enc_w, False))
    add('group', 'encoder activations fp16 only', Fp16Emu(False, enc_a))
    add('group', 'decoder+head weights fp16 only', Fp16Emu(set(W) - enc_w, False))
    add('group', 'decoder+head activations fp16 only', Fp16Emu(False, set(A) - enc_a))
    # candidate mitigations: hi+lo weight pairs (2 MFMAs per product on those layers)
    heads_dec = [w for w in W if ('head' in w or '.fuse.' in w or 'aspp.project' in w) and 'sepconv.0' not in w]
    add('mitigate', 'hi+lo weights: heads pw, fuse pw, ASPP project (VERDICT)', Fp16Emu(True, True, heads_dec))
    add('mitigate', 'hi+lo weights: whole decoder + heads', Fp16Emu(True, True, set(W) - enc_w))
    add('mitigate', 'hi+lo weights: every layer (2x MFMA everywhere)', Fp16Emu(True, True, set(W)))
    add('mitigate', 'hi+lo weights everywhere + fp32 residual stream (block outputs)', Fp16Emu(True, set(A) - {a for a in A if a.count('.') == 2 and a.startswith('encoder.layer')}, set(W)))
    if args.sites:
        for w in W:
            add('weight', w, Fp16Emu({w}, False))
        for a in A:
            add('act', a, Fp16Emu(False, {a}))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w', newline='') as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in rows:
            wr.writerow({k: (f'{v:.4e}' if isinstance(v, float) else v) for k, v in r.items()})
    print('wrote', args.out)


if __name__ == '__main__':
    main()
