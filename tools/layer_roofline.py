"""Per-layer roofline table of the MFMA launches (implicit-GEMM convs and fused separable convs) of one forward
at the bench shape.
  run  : EMP_LAYER_LOG=<log> rocprofv3 --kernel-trace ... -- python3 tools/layer_roofline.py run [batch] [size]
  join : python tools/layer_roofline.py join <trace_dir> <log> [out.csv]
Algorithmic bytes per launch = input + output (+ residual) activations + weights, fp16."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NFWD = 3


def run(B, S):
    import numpy as np
    import torch
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize_params
    dev = torch.device('cuda:0')
    cfg = dict(weights.MITONET_MINI_CFG if os.environ.get('EMP_MODEL') == 'bifpn' else weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    base = synth.em_tiles(min(B, 4), S, seed=1234)
    tiles = torch.from_numpy(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B])[:, None].to(dev)
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    model.reserve(B, S, S)
    for _ in range(NFWD):
        model(tiles, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
    torch.cuda.synchronize()


def join(trace_dir, log, out=None):
    f = glob.glob(os.path.join(trace_dir, '**', '*kernel_trace.csv'), recursive=True)[0]
    ks = [r for r in csv.DictReader(open(f)) if ('conv_igemm' in r['Kernel_Name'] or 'sepconv5' in r['Kernel_Name'] or 'sepconvp_kernel' in r['Kernel_Name'] or 'conv3x3_c' in r['Kernel_Name']) and 'pack' not in r['Kernel_Name']]
    ks.sort(key=lambda r: int(r['Start_Timestamp']))
    fw, cur = [], []
    for line in open(log):
        line = line.strip()
        if line == 'end':
            fw.append(cur)
            cur = []
        elif line:
            cur.append(line.split(','))
    layers = fw[-1]
    assert len(ks) == len(layers) * len(fw), (len(ks), len(layers), len(fw))
    ks = ks[-len(layers):]
    rows = ['kind,layer,M,Cin,Cout,k,stride,dil,us,TFLOPs,alg_GBps,ideal_us,ratio']
    tot = tot_ideal = 0.0
    for lay, k in zip(layers, ks):
        kind, name, M, cin, cout, kh, stride, dil, res, inpix = lay[0], lay[1], *[int(v) for v in lay[2:]]
        us = (int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3
        if kind == 'conv':
            flops = 2.0 * M * cout * cin * kh * kh
            by = 2.0 * (inpix * cin + M * cout * (2 if res else 1) + cout * cin * kh * kh)
        elif kind in ('sepconv', 'sepconvp'):
            flops = 2.0 * M * cin * (25 + cout)
            by = 2.0 * (M * cin + M * cout)
        else:
            flops = 2.0 * M * cin * (25 + cout)
            by = 2.0 * M * cin
        ideal = max(flops / 2.5e15, by / 6.3e12) * 1e6
        tot += us
        tot_ideal += ideal
        rows.append(f'{kind},{name},{M},{cin},{cout},{kh},{stride},{dil},{us:.1f},{flops / us / 1e6:.0f},{by / us / 1e3:.0f},'
                    f'{ideal:.1f},{us / ideal:.2f}')
    rows.append(f'total,,,,,,,,{tot:.1f},,,{tot_ideal:.1f},{tot / tot_ideal:.2f}')
    txt = '\n'.join(rows)
    print(txt)
    if out:
        open(out, 'w').write('# ideal_us = max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s)\n' + txt + '\n')


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 1024)
    else:
        join(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
