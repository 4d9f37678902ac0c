"""Layer-by-layer comparison of the HIP forward with the oracle (fp32 and fp16-format emulation) -- GPU tool.

Prints, per tap, the rms difference relative to the tap's rms against both oracles: the first layer where the
distance to the FORMAT oracle jumps to the distance to the fp32 oracle is a layer whose rounding points the emulation
(or the kernel) gets wrong.   python tools/layer_parity.py [size]"""
import os
import sys

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


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.set_num_threads(min(os.cpu_count() or 1, 32))      # oneDNN convs stop scaling (and oversubscribe) beyond ~32 threads
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
    img = synth.em_tiles(1, size, seed=2024)
    x = torch.from_numpy(normalize(img, 0.57571, 0.12765))[:, None]
    out = model(x.cuda(), 2, False)
    torch.cuda.synchronize()
    t32, t16 = {}, {}
    r32 = pdl_model.pdl_forward(P, x, cfg, 2, False, t32)
    r16 = pdl_model.pdl_forward(P, x, cfg, 2, False, t16, emu=pdl_model.Fp16Emu())
    t32['p1'] = torch.nn.functional.max_pool2d(t32['stem'], 3, 2, 1)
    t16['p1'] = pdl_model.Fp16Emu.r16(torch.nn.functional.max_pool2d(t16['stem'], 3, 2, 1))
    names = {'p1': 'p1'}
    for li, nb in enumerate((3, 4, 6, 3), start=1):
        for b in range(nb):
            for suf in ('.c1', '.c2', ''):
                names[f'encoder.layer{li}.{b}{suf}'] = f'encoder.layer{li}.{b}{suf}'
    for d in ('semantic_decoder', 'instance_decoder'):
        names[f'{d}.aspp.cat'] = f'{d}.aspp.cat'
        names[f'{d}.aspp'] = f'{d}.aspp'
        names[f'{d}.stage0.cat'] = f'{d}.stage0.cat'
    names['semantic_decoder.stage0.out'] = 'semantic_x'
    names['instance_decoder.stage0.out'] = 'instance_x'
    print(f'{"tap":36s} {"vs fp32":>10s} {"vs format":>10s}   (rms difference / rms of the tap)')
    for tap, oname in names.items():
        got = model.tap(tap).float().cpu().permute(0, 3, 1, 2)
        a, b = t32[oname], t16[oname]
        got = got[:, :a.shape[1]]
        sc = a.pow(2).mean().sqrt().item() + 1e-12
        e32 = (got - a).pow(2).mean().sqrt().item() / sc
        e16 = (got - b).pow(2).mean().sqrt().item() / sc
        nz = (got != b).float().mean().item()
        print(f'{tap:36s} {e32:10.3e} {e16:10.3e}   differing elements {nz:.4f}')
    for k in ('ctr_hmp', 'offsets'):
        g = out[k].cpu()
        print(f'{k:36s} max vs fp32 {float((g - r32[k]).abs().max()):.3e}  max vs format {float((g - r16[k]).abs().max()):.3e}')
    c = model.tap_raw('semantic_head.out', (1, 1, size // 4, size // 4)).cpu()
    print(f'{"sem_coarse":36s} max vs fp32 {float((c - t32["sem_coarse"]).abs().max()):.3e}  max vs format {float((c - t16["sem_coarse"]).abs().max()):.3e}')


if __name__ == '__main__':
    main()
