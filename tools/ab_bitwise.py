import os, sys, subprocess, json
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else 'driver'
VAR = os.environ.get('AB_VAR', 'EMP_FUSE_B2B')      # the A/B switch: AB_VAR=EMP_BILINEAR_NO_UP4 python tools/ab_bitwise.py
if mode == 'driver':
    outs = {}
    for v in ('0', '1'):
        env = dict(os.environ, **{VAR: v})
        subprocess.check_call([sys.executable, __file__, 'run', f'/tmp/b2b_{v}.npz'], env=env)
        outs[v] = np.load(f'/tmp/b2b_{v}.npz')
    bad = 0
    for k in outs['0'].files:
        a, b = outs['0'][k], outs['1'][k]
        same = np.array_equal(a, b)
        if not same: bad += 1
        print(f'{k:40s} identical={same}' + ('' if same else f'  max|d|={np.abs(a.astype(np.float64)-b.astype(np.float64)).max():.3e} frac={np.mean(a!=b):.3e}'))
    print('MISMATCHES', bad)
else:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as graft
    graft.load_package()
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.preprocess import normalize
    res = {}
    for name, cfg0 in (('pdl', weights.MITONET_PDL_CFG), ('bifpn', weights.MITONET_MINI_CFG)):
        cfg = dict(cfg0)
        P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=3), cfg)
        model = HipPanopticDeepLab(P, cfg, folded=True, precision=os.environ.get('EMP_TOOL_PRECISION', 'fp16'))
        for B, S in ((4, 1024), (1, 1024), (2, 512)):
            x = torch.from_numpy(normalize(synth.em_tiles(B, S, seed=5), 0.57571, 0.12765))[:, None].cuda()
            out = model(x, 2, False)
            for k, v in out.items():
                res[f'{name}.{B}x{S}.{k}'] = v.float().cpu().numpy()
            for t in ('encoder.layer1.1.c1', 'encoder.layer1.2.c1', 'encoder.layer2.0.c1', 'encoder.layer1.2') + (('semantic_decoder.stage0.cat', 'instance_decoder.stage0.cat') if name == 'pdl' else ()):
                res[f'{name}.{B}x{S}.{t}'] = model.tap(t).float().cpu().numpy()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): model(x, 2, False)
            e1.record(); torch.cuda.synchronize()
            print(name, B, S, VAR, os.environ.get(VAR), 'ms/forward', e0.elapsed_time(e1) / 3)
    np.savez(sys.argv[2], **res)
