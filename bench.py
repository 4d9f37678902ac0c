#!/usr/bin/env python
"""bench.py -- the two halves of BASELINE.json's metric on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3                 # EM tiles/sec (headline, default workload)
    python bench.py --gpus N ...                                    # starts its own N ranks (torchrun child) when
                                                                    # RANK is unset; under torchrun it IS a rank
    python bench.py --workload stack3d --gpus N ...                 # voxels/sec of the 3-D stack path, z-slabs over N GPUs

workload `tiles` (BASELINE configs[1]): one "step" = one pass of the hot path over one batch of synthetic EM tiles
already resident in HBM as uint8: fused normalisation + Panoptic-DeepLab/PointRend forward (fp16 MFMA, fp32
accumulate) + sigmoid + centre NMS/voting + panoptic merge -> int64 label maps on the device, 32 distinct 1024x1024
tiles per GPU.  Tiles are independent: N GPUs shard tiles with no data-path collective ("weak" scaling).  The same
JSON line also carries the Engine2d-level rate (host uint8 tile -> int32 numpy label map incl. force_connected and
both PCIe copies, pipelined), the batch-1 latency, and -- N = 1 only, outside the timed region -- the CPU port beside
it and the 512^3 ortho-plane job (configs[2]).

workload `stack3d` (configs[3] in small): a procedural (hash-seeded, never stored) uint8 volume of `--depth` slices
PER GPU of `--size`^2, xy stack inference with the recursive median through MultiGPUEngine3d's slab pipeline
(RCCL neighbour halo + filtered carry) + matching / tracking on rank 0; value = voxels/s of the whole job.

N > 1, workload `tiles`: after the tile measurement rank 0 runs the z-slab job (`--slab-depth` slices of `--slab-size`^2 per
rank) as a bounded CHILD job on the same GPUs (`slab_job_child`, internal workload `slabjob`) and reports it as the
`stack3d` block -- the headline line never depends on the only part of the bench with neighbour traffic between ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0   # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0      # HBM3E, same guide
PEAK_F32_TFLOPS = 157.3    # dense fp32 MFMA (v_mfma_f32_*_f32), same guide


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['tiles', 'stack3d', 'slabjob'], default='tiles')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--precision', choices=['fp16', 'fp16x3', 'fp32'], default='fp16',
                    help="tiles workload: precision of the HEADLINE network.  The default is the fp16 engine (BASELINE's metric is quoted "
                         "in fp16); 'fp16x3' -- the product's default precision -- times the same step on that mode (profiles: "
                         "`python bench.py --precision fp16x3 --batch 16`), with the roofline block of ITS dominant kernel")
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--micro-batch', type=int, default=0, help='forward in chunks of this many tiles (0 = whole batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-tiles', type=int, default=8)
    ap.add_argument('--stack3d', type=int, default=512, help='tiles workload: side of the 3-D cube of the second metric (0 = skip)')
    ap.add_argument('--engine2d', type=int, default=1, help='tiles workload: also measure the Engine2d-level rate (0 = skip)')
    ap.add_argument('--latency', type=int, default=1, help='tiles workload: also measure the batch-1 latency (0 = skip)')
    ap.add_argument('--fine-boundaries', type=int, default=1, help='tiles workload: also time the fine-boundary post-processing '
                                                                   '(coarse_boundaries=False) on 4 of the tiles (0 = skip)')
    ap.add_argument('--fp32-mode', type=int, default=4, help="tiles workload: batch of the fp32 reference mode's rate (precision='fp32'; 0 = skip); "
                    "the fp16x3 mode is measured at twice this batch")
    ap.add_argument('--depth', type=int, default=128, help='stack3d workload: slices per GPU')
    ap.add_argument('--slab-size', type=int, default=4096, help='tiles workload, N > 1: slice side of the z-slab job of the `stack3d` block')
    ap.add_argument('--slab-depth', type=int, default=16, help='tiles workload, N > 1: slices per rank of that job')
    ap.add_argument('--ks', type=int, default=3, help='stack3d workload: median kernel size')
    ap.add_argument('--slab-timeout', type=float, default=420.0, help='tiles workload, N > 1: bound (s) on the z-slab job, which runs as a child job of its own')
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as a torchrun child BEFORE anything touches the GPU (this
    process never initialises HIP; torch.cuda.device_count() does not) and relay its output and exit code."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if os.environ.get('EMP_BENCH_SHARE_GPU') == '1' and have >= 1:
        have = args.gpus      # diagnostic: the N ranks time-share GPU 0, gloo transport (main)
    if have < args.gpus:
        print(f'bench.py: --gpus {args.gpus} requested but {have} GPU(s) visible; refusing to report a smaller job',
              file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(cfg, P, tile_size, n_tiles, seed, keep=None):
    """Oracle (CPU restatement of the reference engine) on a bounded sample of the same workload.  keep: a dict that
    receives the oracle's head tensors and label map of tile 0 (the checker side of the line's `parity` block)."""
    import torch
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model, postprocess as opp
    torch.set_num_threads(min(os.cpu_count() or 1, 32))  # oneDNN convs stop scaling beyond ~32 threads
    tiles = synth.em_tiles(n_tiles, tile_size, seed=seed)
    heads = []

    def model(x, rs, interp):
        taps = {}
        o = pdl_model.pdl_forward(P, torch.from_numpy(x), cfg, rs, interp, taps)
        o = {k: v.numpy() for k, v in o.items()}
        if keep is not None and not heads:
            heads.append(dict(o, sem_coarse=taps['sem_coarse'].numpy()))
        return o

    eng = opp.RenderEngine(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                           padding_factor=16, coarse_boundaries=True)
    t0 = time.perf_counter()
    for i, t in enumerate(tiles):  # the reference asserts batch 1 (engines.py:306): sequential calls
        pan = eng(normalize(t, 0.57571, 0.12765)[None, None], t.shape, 1)
        if keep is not None and i == 0:
            keep.update(heads[0], pan=pan, tile=t)
    dt = time.perf_counter() - t0
    return {'value': round(n_tiles / dt, 4), 'unit': 'tiles/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n_tiles} x {tile_size}x{tile_size} uint8 EM-like tiles, fp32, sequential batch-1 calls '
                      f'({dt:.1f} s)'}


def parity_block(model, eng, ref, sub, mul, dev):
    """Distance of the engine's float outputs to the fp32 oracle forward on tile 0 of the CPU-baseline sample (the oracle
    ran it anyway: cpu_baseline(keep=)) -- the north star's "within 1e-3 on the float semantic / center heatmaps", stated
    with the numbers: rms AND max, the share of final semantic cells beyond 1e-3 (PointRend refines the 8192 most uncertain
    cells per step; a cell refined on one side only differs by refined - interpolated), and the share of pixels whose
    foreground differs after both pipelines' post-processing.  Gated in tests/test_gpu_parity_fullsize.py."""
    import numpy as np
    import torch
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    x = torch.from_numpy(ref['tile'])[None, None].to(dev)
    o = {k: v.float().cpu().numpy() for k, v in model(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul)).items()}
    h4, w4 = o['ctr_hmp'].shape[-2:]
    coarse = model.tap_raw('semantic_head.out', (1, o['sem_logits'].shape[1], h4, w4)).float().cpu().numpy()
    pan = eng.call_raw(x, sub, mul).cpu().numpy()
    e_ctr = np.abs(o['ctr_hmp'] - ref['ctr_hmp'])
    e_off = np.abs(o['offsets'] - ref['offsets'])
    e_sem = np.abs(sig(coarse) - sig(ref['sem_coarse']))
    e_prob = np.abs(sig(o['sem_logits']) - sig(ref['sem_logits']))
    rms = lambda e: float(np.sqrt((e.astype(np.float64) ** 2).mean()))
    want = np.asarray(ref['pan']).reshape(pan.shape)
    return {'tile': 'tile 0 of the cpu_baseline sample, batch-1 call', 'vs': 'fp32 oracle forward (= the reference forward, tests/golden)',
            'ctr_rms': round(rms(e_ctr), 6), 'ctr_max': round(float(e_ctr.max()), 6),
            'sem_rms': round(rms(e_sem), 6), 'sem_max': round(float(e_sem.max()), 6),
            'sem_note': 'semantic head probability before PointRend (256 x 256 per 1024^2 tile)',
            'off_rms_px': round(rms(e_off), 5), 'off_max_px': round(float(e_off.max()), 5),
            'final_prob_rms': round(rms(e_prob), 6), 'final_prob_frac_over_1e3': round(float((e_prob > 1e-3).mean()), 6),
            'fg_flip_frac': round(float(((pan > 0) != (want > 0)).mean()), 6),
            'foreground_fraction': round(float((want > 0).mean()), 4),
            'tolerance': 'north star: 1e-3.  The fp16 engine (this line\'s `value`) sits AT 1e-3 in rms -- 0.82e-3 .. 1.38e-3 on the centre map '
                         'over 8 tiles x 3 weight seeds (profiles/r05_parity_stats.json) -- and at ~5e-3 in the max norm; the mode that '
                         'meets 1e-3 in the max norm on every sample is precision=\'fp16x3\' (`fp16x3_mode` in this line), the exact one '
                         '\'fp32\'.  Label maps are bit-exact given identical head tensors'}


def x3p_traffic_from_profiles():
    """HBM bytes per launch of conv16x3p_kernel from the committed PMC passes (profiles/r06_x3p_traffic.json, tools/x3p_traffic.sh),
    or None; quoted only while the kernel's source is the profiled one"""
    import hashlib
    path = os.path.join(ROOT, 'profiles', 'r06_x3p_traffic.json')
    try:
        d = json.load(open(path))
        h = hashlib.sha256(open(os.path.join(ROOT, 'empanada-napari_amd', 'csrc', 'conv16x3p.hip'), 'rb').read()).hexdigest()[:16]
        if d.get('source_sha16') != h:
            return None
        return d.get('hbm_bytes_per_launch')
    except Exception:
        return None


def fp32_mode_block(cfg, P, host_tiles, sub, mul, dev, batch=4, steps=3, precision='fp32', ref=None, stack3d=0):
    """Rate of the library's fp32 REFERENCE MODE (precision='fp32', csrc/ref32.hip) or of its fp16x3 mode (the same graph with
    split-fp16 convolutions on the fp16 matrix pipe, csrc/conv16x3.hip) on the same workload: the same step (forward +
    probability + voting + merge -> int64 label maps) over `batch` of the bench's tiles, outside the timed region of
    `value`.  The reference computes this path in fp32 (empanada/inference/engines.py:248-255); these are the modes whose
    float heads are within 1e-3 of it in the MAX norm (tests/test_gpu_fp32_mode.py, tests/test_gpu_fp16x3.py), so their
    rates belong next to the fp16 engine's."""
    import torch
    from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine, logits_to_prob
    m32 = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=precision)
    e32 = PanopticDeepLabRenderEngine(m32, thing_list=[1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)
    x = torch.from_numpy(host_tiles[:batch])[:, None].to(dev)

    def step():
        o = m32(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
        sem = logits_to_prob(o['sem_logits'])
        cells, _, _, kmax = e32.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
        return e32.panoptic_merge_int(sem, cells, kmax)

    step()
    torch.cuda.synchronize()
    if precision == 'fp16x3':
        m32.profile(True)      # HIP-event pairs around the plane region's launches (conv16x3p_kernel) inside the timed region
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = m32.last_flops() / dt / 1e12
    dom = None
    if precision == 'fp16x3':
        dms, dfl, dn = m32.profile_read()
        m32.profile(False)
        if dn:
            ach = dfl / (dms * 1e-3) / 1e12
            dom = {'bound': 'mfma', 'kernel': 'conv16x3p_kernel (256x256 tile over hl32 planes: ResNet layer3 / layer4, ASPP)',
                   'achieved': round(ach, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F16_TFLOPS, 4),
                   'launches_per_step': dn // steps, 'kernel_ms_per_step': round(dms / steps, 3),
                   'kernel_share_of_step': round(dms / steps / (dt * 1e3), 3),
                   'flops_per_step_fp16': dfl / steps, 'traffic': x3p_traffic_from_profiles(),
                   'source': 'HIP events on the forward stream around each launch, inside this timed region; flops = 3 fp16 '
                             'MFMA products per MAC x 2 x M x Cout x K'}
    if precision == 'fp32':
        res = {'tiles_per_s': round(batch / dt, 2), 'ms_per_step': round(dt * 1e3, 2), 'batch': batch, 'steps': steps,
               'tflops': round(tf, 2), 'frac_of_157TF': round(tf / PEAK_F32_TFLOPS, 4), 'peak_tflops': PEAK_F32_TFLOPS,
               'note': "precision='fp32': fp32 maps and weights, exact fp32 MFMA, unfused; heads within 1e-4 of the fp32 oracle in "
                       'the max norm at this size (tests/test_gpu_fp32_mode.py); same step as `value` (forward + voting + merge)'}
    else:
        res = {'tiles_per_s': round(batch / dt, 2), 'ms_per_step': round(dt * 1e3, 2), 'batch': batch, 'steps': steps,
               'tflops_fp32_equivalent': round(tf, 2), 'tflops_fp16_mfma': round(3 * tf, 2),
               'frac_of_fp16_peak': round(3 * tf / PEAK_F16_TFLOPS, 4), 'roofline': dom,
               'note': "precision='fp16x3' -- the product's DEFAULT since round 6: the fp32 mode's graph, every convolution as three "
                       'fp16 MFMAs per product (operands split hi + lo, fp32 accumulate); layer3 / layer4 / ASPP maps as hl32 planes '
                       'on the 256x256 LDS-DMA tile (csrc/conv16x3p.hip), the separable blocks fused (csrc/sepconv_x3.hip); heads '
                       'within 1e-3 of the fp32 forward in the MAX norm on every one of 8 tiles x 3 weight seeds '
                       '(tests/test_gpu_fp16x3.py); same step as `value`'}
    if precision == 'fp16x3':      # the reference API's contract (engines.py:300-325: one tile per call) in the default precision
        one = x[:1]
        for _ in range(3):
            step1 = m32(one, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = m32(one, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
            cells, _, _, kmax = e32.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
            e32.panoptic_merge_int(logits_to_prob(o['sem_logits']), cells, kmax)
        torch.cuda.synchronize()
        res['latency_ms_batch1'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        del step1
        try:      # the API level in the default precision: host uint8 tiles -> int32 numpy label maps (the line's engine2d_tiles_per_s, fp16x3)
            from empanada_napari_amd.inference import Engine2d
            mc = {'model': m32, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
                  'norms': {'mean': 0.57571, 'std': 0.12765}}
            e2 = Engine2d(mc, label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5, device=dev)
            imgs = [host_tiles[i % len(host_tiles)] for i in range(batch * 16)]      # 16 batches, as the fp16 line: fill and drain amortised
            e2.infer_batch(imgs[:2 * batch], batch=batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e2.infer_batch(imgs, batch=batch)
            res['engine2d_tiles_per_s'] = round(len(imgs) / (time.perf_counter() - t0), 2)
            del e2
        except Exception as e:      # noqa: BLE001
            res['engine2d_tiles_per_s'] = {'error': f'{type(e).__name__}: {e}'}
    if stack3d:      # the 3-D half of the metric in this precision (the same job as the line's `stack3d` block, no CPU leg)
        try:
            j3 = stack3d_line(m32, stack3d, with_cpu=False)
            res['stack3d'] = {k: j3[k] for k in ('value', 'unit', 'volume', 'seconds', 'consensus_objects')}
        except Exception as e:      # noqa: BLE001
            res['stack3d'] = {'error': f'{type(e).__name__}: {e}'}
    if ref:      # the checker side: the oracle's fp32 heads of tile 0 of the cpu_baseline sample (cpu_baseline(keep=))
        import numpy as np
        sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
        o = m32(torch.from_numpy(ref['tile'])[None, None].to(dev), 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
        o = {k: v.float().cpu().numpy() for k, v in o.items()}
        h4, w4 = o['ctr_hmp'].shape[-2:]
        coarse = m32.tap_raw('semantic_head.out', (1, o['sem_logits'].shape[1], h4, w4)).float().cpu().numpy()
        res['parity'] = {'vs': 'fp32 oracle forward, tile 0 of the cpu_baseline sample',
                         'ctr_max': float(np.abs(o['ctr_hmp'] - ref['ctr_hmp']).max()),
                         'sem_max': float(np.abs(sig(coarse) - sig(ref['sem_coarse'])).max()),
                         'off_max_px': float(np.abs(o['offsets'] - ref['offsets']).max()),
                         'tolerance': 'north star: 1e-3 -- met in the MAX norm'}
    del m32, e32
    torch.cuda.empty_cache()
    return res


def fine_boundaries_block(model, tiles, sub, mul, batch=4, steps=5):
    """The widget's `fine_boundaries` option (reference: _volume_inference.py:39,183 -> coarse_boundaries=False): the
    instance heads interpolated to the image size and every pixel voted against every centre at step 1
    (postprocess.py:78-169) -- the reference's most expensive post-processing case (11.5 s per 1024^2 tile on CPU,
    BASELINE.md section 2).  Same network and tiles as `value`, outside its timed region; tests/test_gpu_fine_boundaries.py
    holds the parity test at >= 1 500 centres per tile (profiles/r05_fine_boundaries.json)."""
    import torch
    from empanada_napari_amd.engines import PanopticDeepLabRenderEngine, logits_to_prob
    eng = PanopticDeepLabRenderEngine(model, thing_list=[1], label_divisor=100000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=False)
    x = tiles[:batch]

    def step():
        o = model(x, 2, interpolate_ins=True, sub=float(sub), mul=float(mul))
        sem = logits_to_prob(o['sem_logits'])
        cells, _, num, kmax = eng.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
        return eng.panoptic_merge_int(sem, cells, kmax), o, num

    _, o, num = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        eng.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
    e1.record()
    torch.cuda.synchronize()
    return {'ms_per_tile': round(dt * 1e3 / batch, 3), 'tiles_per_s': round(batch / dt, 1),
            'voting_ms_per_tile': round(e0.elapsed_time(e1) / steps / batch, 4), 'batch': batch,
            'centres_per_tile': [int(v) for v in num.cpu().tolist()],
            'note': 'coarse_boundaries=False: forward with interpolate_ins=True + NMS / nearest-centre voting at 1024^2 + merge'}


def cpu_stack_baseline(cfg, P, vol, n_slices=12):
    """The oracle's restatement of the reference's per-axis control flow (3-D engine with the recursive median, dense ->
    RLE, matcher, tracker) on the first ``n_slices`` xy slices of the same volume."""
    import numpy as np
    import torch
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model, postprocess as opp, sparse as osp
    torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def model(x, rs, interp):
        o = pdl_model.pdl_forward(P, torch.from_numpy(x), cfg, rs, interp)
        return {k: v.numpy() for k, v in o.items()}

    eng = opp.RenderEngine3d(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                             padding_factor=16, coarse_boundaries=True, median_kernel_size=3)
    sub = np.ascontiguousarray(vol[:n_slices])
    t0 = time.perf_counter()
    pans = []
    for z in range(sub.shape[0]):
        r = eng(normalize(sub[z], 0.57571, 0.12765)[None, None], sub[z].shape, 1)
        if r is not None:
            pans.append(r[0])
    pans += [s[0] for s in eng.end(1)]
    m = osp.RLEMatcher(1, 10000, 0.25, 0.25)
    stack = [osp.apply_matchers(osp.pan_seg_to_rle_seg(p, [1], 10000, [1], force_connected=True), [m]) for p in pans]
    m.target_rle, m.assign_new = None, False
    tr = osp.InstanceTracker(1, 10000, sub.shape, 'xy')
    for idx in range(len(pans) - 1, -1, -1):
        tr.update(osp.apply_matchers(stack[idx], [m])[1], idx)
    tr.finish()
    dt = time.perf_counter() - t0
    return {'value': round(sub.size / dt, 1), 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n_slices} xy slices of the same volume, ONE axis pass (3-D engine + median + dense->RLE + matcher '
                      f'+ tracker), fp32 ({dt:.1f} s); the three-axis job + consensus is >= 3x this work per voxel'}


def stack_roofline(flops_fwd, voxels_per_axis, axes, ks, seconds):
    """MFMA work of the forward + mandatory HBM traffic of the per-voxel stages, against the time of the whole job."""
    hbm = axes * voxels_per_axis * ((ks + 1) * 4 + 12 + 8.75)     # median (ks+1)*4 B, CCL+RLE ~12 B, voting/merge 8.75 B per voxel
    tf = flops_fwd / seconds / 1e12
    return {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tf / PEAK_F16_TFLOPS, 4), 'traffic': None,
            'note': 'whole-job rate: forward FLOPs of every slice / wall time (host matching and consensus included in the time)',
            'forward_flops': flops_fwd, 'post_hbm_bytes_algorithmic': hbm,
            'post_hbm_gbs_if_alone': round(hbm / seconds / 1e9, 1)}


def stack3d_line(model, size, cfg=None, P=None, with_cpu=True):
    """Second half of BASELINE's metric (configs[2]): ortho-plane 3-D inference + consensus on a synthetic size^3 uint8
    cube held in a zarr v2 directory store, one GPU -- Engine3d.infer_on_axis x 3 + tracker_consensus as a user runs it
    (tools/bench_stack3d.py has the stage breakdown)."""
    import tempfile
    import numpy as np
    import torch
    from empanada_napari_amd import synth, zstore
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.blob_volume(size, size, size, seed=0, n_blobs=max(8, (size // 32) ** 2), fast=True)
    tmp = tempfile.mkdtemp(prefix='emp_bench_')
    src = zstore.open_store(os.path.join(tmp, 'em.zarr'), mode='w').create_array(
        'em', shape=vol.shape, dtype=np.uint8, chunks=(min(256, size),) * 3)
    src[...] = vol
    zvol = zstore.open_store(os.path.join(tmp, 'em.zarr'), mode='r')['em']
    eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
                   min_size=500, min_extent=5)
    flops = [0.0]

    def job(count=False):
        trackers = {}
        v = np.asarray(zvol[...])                # the store is read ONCE per job (counted in the time)
        for name in ('xy', 'xz', 'yz'):
            trackers[name] = eng.infer_on_axis(v, name)[1]
            if count:
                flops[0] += model.last_flops() / max(1, eng.slice_batch((size, size))) * size
        return list(tracker_consensus(trackers, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                      allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32))

    job()                    # one untimed pass, like the W warm-up steps of the tile metric: first launches, and the
    torch.cuda.synchronize()  # caching allocator's first hipMallocs for each axis' block sizes (~50 ms on the xz axis)
    t0 = time.perf_counter()
    out = job(count=True)
    dt = time.perf_counter() - t0
    res = {'metric': 'voxels/sec, 3-D ortho-plane stack + consensus', 'value': round(vol.size / dt, 1), 'unit': 'voxels/s',
           'volume': [size] * 3, 'source': 'zarr v2 directory store (uncompressed, 256^3 chunks), read inside the timed job',
           'seconds': round(dt, 3), 'consensus_objects': len(out[0][2]), 'n_gpus': 1,
           'roofline': stack_roofline(flops[0], vol.size, 3, 3, dt)}
    if with_cpu and cfg is not None:
        try:
            res['cpu_baseline'] = cpu_stack_baseline(cfg, P, vol, 12)
        except Exception as e:
            res['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return res


KERNEL_SOURCES = ('conv_igemm256.hip', 'conv_igemm.hip', 'pdl_net.hip', 'common.h')      # what the dominant kernel's traffic depends on


def kernel_source_hash():
    """sha256 over the sources that decide which launches the dominant kernel gets and what they fetch: a committed PMC
    profile is quoted only while it still describes THIS code (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'empanada-napari_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def traffic_from_profiles():
    """HBM bytes of the dominant kernel from the committed rocprofv3 PMC passes (tools/hbm_traffic.py, separate --pmc runs
    as the guide prescribes): NOT measured in this run -- the JSON names the file, the commit and the source hash it was
    taken at, and a profile whose source hash differs from the tree's is NOT quoted (traffic: null, `traffic_note` says
    stale) instead of sitting next to live numbers."""
    import glob
    cur = kernel_source_hash()
    stale = None
    for p in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic.json')), reverse=True):
        name = os.path.basename(p)
        if os.path.exists(p):
            try:
                tj = json.load(open(p))
                if 'steps_in_trace' not in tj:
                    # rounds 1-4: per-step figures divided by a step count passed on the command line (round 4's was
                    # stale, its file is 1.75x high: VERDICT r04 weak 4) -- never quoted
                    stale = stale or f'profiles/{name} predates the trace-derived step count'
                    continue
                k = tj['per_kernel']['conv_igemm256_kernel']
                src = f'profiles/{name}' + (f" @ {tj['commit']}" if 'commit' in tj else '')
                if tj.get('source_hash') != cur:
                    stale = stale or f"{src} is stale (kernel sources changed since: {tj.get('source_hash')} != {cur})"
                    continue
                return ((k['fetch_corrected'] + k['write']) / k['launches_per_step'], tj['hbm_bytes_per_step'],
                        src + f' (source hash {cur})')
            except Exception:
                continue
    return None, None, stale or 'no committed PMC profile'


def run_tiles(args, rank, local_rank, world, dist_on, dev):
    import numpy as np
    import torch
    import torch.distributed as dist
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine, logits_to_prob
    from empanada_napari_amd.preprocess import normalize_params

    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision=args.precision)      # default 'fp16': BASELINE's metric is quoted in fp16 -- the throughput opt-in (the product's default is 'fp16x3': `fp16x3_mode`)
    eng = PanopticDeepLabRenderEngine(model, thing_list=[1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)
    B, S = args.batch, args.size
    mb = args.micro_batch or B
    host_tiles = synth.em_tiles(B, S, seed=1234 + 1000 * rank)       # B DISTINCT tiles, different per rank
    tiles = torch.from_numpy(host_tiles)[:, None].to(dev)
    sub, mul = normalize_params(0.57571, 0.12765, 255)
    model.reserve(mb, S, S)
    fwd_ms = []

    def step(timed):
        outs = []
        for i in range(0, B, mb):
            x = tiles[i:i + mb]
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            o = model(x, 2, interpolate_ins=False, sub=float(sub), mul=float(mul))
            if timed:
                e1.record()
                fwd_ms.append((e0, e1))
            sem = logits_to_prob(o['sem_logits'])
            cells, _, _, kmax = eng.instance_cells_int(o['ctr_hmp'], o['offsets'], 1)
            outs.append(eng.panoptic_merge_int(sem, cells, kmax))
        return outs

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    model.profile(True)      # HIP-event pairs around every launch of the dominant kernel (256x256 conv tile)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    dom_ms, dom_flops, dom_launches = model.profile_read()
    model.profile(False)
    flops_fwd = model.last_flops() * (B / mb)  # per step (last_flops is per forward call of mb tiles)
    # the same K steps once more WITHOUT the HIP-event pairs around the dominant kernel's launches: what the
    # instrumentation inside the timed region costs (reported, not the headline)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    ms_plain = (time.perf_counter() - t1) * 1e3 / args.steps
    slab_block = None
    if dist_on and args.stack3d > 0:        # outside the timed region of `value`, and outside this job altogether
        slab_block = slab_job_child(args, rank, world)
    fwd_total_ms = sum(a.elapsed_time(b) for a, b in fwd_ms)
    ms_per_step = dt * 1e3 / args.steps
    value = world * B * args.steps / dt
    if rank != 0:
        return None

    fwd_ms_per_step = fwd_total_ms / args.steps
    traffic, step_traffic, traffic_src = (None, None, None)
    if B == 32 and S == 1024 and mb == 32 and args.precision == 'fp16':
        traffic, step_traffic, traffic_src = traffic_from_profiles()
    elif args.precision == 'fp16x3':
        traffic, traffic_src = x3p_traffic_from_profiles(), 'profiles/r06_x3p_traffic.json'
    fwd_tflops = flops_fwd / (fwd_ms_per_step * 1e-3) / 1e12
    # dominant kernel: the 256x256 tile (conv_igemm256w_kernel since round 5), every launch of the timed region bracketed by HIP events on its stream
    achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    res = {
        'metric': 'EM tiles/sec (1024^2 fp16)', 'value': round(value, 2), 'unit': 'tiles/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16',
        'data': 'synthetic',
        'config': {'workload': f'MitoNet-class PanopticDeepLabPR/resnet50 2D inference, {S}x{S} uint8 tiles ({B} distinct per '
                               f'GPU, resident in HBM when the timed region starts), batch {B} per GPU, forward + instance '
                               f"post-processing to int64 label maps on the device; `value` is the fp16 ENGINE (precision='fp16', "
                               f"the explicit throughput opt-in: BASELINE's metric is quoted in fp16); the product's default "
                               f"precision 'fp16x3' -- the one within 1e-3 (max norm) of the reference's fp32 forward -- is timed on "
                               f"the same tiles under `fp16x3_mode`",
                   'precision': args.precision, 'default_precision_of_the_product': 'fp16x3',
                   'tile': S, 'batch_per_gpu': B, 'micro_batch': mb, 'weights': 'seeded random init (seed 0)',
                   'parallelism': f'tile-sharded x{world}, no data-path collective; RCCL ranks: '
                                  f'{world if dist_on and dist.get_backend() == "nccl" else 0}',
                   'ranks_sharing_one_gpu': world if dist_on and dist.get_backend() != 'nccl' else 0},
        'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved / PEAK_F16_TFLOPS, 4), 'traffic': traffic,
                     'traffic_note': 'average HBM bytes per launch of this kernel (2*FETCH_SIZE + WRITE_SIZE, separate '
                                     'rocprofv3 PMC passes); NOT measured in this run: ' + str(traffic_src),
                     'step_traffic_all_kernels': step_traffic,
                     'kernel': ('conv_igemm256w_kernel (256x256 implicit-GEMM tile, whole-line LDS-DMA: ASPP 3x3, layer3/4 convs, projection shortcuts; conv_igemm256_kernel<0, true> with the fused next conv is not counted)'
                                if args.precision == 'fp16' else
                                'conv16x3p_kernel (256x256 tile over hl32 planes, both operands by LDS-DMA, 3 fp16 MFMAs per product: ResNet layer3 / layer4, ASPP; flops = fp16 MFMA flops)'),
                     'launches_per_step': dom_launches / max(args.steps, 1),
                     'kernel_ms_per_step': round(dom_ms / max(args.steps, 1), 3),
                     'kernel_share_of_step': round(dom_ms / max(args.steps, 1) / ms_per_step, 3),
                     'kernel_flops_per_step': dom_flops / max(args.steps, 1),
                     'forward_tflops': round(fwd_tflops, 2), 'forward_frac': round(fwd_tflops / PEAK_F16_TFLOPS, 4),
                     'flops_per_tile': round(flops_fwd / B / 1e9, 2), 'forward_ms_per_step': round(fwd_ms_per_step, 3)},
        'arena_gib': round(model.arena_bytes() / 2 ** 30, 2),
        'ms_per_step_uninstrumented': round(ms_plain, 3),
        'uninstrumented_note': 'the same steps on this rank without the HIP-event pairs around the dominant kernel (untimed extra pass)',
    }
    res['forward_steps_bench_loop'] = args.warmup + 2 * args.steps
    res['forward_steps_note'] = ('forward calls of the step loop: warm-up + timed + the uninstrumented repeat; with --engine2d 0 '
                                 '--latency 0 --fp32-mode 0 --no-cpu-baseline --stack3d 0 that is every forward of the process = the '
                                 'stem_pool_kernel launches of its rocprofv3 trace (tools/step_breakdown.py, tools/hbm_traffic.py '
                                 'count them there); `forward_calls_total` = every forward call of this model object in the process')
    # ---- batch-1 latency (the reference API's contract, engines.py:300-325: one tile per call) ----
    try:
        if not args.latency:
            raise RuntimeError('skipped (--latency 0)')
        one = tiles[:1]
        for _ in range(3):
            eng.call_raw(one, sub, mul)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.call_raw(one, sub, mul)
        torch.cuda.synchronize()
        res['latency_ms_batch1'] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    except Exception as e:
        res['latency_ms_batch1'] = {'error': f'{type(e).__name__}: {e}'}
    # ---- Engine2d level: host uint8 tile -> int32 numpy label map (force_connected + both PCIe copies), pipelined ----
    if args.engine2d and world == 1:
        try:
            from empanada_napari_amd.inference import Engine2d
            mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
                  'norms': {'mean': 0.57571, 'std': 0.12765}}
            e2 = Engine2d(mc, label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5, device=dev)
            imgs = [host_tiles[i % B] for i in range(B * 16)]      # 16 batches: the pipeline's fill and drain (last download + host copy) amortised
            e2.infer_batch(imgs[:2 * B], batch=B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = e2.infer_batch(imgs, batch=B)
            d2 = time.perf_counter() - t0
            res['engine2d_tiles_per_s'] = round(len(imgs) / d2, 2)
            res['engine2d_note'] = (f'Engine2d.infer_batch over {len(imgs)} host uint8 tiles -> int32 numpy label maps: '
                                    'H2D + forward + voting + merge + force_connected (8-connected components per class) '
                                    '+ D2H, uploads/downloads on side streams')
            assert outs[0].dtype == np.int32 and outs[0].shape == (S, S)
        except Exception as e:
            res['engine2d_tiles_per_s'] = {'error': f'{type(e).__name__}: {e}'}
    ref0 = {}
    if world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(cfg, P, S, args.cpu_tiles, 1234, keep=ref0)
        try:
            res['parity'] = parity_block(model, eng, ref0, sub, mul, dev)
        except Exception as e:
            res['parity'] = {'error': f'{type(e).__name__}: {e}'}
        # vs_baseline stays null: BASELINE.md holds no published number for this metric (its section 1: "Nothing").  The
        # ratio to the CPU port timed in this very run is given under its own name -- a reported baseline, not a target.
        try:
            res['vs_cpu_baseline'] = round(res['value'] / res['cpu_baseline']['value'], 1)
        except Exception:
            res['vs_cpu_baseline'] = None
    else:
        res['cpu_baseline'] = None
    if world == 1 and args.fp32_mode > 0 and S <= 1024 and args.precision == 'fp16':
        ref_heads = ref0 if (not args.no_cpu_baseline and 'ctr_hmp' in ref0) else None
        for key, prec, mult in (('fp32_mode', 'fp32', 1), ('fp16x3_mode', 'fp16x3', 4)):      # (16 tiles: a launch of the 256-cout layers fills the chip)
            try:
                res[key] = fp32_mode_block(cfg, P, host_tiles, sub, mul, dev, batch=min(mult * args.fp32_mode, B), precision=prec,
                                           ref=ref_heads, stack3d=args.stack3d if (prec == 'fp16x3' and not dist_on) else 0)
            except Exception as e:      # noqa: BLE001 -- the headline line must not depend on the extra measurement
                res[key] = {'error': f'{type(e).__name__}: {e}'}
    if world == 1 and args.fine_boundaries:
        try:
            res['fine_boundaries'] = fine_boundaries_block(model, tiles, sub, mul, batch=min(4, B))
            res['fine_boundaries_ms_per_tile'] = res['fine_boundaries']['ms_per_tile']
        except Exception as e:      # noqa: BLE001
            res['fine_boundaries'] = {'error': f'{type(e).__name__}: {e}'}
    res['forward_calls_total'] = model.forward_calls
    # the 3-D half of the headline metric, outside the timed region of `value` (rank 0, one GPU)
    res['stack3d'] = slab_block
    if world == 1 and not dist_on and args.stack3d > 0:
        try:
            res['stack3d'] = stack3d_line(model, args.stack3d, cfg, P, with_cpu=not args.no_cpu_baseline)
        except Exception as e:      # the headline line must not depend on the extra measurement
            res['stack3d'] = {'error': f'{type(e).__name__}: {e}'}
    return res


def slab_job_child(args, rank, world):
    """N > 1: the 3-D half of BASELINE's metric (`slab_job_block`) as a CHILD job on the same GPUs, after the tile measurement:
    rank 0 starts a second `torch.distributed.run` of this file (`--workload slabjob`, same N, a port of its own) while the
    ranks of this job idle on the host (a key of the rendezvous store: no kernel spins on a GPU meanwhile), and takes the child's JSON line
    as the `stack3d` block.  The z-slab job is the only part of the bench with neighbour traffic between ranks; whatever
    happens to it -- an exception on one rank, a peer that never answers -- ends with the child (killed as a process
    group after `--slab-timeout` seconds) and an `error` entry, never with the headline line."""
    import datetime
    import signal
    import socket
    import subprocess
    import torch.distributed as dist
    store = dist.distributed_c10d._get_default_store()      # the ranks wait on the rendezvous store: no group, no kernel
    key = 'emp_bench_slab_job_done'
    block = None
    if rank == 0:
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
        drop = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'GROUP_WORLD_SIZE', 'ROLE_RANK',
                'ROLE_WORLD_SIZE', 'ROLE_NAME', 'MASTER_ADDR', 'MASTER_PORT', 'EMP_BENCH_FORCE_DIST')
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith('TORCHELASTIC_')}
        env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), '--gpus', str(world),
               '--workload', 'slabjob', '--slab-size', str(args.slab_size), '--slab-depth', str(args.slab_depth),
               '--ks', str(args.ks)]
        import tempfile
        try:
            # output into files, not pipes: a pipe reaches EOF only when EVERY descendant holding it is gone
            with tempfile.TemporaryFile('w+') as fo, tempfile.TemporaryFile('w+') as fe:
                proc = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True)
                print(f'bench.py: z-slab child job started (pid {proc.pid}, bound {args.slab_timeout:.0f} s)', file=sys.stderr, flush=True)
                try:
                    proc.wait(timeout=args.slab_timeout)
                except subprocess.TimeoutExpired:
                    block = {'error': f'z-slab child job killed after {args.slab_timeout:.0f} s'}
                    started = []            # the launcher's descendants, by PID: the processes started here, nothing else
                    try:
                        import psutil
                        started = psutil.Process(proc.pid).children(recursive=True)
                    except Exception:       # noqa: BLE001
                        pass
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
                    for c in started:
                        try:
                            c.kill()
                        except Exception:   # noqa: BLE001 -- already gone
                            pass
                    try:
                        proc.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        pass
                if block is None:
                    fo.seek(0)
                    fe.seek(0)
                    lines = [ln for ln in fo.read().splitlines() if ln.startswith('{')]
                    if proc.returncode == 0 and lines:
                        block = json.loads(lines[-1])
                    else:
                        block = {'error': f'z-slab child job exit {proc.returncode}: ' + fe.read().strip()[-400:]}
        except Exception as e:      # noqa: BLE001 -- the headline line must not depend on the extra measurement
            block = {'error': f'{type(e).__name__}: {e}'}
        print('bench.py: z-slab child job ' + ('failed: ' + block['error'][:200] if 'error' in block else 'done'), file=sys.stderr, flush=True)
        store.set(key, '1')
    else:
        store.wait([key], datetime.timedelta(seconds=args.slab_timeout + 300))
    return block


def run_slabjob(args, rank, local_rank, world, dist_on, dev):
    """the child job of `slab_job_child`: the bench model, then the z-slab job; rank 0 prints the block"""
    from empanada_napari_amd import weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision='fp16')      # BASELINE's metric is quoted in fp16: the throughput opt-in (the product's default is 'fp16x3': `fp16x3_mode`)
    return slab_job_block(args, model, rank, world, dev)


def slab_job_block(args, model, rank, world, dev):
    """N > 1: the 3-D half of BASELINE's metric on the SAME ranks, after the tile measurement and outside its timed region --
    configs[3] in small: a procedural uint8 volume of `--slab-depth` slices of `--slab-size`^2 PER RANK, xy stack inference
    through MultiGPUEngine3d on RCCL (z-slab / block-interleaved schedule, neighbour halo + filtered carry, slab-wise
    matcher chained through the ranks).  Every rank calls it (SPMD, the ranks of `slab_job_child`'s child job); rank 0 returns the block.  Whatever `--gpus N`
    command the driver runs therefore records the 3-D scaling as well (weak: fixed slices per rank)."""
    import torch
    import torch.distributed as dist
    from empanada_napari_amd import multigpu, synth
    S, D = args.slab_size, args.slab_depth * world
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.ProceduralVolume((D, S, S), seed=7, cell=48, cache=True)
    multigpu.MultiGPUEngine3d.MIN_WORLD = 1
    shared = dist.get_backend() != 'nccl'      # EMP_BENCH_SHARE_GPU: all ranks on this one device
    eng = multigpu.MultiGPUEngine3d(mc, label_divisor=10000, median_kernel_size=args.ks, nms_kernel=3, nms_threshold=0.1,
                                    confidence_thr=0.5, min_size=500, min_extent=5,
                                    devices=[dev.index] * world if shared else None)

    def job():
        st, tr = eng.infer_on_axis(vol, 'xy')
        return len(tr[0].instances) if tr is not None else 0

    job()                                   # untimed: procedural synthesis (cached), first launches at this size
    eng.wait()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    jobs = 2
    nobj = 0
    for _ in range(jobs):
        nobj = job()
    eng.wait()                              # every rank's deferred chain work belongs to the job
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], device='cpu' if shared else dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank != 0:
        return None
    sec = float(t.item()) / jobs
    per_slice = model.last_flops()          # one slice per forward at 4096^2 (Engine3d.slice_batch); else per batch
    bs = max(1, min(64, (1 << 24) // (S * S)))
    flops = per_slice / max(1, min(bs, args.slab_depth)) * D
    tf = flops / sec / 1e12
    tm = getattr(eng, 'last_timing', None) or []
    slab = None
    if tm:
        r0 = tm[0]
        slab = {'ranks': len(tm), 'gpu_ms_per_slice': round(1e3 * r0['gpu_s'] / max(1, r0['slices']), 4),
                'rank0_host_ms_per_slice': round(1e3 * (r0['tail_s'] + getattr(eng, 'last_merge_s', 0.0)) / max(1, D), 4),
                'per_rank': [{k: (round(v, 5) if isinstance(v, float) else
                                  {a: round(b, 5) for a, b in v.items()} if isinstance(v, dict) else v) for k, v in x.items()} for x in tm]}
    return {'metric': 'voxels/sec, 3-D stack (xy) z-slab inference', 'value': round(float(D) * S * S / sec, 1), 'unit': 'voxels/s',
            'n_gpus': 1 if shared else world, 'rccl_ranks': 0 if shared else world, 'ranks_sharing_one_gpu': world if shared else 0,
            'scaling': 'weak', 'seconds_per_job': round(sec, 4), 'jobs_timed': jobs,
            'volume': [D, S, S], 'slices_per_rank': args.slab_depth, 'ks': args.ks, 'tracked_objects': nobj,
            'schedule': 'block-interleaved' if os.environ.get('EMP_MG_BLOCK', '1') != '0' else 'contiguous slabs',
            'roofline': {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': PEAK_F16_TFLOPS * (1 if shared else world), 'unit': 'TFLOP/s',
                         'frac': round(tf / (PEAK_F16_TFLOPS * (1 if shared else world)), 4), 'traffic': None, 'forward_flops': flops,
                         'note': 'whole-job rate over all ranks: forward FLOPs of every slice / wall time (median, voting, '
                                 'merge, run extraction, matcher chain included)'},
            'slab_pipeline': slab, 'cpu_baseline': None}


def run_stack3d(args, rank, local_rank, world, dist_on, dev):
    """z-slab stack inference over `world` GPUs on a procedural volume (weak scaling: --depth slices per GPU)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from empanada_napari_amd import multigpu, synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab
    from empanada_napari_amd.inference import Engine3d

    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True, precision='fp16')      # BASELINE's metric is quoted in fp16: the throughput opt-in (the product's default is 'fp16x3': `fp16x3_mode`)
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    S, D = args.size, args.depth * world
    vol = synth.ProceduralVolume((D, S, S), seed=7, cell=48, cache=True)    # synthesised in the warm-up pass
    kw = dict(label_divisor=10000, median_kernel_size=args.ks, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
              min_size=500, min_extent=5)
    ranks_one_gpu = int(os.environ.get('EMP_BENCH_RANKS_ON_ONE_GPU', '0'))
    n_ranks = world
    if ranks_one_gpu > 1 and not dist_on:
        # diagnostic (no multi-GPU box in the builder's reach): R real ranks of the slab pipeline time-share THIS GPU, gloo
        # transport; the GPU phase is R x slower than on R GPUs, the host-side matcher chain is what it would be there
        D = args.depth * ranks_one_gpu
        vol = synth.ProceduralVolume((D, S, S), seed=7, cell=48, cache=True)
        mc = dict(mc, model=weights.seeded_state_dict(cfg, seed=0))
        eng = multigpu.MultiGPUEngine3d(mc, world_size=ranks_one_gpu, dist_backend='gloo', devices=[0] * ranks_one_gpu, **kw)
        host = [None]

        def job():
            if host[0] is None:
                host[0] = vol.block(0, 0, D, dev).cpu().numpy()
            return eng.infer_on_axis(host[0], 'xy')
        n_ranks = ranks_one_gpu
    elif dist_on:
        multigpu.MultiGPUEngine3d.MIN_WORLD = 1
        shared = dist.get_backend() != 'nccl'      # EMP_BENCH_SHARE_GPU: all ranks on this one device, gloo transport
        eng = multigpu.MultiGPUEngine3d(mc, devices=[dev.index] * world if shared else None, **kw)
        job = lambda: eng.infer_on_axis(vol, 'xy')
    else:
        e3 = Engine3d(mc, device=dev, **kw)
        host = [None]

        def job():
            if host[0] is None:     # one GPU: the volume is synthesised once on the device and handed over as numpy
                host[0] = vol.block(0, 0, D, dev).cpu().numpy()
            return e3.infer_on_axis(host[0], 'xy')
    for _ in range(max(1, min(args.warmup, 1))):
        st, tr = job()
        if tr is not None:
            len(tr[0].instances)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    nobj, prev = 0, None
    for _ in range(args.steps):      # a job's host-side matching / tracking tail runs behind the next job's GPU work
        st, tr = job()
        if prev is not None:
            nobj = len(prev[0].instances)        # joins the previous job's deferred pass
        prev = tr
    if prev is not None:
        nobj = len(prev[0].instances)
    if dist_on and hasattr(eng, 'wait'):
        eng.wait()                          # SPMD ranks: the deferred backward chain / gather of the last job
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    vox = float(D) * S * S
    sec = dt / args.steps
    # forward FLOPs of the whole job: last_flops() is one forward call of this rank (a batch of slices of equal size)
    bs = max(1, min(64, (1 << 24) // (S * S)))
    per_slice = model.last_flops() / max(1, min(bs, args.depth if args.depth % bs == 0 else args.depth % bs))
    flops = per_slice * D
    tf = flops / sec / 1e12
    slab = None
    if getattr(locals().get('eng'), 'last_timing', None):
        # the slab pipeline's split of a job (last job): per rank the GPU phase (forward, halo / carry, median, voting,
        # merge, run extraction) and the host matcher (total, and its tail behind the GPU phase: ghosts, chains, tracking);
        # rank 0 also concatenates the per-slab tracks and filters
        tm = eng.last_timing
        r0 = tm[0]
        slab = {'ranks': len(tm),
                'gpu_ms_per_slice': round(1e3 * r0['gpu_s'] / max(1, r0['slices']), 4),
                'rank0_host_ms_per_slice': round(1e3 * (r0['tail_s'] + getattr(eng, 'last_merge_s', 0.0)) / max(1, D), 4),
                'rank0_matcher_ms_per_own_slice': round(1e3 * r0['host_s'] / max(1, r0['slices']), 4),
                'per_rank': [{k: (round(v, 5) if isinstance(v, float) else
                                  {a: round(b, 5) for a, b in v.items()} if isinstance(v, dict) else v) for k, v in t.items()} for t in tm],
                'merge_and_filter_s': round(getattr(eng, 'last_merge_s', 0.0), 5),
                'note': 'rank0_host_ms_per_slice = (rank 0 matcher tail behind its GPU phase + track concatenation + size '
                        'filters) / ALL slices of the job: the sequential host work left on rank 0 per slice; '
                        'gpu_ms_per_slice = rank 0 GPU phase / its own slices'}
    if ranks_one_gpu > 1:
        eng.close()
    shared_spmd = dist_on and dist.get_backend() != 'nccl'      # EMP_BENCH_SHARE_GPU: the ranks time-share ONE device over gloo
    gpus = 1 if shared_spmd else world
    return {'metric': 'voxels/sec, 3-D stack (xy) z-slab inference', 'value': round(vox / sec, 1), 'unit': 'voxels/s',
            'n_gpus': gpus, 'steps': args.steps, 'warmup': 1, 'ms_per_step': round(sec * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'procedural uint8 volume {D}x{S}x{S} ({args.depth} slices per GPU), xy stack inference, '
                                   f'recursive median ks={args.ks}, z-slabs over {world} GPU(s), neighbour halo + filtered '
                                   f'carry over RCCL, slab-wise C++ matcher + tracker on every rank (forward / backward '
                                   f'state chained through the ranks), per-slab tracks to rank 0',
                       'rccl_ranks': world if dist_on and not shared_spmd else 0, 'tracked_objects': nobj,
                       'ranks_sharing_one_gpu': ranks_one_gpu if ranks_one_gpu > 1 else (world if shared_spmd else 0),
                       'parallelism': f'z-slab x{n_ranks}'},
            'roofline': {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': PEAK_F16_TFLOPS * gpus, 'unit': 'TFLOP/s',
                         'frac': round(tf / (PEAK_F16_TFLOPS * gpus), 4), 'traffic': None,
                         'note': 'whole-job rate: forward FLOPs of every slice / wall time of the job over all ranks (median, '
                                 'voting, merge, run extraction and the host matcher included in the time); kernel-level '
                                 'roofline: the tiles workload', 'forward_flops': flops},
            'slab_pipeline': slab, 'cpu_baseline': None}


def main():
    args = parse_args()
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    import torch
    import __graft_entry__ as graft
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} does not match the launcher\'s WORLD_SIZE {world}', file=sys.stderr)
        sys.exit(2)
    dist_on = world > 1 or os.environ.get('EMP_BENCH_FORCE_DIST') == '1'   # the env switch runs the RCCL path on one GPU
    # EMP_BENCH_SHARE_GPU=1 (diagnostic; the builder's boxes have ONE GPU): the N ranks time-share GPU 0 and talk over gloo
    # (device maps staged through the host) -- the same code path of every rank as on N GPUs, not a scaling measurement;
    # the JSON line says so (`ranks_sharing_one_gpu`)
    share = world > 1 and os.environ.get('EMP_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    graft.load_package()
    run = {'tiles': run_tiles, 'stack3d': run_stack3d, 'slabjob': run_slabjob}[args.workload]
    res = run(args, rank, local_rank, world, dist_on, dev)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
