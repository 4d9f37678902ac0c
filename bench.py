#!/usr/bin/env python
"""bench.py -- tiles/sec of the MitoNet-class 2D hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic EM tiles that
is already resident in HBM as uint8: fused normalisation + Panoptic-DeepLab/
PointRend forward (fp16 MFMA, fp32 accumulate) + sigmoid + centre NMS/voting +
panoptic merge -> int64 label maps on the device (BASELINE.json configs[1]:
1024x1024 tiles, batch 32).  Tiles are independent, so N GPUs shard tiles with
no data-path collective ("weak" scaling: 32 tiles per GPU per step).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PEAK_F16_TFLOPS = 2500.0  # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(cfg, P, tile_size, n_tiles, seed):
    """Oracle (CPU restatement of the reference engine) on a bounded sample of the same workload."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.preprocess import normalize
    from oracle import pdl_model, postprocess as opp
    torch.set_num_threads(min(os.cpu_count() or 1, 32))  # oneDNN convs stop scaling beyond ~32 threads
    tiles = synth.em_tiles(n_tiles, tile_size, seed=seed)

    def model(x, rs, interp):
        o = pdl_model.pdl_forward(P, torch.from_numpy(x), cfg, rs, interp)
        return {k: v.numpy() for k, v in o.items()}

    eng = opp.RenderEngine(model, [1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3, confidence_thr=0.5,
                           padding_factor=16, coarse_boundaries=True)
    t0 = time.perf_counter()
    for t in tiles:  # the reference asserts batch 1 (engines.py:306): sequential calls
        eng(normalize(t, 0.57571, 0.12765)[None, None], t.shape, 1)
    dt = time.perf_counter() - t0
    return {'value': round(n_tiles / dt, 4), 'unit': 'tiles/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n_tiles} x {tile_size}x{tile_size} uint8 EM-like tiles, fp32, sequential batch-1 calls '
                      f'({dt:.1f} s)'}


def stack3d_line(model, size):
    """Second half of BASELINE's metric (configs[2]): ortho-plane 3-D inference + consensus on a synthetic size^3 uint8
    cube, one GPU -- Engine3d.infer_on_axis x 3 + tracker_consensus as a user runs it (tools/bench_stack3d.py has the
    stage breakdown and the CPU port beside it)."""
    from empanada_napari_amd import synth
    from empanada_napari_amd.inference import Engine3d, tracker_consensus
    mc = {'model': model, 'thing_list': [1], 'labels': [1], 'class_names': {1: 'mito'}, 'padding_factor': 16,
          'norms': {'mean': 0.57571, 'std': 0.12765}}
    vol = synth.blob_volume(size, size, size, seed=0, n_blobs=max(8, (size // 32) ** 2), fast=True)
    eng = Engine3d(mc, label_divisor=10000, median_kernel_size=3, nms_kernel=3, nms_threshold=0.1, confidence_thr=0.5,
                   min_size=500, min_extent=5)
    def job():
        trackers = {name: eng.infer_on_axis(vol, name)[1] for name in ('xy', 'xz', 'yz')}
        return list(tracker_consensus(trackers, None, mc, label_divisor=10000, pixel_vote_thr=2, cluster_iou_thr=0.75,
                                      allow_one_view=False, min_size=500, min_extent=5, dtype=np.uint32))

    job()                    # one untimed pass, like the W warm-up steps of the tile metric: first launches, and the
    torch.cuda.synchronize()  # caching allocator's first hipMallocs for each axis' block sizes (~50 ms on the xz axis)
    t0 = time.perf_counter()
    out = job()
    dt = time.perf_counter() - t0
    return {'metric': 'voxels/sec, 3-D ortho-plane stack + consensus', 'value': round(vol.size / dt, 1), 'unit': 'voxels/s',
            'volume': [size] * 3, 'seconds': round(dt, 3), 'consensus_objects': len(out[0][2]), 'n_gpus': 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--micro-batch', type=int, default=0, help='forward in chunks of this many tiles (0 = whole batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-tiles', type=int, default=8)
    ap.add_argument('--stack3d', type=int, default=512, help='side of the 3-D cube of the second metric line (0 = skip)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist_on = world > 1 or os.environ.get('EMP_BENCH_FORCE_DIST') == '1'   # the env switch runs the RCCL path on one GPU
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    graft.load_package()
    from empanada_napari_amd import synth, weights
    from empanada_napari_amd.engines import HipPanopticDeepLab, PanopticDeepLabRenderEngine
    from empanada_napari_amd.preprocess import normalize_params

    cfg = dict(weights.MITONET_PDL_CFG)
    P = weights.fold_state_dict(weights.seeded_state_dict(cfg, seed=0), cfg)
    model = HipPanopticDeepLab(P, cfg, device=dev, folded=True)
    eng = PanopticDeepLabRenderEngine(model, thing_list=[1], label_divisor=10000, nms_threshold=0.1, nms_kernel=3,
                                      confidence_thr=0.5, padding_factor=16, coarse_boundaries=True)
    B, S = args.batch, args.size
    mb = args.micro_batch or B
    # synthetic tiles: a few distinct ones tiled to the batch (generation cost only), different per rank
    base = synth.em_tiles(min(B, 4), S, seed=1234 + rank)
    tiles = torch.from_numpy(np.concatenate([base] * ((B + len(base) - 1) // len(base)))[:B])[:, None].to(dev)
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
            from empanada_napari_amd.engines import logits_to_prob
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
        out = step(True)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    dom_ms, dom_flops, dom_launches = model.profile_read()
    model.profile(False)
    fwd_total_ms = sum(a.elapsed_time(b) for a, b in fwd_ms)
    flops_fwd = model.last_flops() * (B / mb)  # per step (last_flops is per forward call of mb tiles)
    ms_per_step = dt * 1e3 / args.steps
    value = world * B * args.steps / dt

    if rank == 0:
        fwd_ms_per_step = fwd_total_ms / args.steps
        traffic = step_traffic = None
        try:  # HBM bytes from the committed PMC passes (only valid for the default workload)
            if B == 32 and S == 1024 and mb == 32:
                tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')))
                step_traffic = tj['hbm_bytes_per_step']
                k = tj['per_kernel']['conv_igemm256_kernel']
                traffic = (k['fetch_corrected'] + k['write']) / k['launches_per_step']     # per launch, like `achieved`
        except Exception:
            traffic = step_traffic = None
        fwd_tflops = flops_fwd / (fwd_ms_per_step * 1e-3) / 1e12
        # dominant kernel: conv_igemm256_kernel, every launch of the timed region bracketed by HIP events on its stream
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        res = {
            'metric': 'EM tiles/sec (1024^2 fp16)', 'value': round(value, 2), 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16',
            'data': 'synthetic',
            'config': {'workload': f'MitoNet-class PanopticDeepLabPR/resnet50 2D inference, {S}x{S} uint8 tiles, '
                                   f'batch {B} per GPU, forward + instance post-processing to int64 label maps',
                       'tile': S, 'batch_per_gpu': B, 'micro_batch': mb, 'weights': 'seeded random init (seed 0)',
                       'parallelism': f'tile-sharded x{world}, no collective'},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_F16_TFLOPS, 4), 'traffic': traffic,
                         'traffic_note': 'average HBM bytes per launch of this kernel (2*FETCH_SIZE + WRITE_SIZE, separate '
                                         'rocprofv3 PMC passes, profiles/r01_hbm_traffic.json)',
                         'step_traffic_all_kernels': step_traffic,
                         'kernel': 'conv_igemm256_kernel (256x256 implicit-GEMM tile: ASPP 3x3, layer3/4 convs)',
                         'launches_per_step': dom_launches / max(args.steps, 1),
                         'kernel_ms_per_step': round(dom_ms / max(args.steps, 1), 3),
                         'kernel_share_of_step': round(dom_ms / max(args.steps, 1) / ms_per_step, 3),
                         'kernel_flops_per_step': dom_flops / max(args.steps, 1),
                         'forward_tflops': round(fwd_tflops, 2), 'forward_frac': round(fwd_tflops / PEAK_F16_TFLOPS, 4),
                         'flops_per_tile': round(flops_fwd / B / 1e9, 2), 'forward_ms_per_step': round(fwd_ms_per_step, 3)},
            'arena_gib': round(model.arena_bytes() / 2 ** 30, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(cfg, P, S, args.cpu_tiles, 1234)
        else:
            res['cpu_baseline'] = None
        # the 3-D half of the headline metric, outside the timed region of `value` (rank 0, one GPU)
        res['stack3d'] = None
        if world == 1 and args.stack3d > 0:
            try:
                res['stack3d'] = stack3d_line(model, args.stack3d)
            except Exception as e:      # the headline line must not depend on the extra measurement
                res['stack3d'] = {'error': f'{type(e).__name__}: {e}'}
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
