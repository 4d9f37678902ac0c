"""torch-CPU fp32 restatement of the reference network forward (oracle, test only).

Consumes the folded ``name -> (w, b)`` parameters of
``empanada-napari_amd/weights.py::fold_state_dict`` and reproduces, op by op,
``QuantizablePanopticDeepLabPR.forward`` in eval mode
(reference: empanada/models/quantization/panoptic_deeplab.py:194-250).
Pinned against the reference itself by ``oracle/gen_golden.py``
(tests/golden/pdl_forward_*.npz).
"""
import os

import torch
import torch.nn.functional as F

RESNET50_LAYERS = (3, 4, 6, 3)


def precise_block(pre, ins_decoder=True, bifpn=False):
    """Which separable blocks the HIP engine computes with an exact depthwise half (csrc/sepconv_precise.hip: fp32 taps,
    depthwise result carried as an fp16 hi + lo pair; pointwise weights fp16) -- the rule of pdl_net.hip's ``precise_layer``: by default the
    blocks the centre heat-map depends on (the fusion convs of the decoder that feeds ``ins_center`` and the
    ``ins_center`` head); EMP_PRECISE_SEPCONV=2 every fused 5x5 block, 0 none, 6 the ``ins_xy`` head too.  ``pre`` is the
    block's parameter prefix (``<pre>.sepconv.0`` / ``.1``).  Test infrastructure mirrors the switch so that A/B runs stay comparable."""
    mode = int(os.environ.get('EMP_PRECISE_SEPCONV', '1'))
    if mode <= 0:
        return False
    if mode == 2:
        return True
    dec = 'instance_decoder.' if ins_decoder else 'semantic_decoder.'
    head, fuse = pre.startswith('ins_center.'), pre.startswith(dec)
    if mode == 3:          # A/B switches of the engine: only the head / only the decoder's fusion convs
        return head
    if mode == 4:
        return fuse
    if (mode == 6 or (mode == 1 and bifpn)) and pre.startswith('ins_xy.'):     # the offsets head: default on BiFPN networks
        return True
    return head or fuse


def precise_node(pre, ins_decoder=True, C=128, Cout=128):
    """pdl_net.hip's ``precise_node``: the 3x3 separable node blocks of the BiFPN that feeds the centre / offset heads run
    with the exact depthwise half by default (EMP_PRECISE_SEPCONV 1 / 6), those of both FPNs with 2 / 5, none otherwise --
    where the kernel supports the shape (``sepconvp_supported``)."""
    if not (C % 64 == 0 and 128 <= C <= 512 and Cout in (128, 256)):
        return False
    mode = int(os.environ.get('EMP_PRECISE_SEPCONV', '1'))
    if mode in (2, 5):
        return True
    if mode not in (1, 6):
        return False
    return pre.startswith('instance_fpn.' if ins_decoder else 'semantic_fpn.')


def wsplit_on():
    """pdl_net.hip's ``wsplit_on`` for a BiFPN network: the weights of the precise 128-cout separable blocks' pointwise
    convs and of the centre decoder's transposed convs are carried as fp16 hi + lo pairs (exact to ~2^-22)."""
    return os.environ.get('EMP_PRECISE_WSPLIT', '1')[:1] != '0' and int(os.environ.get('EMP_PRECISE_SEPCONV', '1')) in (1, 2, 5, 6)


def fsplit_on():
    """pdl_net.hip's ``fsplit_on``: the fused maps of the precise FPN's nodes travel as fp16 hi + lo pairs"""
    return wsplit_on() and os.environ.get('EMP_PRECISE_FSPLIT', '1')[:1] != '0'


def engine_emu(P, cfg, weights=True, acts=True):
    """``Fp16Emu`` configured like the HIP engine for this network: fp16 everywhere except where the engine carries hi + lo
    pairs (BiFPN: ``wsplit_names`` weights, the centre FPN's fused maps)."""
    emu = Fp16Emu(weights, acts, wsplit_names(P, cfg))
    if 'BiFPN' in cfg.get('arch', '') and fsplit_on():
        emu.split_acts_prefix = ('instance_fpn.' if cfg['ins_decoder'] else 'semantic_fpn.')
    return emu


def wsplit_names(P, cfg):
    """the parameters of a BiFPN network that the engine carries as hi + lo pairs (``Fp16Emu(split_weights=...)``)"""
    if 'BiFPN' not in cfg.get('arch', '') or not wsplit_on():
        return set()
    insd = bool(cfg['ins_decoder'])
    dec = 'instance' if insd else 'semantic'
    out = {k for k in P if k.startswith(f'{dec}_decoder.upsamplings.')}
    for k, (w, _) in P.items():
        if not k.endswith('.sepconv.1') or w.shape[0] != 128:
            continue
        pre = k[:-len('.sepconv.1')]
        if '.after_combines.' in k:
            if precise_node(k, insd, w.shape[1], w.shape[0]):
                out.add(k)
        elif precise_block(pre, insd, True) and not (pre.endswith('.head.0.0') and P[pre[:-len('.head.0.0')] + '.head.1'][0].shape[0] > 2):
            out.add(k)
    return out


def _t(a):
    return None if a is None else torch.from_numpy(a)


class Fp16Emu:
    """Storage-format emulation of the HIP engine for the error budget and the 'kernels are exact up to
    the fp16 format' parity test: fp32 arithmetic as everywhere in this oracle, but the tensors the engine
    keeps in fp16 (conv / depthwise weights, every activation map written to HBM or
    handed to the matrix pipe) are rounded to fp16 (round to nearest even) where the engine rounds them.

    ``weights`` / ``acts``: True (every site), False (none) or a set of site names.  Weight sites are the
    parameter names; activation sites are the name of the producing layer (a bottleneck block rounds once,
    after the residual add + ReLU, under the block's name), ``bilinear:<decoder>``, ``pr.features``.
    ``split_weights``: parameter names whose weights the engine carries as an fp16 hi + lo pair (exact to
    ~2^-22 relative): rounded to that pair instead of to one fp16."""

    split_acts_prefix = None      # activation sites '<prefix>...fuse<i>' are carried as hi + lo pairs (engine_emu)

    def __init__(self, weights=True, acts=True, split_weights=()):
        self.weights, self.acts, self.split_weights = weights, acts, set(split_weights)
        self.names = {}
        self.sites_w, self.sites_a = [], []

    def bind(self, P):
        self.names = {id(v[0]): k for k, v in P.items()}
        return self

    @staticmethod
    def r16(t):
        return t.to(torch.float16).to(torch.float32)

    def w(self, w):
        name = self.names.get(id(w))
        t = _t(w)
        if name is None:
            return t
        if name not in self.sites_w:
            self.sites_w.append(name)
        if name in self.split_weights:
            hi = self.r16(t)
            return hi + self.r16(t - hi)
        on = self.weights is True or (self.weights and name in self.weights)
        return self.r16(t) if on else t

    def a(self, site, x):
        if site not in self.sites_a:
            self.sites_a.append(site)
        on = self.acts is True or (self.acts and site in self.acts)
        if on and self.split_acts_prefix and site.startswith(self.split_acts_prefix) and '.fuse' in site:
            hi = self.r16(x)
            return hi + self.r16(x - hi)
        return self.r16(x) if on else x


_EMU = None          # set by model_forward(..., emu=...) for the duration of one forward


def _conv(x, p, stride=1, padding=0, dilation=1, groups=1, relu=False, site=None, w32=False):
    """site: activation-rounding site of the output (None: the engine does not round here);
    w32: the engine keeps this layer's weights in fp32."""
    w, b = p
    wt = _t(w) if (_EMU is None or w32) else _EMU.w(w)
    y = F.conv2d(x, wt, _t(b), stride, padding, dilation, groups)
    y = F.relu(y) if relu else y
    if _EMU is not None and site is not None:
        y = _EMU.a(site if site is not True else _EMU.names.get(id(w)), y)
    return y


def _round(site, x):
    return x if _EMU is None else _EMU.a(site, x)


def resnet50_forward(P, x, output_stride=16, taps=None):
    """encoders/resnet.py:217-229 with Bottleneck blocks (:109-129); BN folded."""
    x = _conv(x, P['encoder.conv1'], stride=2, padding=3, relu=True, w32=True)   # stem.hip: fp16 hi/lo split, fp32-exact
    if taps is not None:
        taps['stem'] = x
    p1 = _round('encoder.conv1', F.max_pool2d(x, kernel_size=3, stride=2, padding=1))
    pyr = [p1]
    x = p1
    for li, nblocks in enumerate(RESNET50_LAYERS, start=1):
        stride = 1 if li == 1 else 2
        dilation = 1
        if li == 4 and output_stride == 16:  # resnet.py:173-175
            stride, dilation = 1, 2
        for b in range(nblocks):
            pre = f'encoder.layer{li}.{b}'
            s = stride if b == 0 else 1
            identity = x
            out = _conv(x, P[f'{pre}.conv1'], relu=True, site=True)
            if taps is not None:
                taps[pre + '.c1'] = out
            out = _conv(out, P[f'{pre}.conv2'], stride=s, padding=dilation, dilation=dilation, relu=True, site=True)
            if taps is not None:
                taps[pre + '.c2'] = out
            out = _conv(out, P[f'{pre}.conv3'])
            if b == 0:
                identity = _conv(x, P[f'{pre}.downsample.0'], stride=s)
            x = _round(pre, F.relu(out + identity))       # one rounding per block: conv3 (+ shortcut) epilogue
            if taps is not None:
                taps[pre] = x
        pyr.append(x)
    return pyr  # [p1, p2, p3, p4, p5]


def regnet_forward(P, x, layout, taps=None):
    """encoders/regnet.py:160-166 (RegNet.forward) with Stem (:38-49), BottleneckBlock (:79-97) and Bottleneck
    (:51-77); BN folded.  ``layout``: w_stem / widths / depths / groups / use_se [/ strides] (weights.regnet_cfg); the
    stride of each stage's first block is 2 (RegNetConfig.strides, regnet.py:170) -- the model's ``stage4_stride`` never
    reaches a RegNet built by name (weights.regnet_stage_strides).  The squeeze-excite
    gate pools over a 1 x 1 window (blocks.py:38: nn.AvgPool2d((1, 1)) -- the identity), so it is a per-PIXEL gate
    x * sigmoid(W2 relu(W1 x)): restated as the reference computes it, not as the paper defines it."""
    x = _conv(x, P['encoder.stem.cbr.0'], stride=2, padding=1, relu=True)
    pyr = [x]
    if taps is not None:
        taps['stem'] = x
    strides = layout.get('strides', [2, 2, 2, 2])
    for si, (d, g) in enumerate(zip(layout['depths'], layout['groups']), start=1):
        for b in range(1, d + 1):
            pre = f'encoder.stage{si}.block{b}'
            s = strides[si - 1] if b == 1 else 1
            out = _conv(x, P[f'{pre}.bottleneck.a.0'], relu=True)
            out = _conv(out, P[f'{pre}.bottleneck.b.0'], stride=s, padding=1, groups=g, relu=True)
            if layout['use_se']:
                gate = _conv(out, P[f'{pre}.bottleneck.se.se.0'], relu=True)
                out = out * torch.sigmoid(_conv(gate, P[f'{pre}.bottleneck.se.se.2']))
            out = _conv(out, P[f'{pre}.bottleneck.c.0'])
            short = _conv(x, P[f'{pre}.downsample.conv.0'], stride=s) if f'{pre}.downsample.conv.0' in P else x
            x = F.relu(short + out)
            if taps is not None:
                taps[pre] = x
        pyr.append(x)
    return pyr  # [stem, stage1, stage2, stage3, stage4]


def encoder_forward(P, x, cfg, output_stride, taps=None):
    """the pyramid of ``cfg``'s encoder: index 0 is never read by a decoder, 1..4 are the stage outputs"""
    if str(cfg.get('encoder', 'resnet50')).startswith('regnet'):
        if _EMU is not None:
            raise NotImplementedError('no format emulation of the fp16 engine exists for RegNet encoders (their fp16 layers are checked '
                                      'teacher-forced, tests/test_gpu_regnet.py)')
        return regnet_forward(P, x, cfg['regnet'], taps)
    return resnet50_forward(P, x, output_stride, taps)


def aspp_forward(P, pre, x, rates, taps=None):
    """decoders/aspp.py:96-102 (+ ASPPPooling.forward :45-48)."""
    res = [_conv(x, P[f'{pre}.convs.0.0'], relu=True, site=True)]
    for i, r in enumerate(rates, start=1):
        res.append(_conv(x, P[f'{pre}.convs.{i}.0'], padding=r, dilation=r, relu=True, site=True))
    if taps is not None:
        taps[f'{pre}.cat'] = torch.cat(res, dim=1)
    size = x.shape[-2:]
    pooled = F.adaptive_avg_pool2d(x, 1)
    pooled = _conv(pooled, P[f'{pre}.convs.4.aspp_pooling.1'], relu=True, w32=True)
    if _EMU is not None:
        # the engine's form (pdl_net.hip): the pooled branch is a per-image constant, its slice of the projection
        # stays fp32 and enters as a per-image bias; the other four branches go through the fp16 GEMM
        w, b = P[f'{pre}.project.0']
        c4 = sum(r.shape[1] for r in res)
        y = F.conv2d(torch.cat(res, dim=1), _EMU.w(w)[:, :c4], _t(b)) + F.conv2d(pooled, _t(w)[:, c4:])
        return _EMU.a(f'{pre}.project.0', F.relu(y))
    res.append(F.interpolate(pooled, size=size, mode='bilinear', align_corners=True))
    return _conv(torch.cat(res, dim=1), P[f'{pre}.project.0'], relu=True)


def decoder_forward(P, pre, pyr, low_level_stages, rates, taps=None, ins_decoder=True):
    """decoders/panoptic_deeplab.py:68-80."""
    x = aspp_forward(P, f'{pre}.aspp', pyr[-1], rates, taps)
    if taps is not None:
        taps[f'{pre}.aspp'] = x
    for i, stage in enumerate(low_level_stages):
        l = _conv(pyr[stage], P[f'{pre}.project.{i}.0'], relu=True, site=True)
        x = _round(f'bilinear:{pre}.{i}', F.interpolate(x, size=l.shape[2:], mode='bilinear', align_corners=True))
        x = torch.cat((x, l), dim=1)
        if taps is not None:
            taps[f'{pre}.stage{i}.cat'] = x
        # sepconv_precise.hip: fp32 taps, the depthwise result as an fp16 hi + lo pair (not rounded), fp16 pointwise
        # weights; sepconv.hip: fp16 taps, fp16 depthwise result, fp16 weights
        prec = precise_block(f'{pre}.fuse.{i}.0', ins_decoder)
        x = _conv(x, (P[f'{pre}.fuse.{i}.0.sepconv.0'][0], None), padding=2, groups=x.shape[1], site=None if prec else True, w32=prec)
        x = _conv(x, P[f'{pre}.fuse.{i}.0.sepconv.1'], relu=True, site=True)
    return x


def head_forward(P, pre, x, ins_decoder=True, bifpn=False):
    """heads.py:12-19."""
    prec = precise_block(f'{pre}.head.0.0', ins_decoder, bifpn) and P[f'{pre}.head.1'][0].shape[0] <= 2
    x = _conv(x, (P[f'{pre}.head.0.0.sepconv.0'][0], None), padding=2, groups=x.shape[1], site=None if prec else True, w32=prec)
    x = _conv(x, P[f'{pre}.head.0.0.sepconv.1'], relu=True)      # fused head: this map stays fp32 on chip
    return _conv(x, P[f'{pre}.head.1'], w32=True)


def calculate_uncertainty(logits):
    """point_rend.py:11-31."""
    if logits.size(1) == 1:
        return -(torch.abs(logits))
    top2 = torch.topk(logits, k=2, dim=1)[0]
    return (top2[:, 1] - top2[:, 0]).unsqueeze(1)


def point_sample(features, point_coords):
    """point_rend.py:33-60 (bilinear, align_corners=False, zero padding)."""
    out = F.grid_sample(features, 2.0 * point_coords.unsqueeze(2) - 1.0, mode='bilinear', align_corners=False)
    return out.squeeze(3)


def uncertain_points_on_grid(uncertainty_map, num_points):
    """point_rend.py:108-137."""
    R, _, H, W = uncertainty_map.shape
    h_step = 1.0 / float(H)
    w_step = 1.0 / float(W)
    num_points = min(H * W, num_points)
    idx = torch.topk(uncertainty_map.view(R, H * W), k=num_points, dim=1)[1]
    coords = torch.zeros(R, num_points, 2, dtype=torch.float)
    coords[:, :, 0] = 0.5 * w_step + w_step * (idx % W).float()
    coords[:, :, 1] = 0.5 * h_step + h_step * torch.div(idx, W, rounding_mode='floor').float()
    return idx, coords


def point_head_forward(P, fine, coarse, num_fc):
    """point_rend.py:181-188 (Conv1d k=1 MLP, coarse concatenated at every layer)."""
    if _EMU is not None:   # the engine's MLP input rows are fp16: sampled features and the coarse logits beside them
        fine, coarse = _EMU.a('pr.features', fine), _EMU.a('pr.coarse', coarse)
    x = torch.cat([fine, coarse], dim=1)
    for k in range(num_fc):
        w, b = P[f'semantic_pr.point_head.fc_layers.{k}.0']
        wt = _t(w) if _EMU is None else _EMU.w(w)
        x = _round(f'semantic_pr.point_head.fc_layers.{k}.0', F.relu(F.conv1d(x, wt, _t(b))))
        x = torch.cat([x, coarse], dim=1)
    w, b = P['semantic_pr.point_head.predictor']
    return F.conv1d(x, _t(w), _t(b))       # fp32 weights in the engine (head1x1)


def point_rend_forward(P, coarse_logits, features, steps, num_points, num_fc, taps=None):
    """point_rend.py:241-269 (eval branch)."""
    sem = coarse_logits.clone()
    for s in range(steps):
        sem = F.interpolate(sem, scale_factor=2.0, mode='bilinear', align_corners=False)
        unc = calculate_uncertainty(sem)
        idx, coords = uncertain_points_on_grid(unc, num_points)
        coarse_pts = point_sample(coarse_logits, coords)
        fine_pts = point_sample(features, coords)
        logits = point_head_forward(P, fine_pts, coarse_pts, num_fc)
        N, C, H, W = sem.shape
        if taps is not None:
            taps[f'pr.step{s}.idx'] = idx
            taps[f'pr.step{s}.point_logits'] = logits
        sem = sem.reshape(N, C, H * W).scatter_(2, idx.unsqueeze(1).expand(-1, C, -1), logits).view(N, C, H, W)
    return sem


@torch.no_grad()
def pdl_forward(P, x, cfg, render_steps=2, interpolate_ins=True, taps=None, emu=None):
    """QuantizablePanopticDeepLabPR.forward, eval (quantization/panoptic_deeplab.py:238-250).

    P: folded params (numpy), x: (N,1,H,W) fp32 torch tensor, H,W % 16 == 0.
    Returns dict(sem_logits, ctr_hmp, offsets) of fp32 torch tensors.
    emu: None = the reference's fp32 forward; an Fp16Emu = the same forward with the HIP engine's
    fp16 storage roundings (class docstring).
    """
    global _EMU
    if emu is not None:
        _EMU = emu.bind(P)
        try:
            return pdl_forward(P, x, cfg, render_steps, interpolate_ins, taps, None)
        finally:
            _EMU = None
    pyr = encoder_forward(P, x, cfg, cfg['stage4_stride'], taps)
    stages, rates = cfg['low_level_stages'], cfg['atrous_rates']
    insd = bool(cfg['ins_decoder'])
    semantic_x = decoder_forward(P, 'semantic_decoder', pyr, stages, rates, taps, insd)
    instance_x = decoder_forward(P, 'instance_decoder', pyr, stages, rates, taps, insd) if insd else semantic_x
    sem = head_forward(P, 'semantic_head', semantic_x, insd)
    ctr = head_forward(P, 'ins_center', instance_x, insd)
    off = head_forward(P, 'ins_xy', instance_x, insd)
    if taps is not None:
        taps.update(semantic_x=semantic_x, instance_x=instance_x, sem_coarse=sem)
    sem_logits = point_rend_forward(P, sem, semantic_x, render_steps,
                                    cfg['subdivision_num_points'], cfg['num_fc'], taps)
    if interpolate_ins:  # Interpolate2d(4, bilinear, align_corners=True), panoptic_deeplab.py:89,233-234
        ctr = F.interpolate(ctr, scale_factor=4.0, mode='bilinear', align_corners=True)
        off = F.interpolate(off, scale_factor=4.0, mode='bilinear', align_corners=True)
    return {'sem_logits': sem_logits, 'ctr_hmp': ctr, 'offsets': off}


# ----------------------------------------------------------------------------
# PanopticBiFPN(PR)  (models/panoptic_bifpn.py, decoders/bifpn.py)
# ----------------------------------------------------------------------------
def _silu(x):
    return x * torch.sigmoid(x)


def _fusion_weights(P, name, eps=1e-4):
    """bifpn.py:52-55 / 106-109: relu, then divide by (sum + eps)."""
    w = F.relu(_t(P[name][0]))
    return w / (w.sum() + eps)


def _sep3(P, pre, x, ins_decoder=True):
    """shared after_combine block: depthwise 3x3 + pointwise (+folded BN) + SiLU (bifpn.py:35,91).  Format emulation: the
    engine rounds the depthwise result to fp16 unless the node runs on sepconv_precise.hip (``precise_node``), and the
    node's output once, after the SiLU."""
    cout = P[f'{pre}.after_combines.0.0.sepconv.1'][0].shape[0]
    prec = precise_node(pre, ins_decoder, x.shape[1], cout)
    x = _conv(x, (P[f'{pre}.after_combines.0.0.sepconv.0'][0], None), padding=1, groups=x.shape[1],
              site=None if prec else True, w32=prec)
    return _round(f'{pre}.after_combines.0.0.sepconv.1', _silu(_conv(x, P[f'{pre}.after_combines.0.0.sepconv.1'])))


def _resample(P, pre, i, x):
    k = f'{pre}.resamplings.{i}.conv.0'
    return _conv(x, P[k], site=True) if k in P else x          # Resample2d is the identity when nin == fpn_dim (blocks.py:62-67)


def bifpn_layer(P, pre, feats, eps=1e-4, ins_decoder=True):
    """BiFPNLayer.forward (bifpn.py:147-156): feats = [P3..P7] -> [P3'..P7'].  (Format emulation: the engine writes every
    fused map -- the input of a node's separable conv -- to HBM in fp16: sites ``<pre>.<dir>.fuse<i>``.)"""
    # top-down over [P7, P6, P5, P4, P3]
    rev = feats[::-1]
    w = _fusion_weights(P, f'{pre}.top_down_fpn.weights')
    td = [rev[0]]
    for i in range(4):
        hi = _resample(P, f'{pre}.top_down_fpn', i, rev[i + 1])
        up = F.interpolate(td[-1], scale_factor=2.0, mode='nearest')
        fused = _round(f'{pre}.top_down_fpn.fuse{i}', (w[i] * up + w[i + 1] * hi) / (w[i] + w[i + 1] + eps))
        td.append(_sep3(P, f'{pre}.top_down_fpn', fused, ins_decoder))
    tdr = td[::-1]                                   # [P3', P4', P5', P6', P7]
    w = _fusion_weights(P, f'{pre}.bottom_up_fpn.weights')
    pyr = feats[1:]                                  # [P4, P5, P6, P7]
    bu = [tdr[0]]
    for i in range(4):
        down = F.max_pool2d(bu[-1], 3, stride=2, padding=1)
        lo = _resample(P, f'{pre}.bottom_up_fpn', i, pyr[i])
        if i < 3:
            fused = (w[i] * down + w[i + 1] * lo + w[i + 2] * tdr[i + 1]) / (w[i] + w[i + 1] + w[i + 2] + eps)
        else:
            fused = (w[i] * down + w[i + 1] * lo) / (w[i] + w[i + 1] + eps)
        bu.append(_sep3(P, f'{pre}.bottom_up_fpn', _round(f'{pre}.bottom_up_fpn.fuse{i}', fused), ins_decoder))
    return bu


def bifpn_forward_decoder(P, dec, pyr345, p2f, n_layers, taps=None, ins_decoder=True):
    """BiFPN.forward (bifpn.py:185-196) + BiFPNDecoder.forward (:226-236)."""
    fp = f'{dec}_fpn'
    p6 = F.max_pool2d(_conv(pyr345[-1], P[f'{fp}.p6_resample.conv.0'], site=True), 3, stride=2, padding=1)
    p7 = F.max_pool2d(p6, 3, stride=2, padding=1)
    feats = list(pyr345) + [p6, p7]
    for li in range(n_layers):
        feats = bifpn_layer(P, f'{fp}.bifpns.{li}', feats, ins_decoder=ins_decoder)
        if taps is not None:
            taps[f'{fp}.layer{li}.P3'] = feats[0]
    seq = ([p2f] + feats)[::-1]                      # [P7, P6, P5, P4, P3, P2]
    x = seq[0]
    for i in range(5):
        w, b = P[f'{dec}_decoder.upsamplings.{i}.0']
        wt = _t(w) if _EMU is None else _EMU.w(w)
        x = _round(f'{dec}_decoder.upsamplings.{i}.0', F.relu(F.conv_transpose2d(x, wt, _t(b), stride=2)))
        x = torch.cat([x, seq[i + 1]], dim=1)
    prec = precise_block(f'{dec}_decoder.fusion.0', ins_decoder)
    x = _conv(x, (P[f'{dec}_decoder.fusion.0.sepconv.0'][0], None), padding=2, groups=x.shape[1], site=None if prec else True, w32=prec)
    return _conv(x, P[f'{dec}_decoder.fusion.0.sepconv.1'], relu=True, site=True)


@torch.no_grad()
def bifpn_forward(P, x, cfg, render_steps=2, interpolate_ins=True, taps=None, emu=None):
    """QuantizablePanopticBiFPNPR.forward, eval (quantization/panoptic_bifpn.py:147-161).  emu: as ``pdl_forward``."""
    global _EMU
    if emu is not None:
        _EMU = emu.bind(P)
        try:
            return bifpn_forward(P, x, cfg, render_steps, interpolate_ins, taps, None)
        finally:
            _EMU = None
    insd = bool(cfg['ins_decoder'])
    pyr = encoder_forward(P, x, cfg, 32, taps)          # the BiFPN models build their encoder at its default output stride 32
    p2f = _conv(pyr[1], P['p2_resample.conv.0'], site=True)
    nl = cfg['fpn_layers']
    semantic_x = bifpn_forward_decoder(P, 'semantic', pyr[2:], p2f, nl, taps, insd)
    instance_x = bifpn_forward_decoder(P, 'instance', pyr[2:], p2f, nl, taps, insd) if insd else semantic_x
    sem = head_forward(P, 'semantic_head', semantic_x, insd, True)
    ctr = head_forward(P, 'ins_center', instance_x, insd, True)
    off = head_forward(P, 'ins_xy', instance_x, insd, True)
    if taps is not None:
        taps.update(semantic_x=semantic_x, instance_x=instance_x, sem_coarse=sem)
    sem_logits = point_rend_forward(P, sem, semantic_x, render_steps, cfg['subdivision_num_points'], cfg['num_fc'], taps)
    if interpolate_ins:
        ctr = F.interpolate(ctr, scale_factor=4.0, mode='bilinear', align_corners=True)
        off = F.interpolate(off, scale_factor=4.0, mode='bilinear', align_corners=True)
    return {'sem_logits': sem_logits, 'ctr_hmp': ctr, 'offsets': off}


def model_forward(P, x, cfg, render_steps=2, interpolate_ins=True, taps=None, emu=None):
    if 'BiFPN' in cfg.get('arch', ''):
        return bifpn_forward(P, x, cfg, render_steps, interpolate_ins, taps, emu)
    return pdl_forward(P, x, cfg, render_steps, interpolate_ins, taps, emu)


# ----------------------------------------------------------------------------
# teacher-forced layer check (test infrastructure for the HIP engine's full-size parity)
# ----------------------------------------------------------------------------
@torch.no_grad()
def teacher_forced_layers(P, cfg, x, tap):
    """Every layer of the Panoptic-DeepLab forward evaluated ON THE ENGINE'S OWN INPUTS.

    Two fp16 pipelines whose fp32 sums are merely ordered differently drift apart layer after layer (one flipped
    last bit perturbs every output that reads it, which flips more bits: after ~10 layers 40 % of the elements
    differ by an ulp and the end-to-end distance between the pipelines equals their distance to fp32 --
    tools/layer_parity.py), so an end-to-end comparison cannot separate kernel error from format error.  Layer by
    layer it can: ``tap(name)`` returns the engine's map ``name`` as an (N,C,H,W) fp32 tensor; this generator
    recomputes each layer from the engine's INPUT map(s) in fp32 with the engine's fp16-rounded weights and yields
    ``(output tap name, expected fp32 value before the output rounding, rounds_to_fp16)``.  A correct kernel equals
    the expectation rounded to fp16 except where its fp32 sum, accumulated in another order, falls on the other side
    of a rounding boundary (one ulp, a fraction of a percent of the elements).
    Covers encoder, ASPP, decoder stage(s) and the three heads (heads.py:12-19); PointRend is checked on identical
    inputs by ``point_rend_forward`` directly."""
    r16 = Fp16Emu.r16

    def W(name, fp32=False):
        w, b = P[name]
        return (_t(w) if fp32 else r16(_t(w))), _t(b)

    def conv(xin, name, stride=1, padding=0, dilation=1, groups=1):
        w, b = W(name)
        return F.conv2d(xin, w, b if groups == 1 else None, stride, padding, dilation, groups)

    pyr = yield from _tf_encoder(P, cfg, x, tap)
    p5 = tap(pyr[4])
    rates = cfg['atrous_rates']
    decs = ['semantic_decoder'] + (['instance_decoder'] if cfg['ins_decoder'] else [])
    for d in decs:
        a = f'{d}.aspp'
        res = [F.relu(conv(p5, f'{a}.convs.0.0'))]
        for i, r in enumerate(rates, start=1):
            res.append(F.relu(conv(p5, f'{a}.convs.{i}.0', 1, r, r)))
        yield f'{a}.cat', torch.cat(res, dim=1), True
        w, b = W(f'{a}.convs.4.aspp_pooling.1', fp32=True)
        pooled = F.relu(F.conv2d(F.adaptive_avg_pool2d(p5, 1), w, b))
        w, b = P[f'{a}.project.0']
        cat = tap(f'{a}.cat')
        c4 = cat.shape[1]
        y = F.conv2d(cat, r16(_t(w))[:, :c4], _t(b)) + F.conv2d(pooled, _t(w)[:, c4:])
        yield a, F.relu(y), True
        xn = a
        for i, stage in enumerate(cfg['low_level_stages']):
            low = F.relu(conv(tap(pyr[stage]), f'{d}.project.{i}.0'))
            up = F.interpolate(tap(xn), size=low.shape[2:], mode='bilinear', align_corners=True)
            yield f'{d}.stage{i}.cat', torch.cat((up, low), dim=1), True
            cat = tap(f'{d}.stage{i}.cat')[:, :up.shape[1] + low.shape[1]]     # the engine pads the concat to 64 channels
            prec = precise_block(f'{d}.fuse.{i}.0', bool(cfg['ins_decoder']))
            yield f'{d}.stage{i}.out', F.relu(_tf_sepconv(P, cat, f'{d}.fuse.{i}.0', 2, prec)), True
            xn = f'{d}.stage{i}.out'
    last = len(cfg['low_level_stages']) - 1
    semx = tap(f'semantic_decoder.stage{last}.out')
    insx = tap(f'instance_decoder.stage{last}.out') if cfg['ins_decoder'] else semx
    yield from _tf_heads(P, semx, insx, tap, bool(cfg['ins_decoder']))


def _tf_weights(P, name, fp32=False, split=False):
    """the layer's weights as the engine holds them: fp16, fp32, or (split) an fp16 hi + lo pair"""
    w, b = P[name]
    t = _t(w)
    if split:
        hi = Fp16Emu.r16(t)
        return hi + Fp16Emu.r16(t - hi), _t(b)
    return (t if fp32 else Fp16Emu.r16(t)), _t(b)


def _tf_conv(P, xin, name, stride=1, padding=0, dilation=1, groups=1, fp32=False, split=False):
    w, b = _tf_weights(P, name, fp32, split)
    return F.conv2d(xin, w, b if groups == 1 else None, stride, padding, dilation, groups)


def _tf_sepconv(P, xin, pre, pad, precise, wsplit=False):
    """``pre``.sepconv.0 (depthwise) -> ``pre``.sepconv.1 (pointwise + folded BN) before the activation.  precise
    (sepconv_precise.hip): fp32 taps and the depthwise result carried as an fp16 hi + lo pair -- an exact depthwise half
    on the engine's fp16 input map -- with fp16 pointwise weights.  Otherwise (sepconv.hip, or the unfused dwconv + conv
    pair, which are bit-identical): fp16 taps, fp16 intermediate map, fp16 pointwise weights."""
    dw = _tf_conv(P, xin, f'{pre}.sepconv.0', 1, pad, 1, xin.shape[1], fp32=precise)
    if not precise:
        dw = Fp16Emu.r16(dw)
    return _tf_conv(P, dw, f'{pre}.sepconv.1', split=precise and wsplit and P[f'{pre}.sepconv.1'][0].shape[0] == 128)


def _tf_encoder(P, cfg, x, tap):
    """ResNet-50 part of the teacher-forced check (both network families); returns the pyramid's tap names."""
    def conv(xin, name, stride=1, padding=0, dilation=1, groups=1):
        return _tf_conv(P, xin, name, stride, padding, dilation, groups)

    w, b = _tf_weights(P, 'encoder.conv1', fp32=True)
    yield 'p1', F.max_pool2d(F.relu(F.conv2d(x, w, b, 2, 3)), 3, 2, 1), True
    xname = 'p1'
    pyr = ['p1']
    for li, nblocks in enumerate(RESNET50_LAYERS, start=1):
        stride = 1 if li == 1 else 2
        dil = 1
        if li == 4 and cfg.get('stage4_stride', 32) == 16:
            stride, dil = 1, 2
        for bidx in range(nblocks):
            pre = f'encoder.layer{li}.{bidx}'
            s = stride if bidx == 0 else 1
            xin = tap(xname)
            yield pre + '.c1', F.relu(conv(xin, pre + '.conv1')), True
            yield pre + '.c2', F.relu(conv(tap(pre + '.c1'), pre + '.conv2', s, dil, dil)), True
            idn = conv(xin, pre + '.downsample.0', s) if bidx == 0 else xin
            yield pre, F.relu(conv(tap(pre + '.c2'), pre + '.conv3') + idn), True
            xname = pre
        pyr.append(xname)
    return pyr


def _tf_heads(P, semx, insx, tap, ins_decoder=True, wsplit=False, bifpn=False):
    """heads.py:12-19 on the engine's decoder outputs: 5x5 separable conv + ReLU + fp32 1x1.  The engine fuses the
    whole head into one launch when it has at most two output planes (the 256-channel map then never leaves the CU,
    fp32); a wider head (multi-class semantic) stores that map in fp16 as ``<head>.pw`` and runs the 1x1 from it."""
    for head, xin in (('semantic_head', semx), ('ins_center', insx), ('ins_xy', insx)):
        w, b = _tf_weights(P, f'{head}.head.1', fp32=True)
        prec = precise_block(f'{head}.head.0.0', ins_decoder, bifpn) and w.shape[0] <= 2
        y = F.relu(_tf_sepconv(P, xin, f'{head}.head.0.0', 2, prec, wsplit))
        if w.shape[0] > 2:
            yield head + '.pw', y, True
            y = tap(head + '.pw')
        yield head + '.out', F.conv2d(y, w, b), False


@torch.no_grad()
def teacher_forced_layers_bifpn(P, cfg, x, tap):
    """The PanopticBiFPN forward (bifpn.py:185-236, panoptic_bifpn.py:70-82) layer by layer ON THE ENGINE'S OWN INPUTS,
    same contract as ``teacher_forced_layers``.  Engine tap names (pdl_net.hip): ``p2f``; per decoder ``{d}_fpn.p6pre``,
    ``.in.P6``, ``.in.P7`` (max-pools: exact), per BiFPN layer ``li`` and level ``P{3..7}`` the resampled inputs ``.rtd`` /
    ``.rbu`` (layer 0, P3..P5), the fused maps ``.fuse`` (top-down node) / ``.fuseb`` (bottom-up node) and the node
    outputs ``.td`` / ``.bu``; ``{d}_decoder.cat{i}`` and ``.out``; then the three heads."""
    r16 = Fp16Emu.r16

    def conv(xin, name, stride=1, padding=0, dilation=1, groups=1):
        return _tf_conv(P, xin, name, stride, padding, dilation, groups)

    def sep3(pre, fused_name):
        fz = tap(fused_name)
        Fc = P[f'{pre}.after_combines.0.0.sepconv.0'][0].shape[0]
        if fz.shape[1] == 2 * Fc:          # the engine's fused map as channels [hi | lo] (pdl_net.hip fsplit_on)
            fz = fz[:, :Fc] + fz[:, Fc:]
        cout = P[f'{pre}.after_combines.0.0.sepconv.1'][0].shape[0]
        prec = precise_node(pre, bool(cfg['ins_decoder']), fz.shape[1], cout)
        dw = _tf_conv(P, fz, f'{pre}.after_combines.0.0.sepconv.0', 1, 1, 1, fz.shape[1], fp32=prec)
        if not prec:
            dw = r16(dw)
        return _silu(_tf_conv(P, dw, f'{pre}.after_combines.0.0.sepconv.1', split=prec and ws and cout == 128))

    ws = wsplit_on()
    cdec = 'instance' if cfg['ins_decoder'] else 'semantic'
    pyr = yield from _tf_encoder(P, cfg, x, tap)
    yield 'p2f', conv(tap(pyr[1]), 'p2_resample.conv.0'), True
    eps = 1e-4
    decs = ['semantic'] + (['instance'] if cfg['ins_decoder'] else [])
    for d in decs:
        fp = f'{d}_fpn'
        yield f'{fp}.p6pre', conv(tap(pyr[4]), f'{fp}.p6_resample.conv.0'), True
        yield f'{fp}.in.P6', F.max_pool2d(tap(f'{fp}.p6pre'), 3, 2, 1), True
        yield f'{fp}.in.P7', F.max_pool2d(tap(f'{fp}.in.P6'), 3, 2, 1), True
        feat = [pyr[2], pyr[3], pyr[4], f'{fp}.in.P6', f'{fp}.in.P7']              # P3..P7
        for li in range(cfg['fpn_layers']):
            L, pre = f'{fp}.l{li}', f'{fp}.bifpns.{li}'
            dp = f'{pre}.top_down_fpn'
            w = _fusion_weights(P, f'{dp}.weights')
            td_prev = feat[4]
            for i in range(4):
                lv = 3 - i
                q = f'{L}.P{3 + lv}'
                hi = feat[lv]
                if f'{dp}.resamplings.{i}.conv.0' in P:
                    yield q + '.rtd', conv(tap(feat[lv]), f'{dp}.resamplings.{i}.conv.0'), True
                    hi = q + '.rtd'
                up = F.interpolate(tap(td_prev), scale_factor=2.0, mode='nearest')
                fused = (w[i] * up + w[i + 1] * tap(hi)) / (w[i] + w[i + 1] + eps)
                yield q + '.fuse', fused, True
                yield q + '.td', sep3(dp, q + '.fuse'), True
                td_prev = q + '.td'
            dp = f'{pre}.bottom_up_fpn'
            w = _fusion_weights(P, f'{dp}.weights')
            bu_prev = f'{L}.P3.td'
            new = [bu_prev]
            for i in range(4):
                lv = i + 1
                q = f'{L}.P{3 + lv}'
                lo = feat[lv]
                if f'{dp}.resamplings.{i}.conv.0' in P:
                    yield q + '.rbu', conv(tap(feat[lv]), f'{dp}.resamplings.{i}.conv.0'), True
                    lo = q + '.rbu'
                down = F.max_pool2d(tap(bu_prev), 3, 2, 1)
                if i < 3:
                    fused = (w[i] * down + w[i + 1] * tap(lo) + w[i + 2] * tap(q + '.td')) / (w[i] + w[i + 1] + w[i + 2] + eps)
                else:
                    fused = (w[i] * down + w[i + 1] * tap(lo)) / (w[i] + w[i + 1] + eps)
                yield q + '.fuseb', fused, True
                yield q + '.bu', sep3(dp, q + '.fuseb'), True
                bu_prev = q + '.bu'
                new.append(bu_prev)
            feat = new
        dp = f'{d}_decoder'
        skips = [feat[3], feat[2], feat[1], feat[0], 'p2f']
        xn = feat[4]
        for i in range(5):
            w, b = _tf_weights(P, f'{dp}.upsamplings.{i}.0', split=ws and d == cdec)
            up = F.relu(F.conv_transpose2d(tap(xn), w, b, stride=2))     # cat{i-1} carries 2F channels, all of them inputs
            yield f'{dp}.cat{i}', torch.cat([up, tap(skips[i])], dim=1), True
            xn = f'{dp}.cat{i}'
        yield f'{dp}.out', F.relu(_tf_sepconv(P, tap(xn), f'{dp}.fusion.0', 2, precise_block(f'{dp}.fusion.0', bool(cfg['ins_decoder'])), ws)), True
    semx = tap('semantic_decoder.out')
    insx = tap('instance_decoder.out') if cfg['ins_decoder'] else semx
    yield from _tf_heads(P, semx, insx, tap, bool(cfg['ins_decoder']), ws, True)
