"""Parameter bookkeeping for the Panoptic-DeepLab(PointRend) hot path.

Three jobs, all host side (numpy / torch CPU tensors only):

* ``pdl_spec(cfg)``       -- the ordered list of learnable layers of
  ``QuantizablePanopticDeepLabPR`` (reference:
  empanada/models/quantization/panoptic_deeplab.py:35-98,148-168,
  encoders/resnet.py:143-215, decoders/panoptic_deeplab.py:25-62,
  decoders/aspp.py:51-94, heads.py:10-16, point_rend.py:146-176) with the
  module paths the reference's ``state_dict()`` uses.
* ``seeded_state_dict``   -- deterministic random-init weights in the
  reference's *unfused* key layout (the real MitoNet weights live on Zenodo;
  benchmarks and parity tests use these, the reference model can
  ``load_state_dict`` them verbatim).
* ``fold_state_dict``     -- accepts either the unfused layout or the layout
  of the exported TorchScript model (``fuse_model()`` output,
  empanada_napari/_train.py:59-73) and returns one folded fp32
  ``(weight, bias)`` pair per convolution: BatchNorm (eval) is folded into the
  preceding convolution, which is what the HIP engine consumes.
"""
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-5

# MitoNet_v1-class configuration (empanada_napari/training/pdl_model.yaml:1-20,
# empanada_napari/configs/MitoNet_v1.yaml).
MITONET_PDL_CFG = dict(
    arch='PanopticDeepLabPR', encoder='resnet50', num_classes=1,
    stage4_stride=16, decoder_channels=256, low_level_stages=[1],
    low_level_channels_project=[32], atrous_rates=[2, 4, 6],
    aspp_channels=None, ins_decoder=True, ins_ratio=0.5, num_fc=3,
    subdivision_num_points=8192,
)

# MitoNet_v1_mini-class configuration (empanada_napari/training/bifpn_model.yaml:1-15)
MITONET_MINI_CFG = dict(
    arch='PanopticBiFPNPR', encoder='resnet50', num_classes=1, fpn_dim=128, fpn_layers=3,
    ins_decoder=True, depthwise=True, num_fc=3, subdivision_num_points=8192,
)

RESNET50_LAYERS = (3, 4, 6, 3)
RESNET_PLANES = (64, 128, 256, 512)


def _L(name, shape, bn=None, bias=False, kind='conv', gain=1.0, bn_gamma=1.0,
       zero_mean=False, bias_mean=0.0):
    return dict(name=name, shape=tuple(shape), bn=bn, bias=bias, kind=kind,
                gain=gain, bn_gamma=bn_gamma, zero_mean=zero_mean, bias_mean=bias_mean)


def resnet50_spec(prefix='encoder', in_channels=1):
    """Layers of the 1-channel ResNet50 (encoders/resnet.py:143-215)."""
    layers = [_L(f'{prefix}.conv1', (64, in_channels, 7, 7), bn=f'{prefix}.bn1')]
    inplanes = 64
    for li, (nblocks, planes) in enumerate(zip(RESNET50_LAYERS, RESNET_PLANES), start=1):
        for b in range(nblocks):
            p = f'{prefix}.layer{li}.{b}'
            layers.append(_L(f'{p}.conv1', (planes, inplanes, 1, 1), bn=f'{p}.bn1'))
            layers.append(_L(f'{p}.conv2', (planes, planes, 3, 3), bn=f'{p}.bn2'))
            # a small last-BN gamma keeps the residual stream of a random-init
            # net O(1) (cf. zero_init_residual, resnet.py:184-192)
            layers.append(_L(f'{p}.conv3', (planes * 4, planes, 1, 1), bn=f'{p}.bn3', bn_gamma=0.35))
            if b == 0:
                layers.append(_L(f'{p}.downsample.0', (planes * 4, inplanes, 1, 1),
                                 bn=f'{p}.downsample.1', bn_gamma=0.7))
            inplanes = planes * 4
    return layers


# the two RegNets the reference can export (quantization/encoders/__init__.py; regnet.py:266-271,296-301):
# (depth, w_0, w_a, w_m, group width, squeeze-excite)
REGNET_PARAMS = {
    'regnetx_6p4gf': (17, 184, 60.83, 2.07, 56, False),
    'regnety_6p4gf': (25, 112, 33.22, 2.27, 72, True),
}
REGNET_STEM = 32      # RegNetConfig.w_stem, regnet.py:168


def regnet_layout(depth, w_0, w_a, w_m, group_w, use_se, q=8):
    """Stage widths / depths / groups of a RegNet from its generating parameters (regnet.py:191-260, after
    arXiv:2003.13678 eqs 2-4): block widths on the line w_0 + j w_a, snapped to w_0 w_m^s, rounded to multiples of q and
    grouped into stages; bottleneck ratio 1, so a width only has to be a multiple of its group width."""
    u = w_0 + np.arange(depth) * w_a
    s_ = np.round(np.log(u / w_0) / np.log(w_m))
    w = (q * np.round(w_0 * np.power(w_m, s_) / q)).astype(int)
    widths, depths = np.unique(w, return_counts=True)
    if len(widths) != 4:
        raise ValueError('RegNet parameters must give four stages')
    out_w, groups = [], []
    for wi in widths.tolist():
        gw = min(int(group_w), int(wi))
        wi = max(gw, int(gw * round(wi / gw)))
        out_w.append(wi)
        groups.append(wi // gw)
    return dict(w_stem=REGNET_STEM, widths=out_w, depths=[int(d) for d in depths], groups=groups, use_se=bool(use_se))


def is_regnet(cfg):
    return str((cfg or {}).get('encoder', 'resnet50')).startswith('regnet')


def regnet_cfg(cfg):
    """the RegNet layout of ``cfg``: explicit (``cfg['regnet']``, what ``infer_cfg`` reads off an export) or derived from the
    encoder's name"""
    if cfg.get('regnet'):
        return cfg['regnet']
    name = cfg['encoder']
    if name not in REGNET_PARAMS:
        raise NotImplementedError(f'encoder {name!r}: only {sorted(REGNET_PARAMS)} and resnet50 are built')
    return regnet_layout(*REGNET_PARAMS[name])


def encoder_widths(cfg):
    """``encoder.cfg.widths``: channels of pyramid levels 1..4 (resnet.py:207-208, regnet.py:244-245)"""
    if is_regnet(cfg):
        return list(regnet_cfg(cfg)['widths'])
    return [p * 4 for p in RESNET_PLANES]


def regnet_stage_strides(cfg):
    """stride of each stage's first block: RegNetConfig.strides = [2, 2, 2, 2] (regnet.py:170).  ``stage4_stride`` does NOT
    reach a RegNet built by name: the constructors hand ``output_stride`` to RegNetConfig (where it becomes an unused
    attribute, regnet.py:262-271,296-301) and RegNet.__init__ keeps its default 32 -- a PanopticDeepLab on a RegNet runs
    its ASPP at stride 32 whatever the YAML says.  An export built differently carries its strides in the scripted
    convolutions; ``infer_cfg`` stores them under ``cfg['regnet']['strides']``."""
    return list(regnet_cfg(cfg).get('strides', [2, 2, 2, 2]))


def regnet_spec(cfg, prefix='encoder', in_channels=1):
    """Layers of the RegNet encoder (encoders/regnet.py:38-160): 3x3 stride-2 stem, four stages of bottleneck blocks
    a (1x1) -> b (grouped 3x3, strided in a stage's first block) -> [se] -> c (1x1, no activation) + shortcut (1x1
    strided conv where shape changes), ReLU."""
    r = regnet_cfg(cfg)
    L = [_L(f'{prefix}.stem.cbr.0', (r['w_stem'], in_channels, 3, 3), bn=f'{prefix}.stem.cbr.1')]
    w_in = r['w_stem']
    strides = regnet_stage_strides(cfg)
    for si, (w, d, g) in enumerate(zip(r['widths'], r['depths'], r['groups']), start=1):
        for b in range(1, d + 1):
            p = f'{prefix}.stage{si}.block{b}'
            stride = strides[si - 1] if b == 1 else 1
            L.append(_L(f'{p}.bottleneck.a.0', (w, w_in, 1, 1), bn=f'{p}.bottleneck.a.1'))
            L.append(_L(f'{p}.bottleneck.b.0', (w, w // g, 3, 3), bn=f'{p}.bottleneck.b.1'))
            if r['use_se']:
                L.append(_L(f'{p}.bottleneck.se.se.0', (w // 4, w, 1, 1), bias=True))
                L.append(_L(f'{p}.bottleneck.se.se.2', (w, w // 4, 1, 1), bias=True, gain=0.5, bias_mean=1.0))
            L.append(_L(f'{p}.bottleneck.c.0', (w, w, 1, 1), bn=f'{p}.bottleneck.c.1', bn_gamma=0.35))
            if w_in != w or stride > 1:
                L.append(_L(f'{p}.downsample.conv.0', (w, w_in, 1, 1), bn=f'{p}.downsample.conv.1', bn_gamma=0.7))
            w_in = w
    return L


def encoder_spec(cfg):
    return regnet_spec(cfg) if is_regnet(cfg) else resnet50_spec()


def pdl_decoder_spec(prefix, in_ch, dec_ch, low_level_channels, low_level_project, aspp_ch=None):
    """decoders/panoptic_deeplab.py:25-62 + decoders/aspp.py:51-94."""
    aspp_ch = aspp_ch or dec_ch
    L = []
    L.append(_L(f'{prefix}.aspp.convs.0.0', (aspp_ch, in_ch, 1, 1), bn=f'{prefix}.aspp.convs.0.1'))
    for i in (1, 2, 3):
        L.append(_L(f'{prefix}.aspp.convs.{i}.0', (aspp_ch, in_ch, 3, 3), bn=f'{prefix}.aspp.convs.{i}.1'))
    L.append(_L(f'{prefix}.aspp.convs.4.aspp_pooling.1', (aspp_ch, in_ch, 1, 1)))
    L.append(_L(f'{prefix}.aspp.project.0', (aspp_ch, 5 * aspp_ch, 1, 1), bn=f'{prefix}.aspp.project.1'))
    for i, (lc, lp) in enumerate(zip(low_level_channels, low_level_project)):
        L.append(_L(f'{prefix}.project.{i}.0', (lp, lc, 1, 1), bn=f'{prefix}.project.{i}.1'))
    for i, lp in enumerate(low_level_project):
        fin = (aspp_ch if i == 0 else dec_ch) + lp
        L.append(_L(f'{prefix}.fuse.{i}.0.sepconv.0', (fin, 1, 5, 5), kind='dw'))
        L.append(_L(f'{prefix}.fuse.{i}.0.sepconv.1', (dec_ch, fin, 1, 1), bn=f'{prefix}.fuse.{i}.1'))
    return L


def pdl_head_spec(prefix, nin, ncls, out_std=1.0, bias_mean=0.0):
    """heads.py:10-16: 5x5 separable + BN + ReLU, then 1x1 conv with bias."""
    return [
        _L(f'{prefix}.head.0.0.sepconv.0', (nin, 1, 5, 5), kind='dw'),
        _L(f'{prefix}.head.0.0.sepconv.1', (nin, nin, 1, 1), bn=f'{prefix}.head.0.1'),
        _L(f'{prefix}.head.1', (ncls, nin, 1, 1), bias=True, gain=out_std, zero_mean=True, bias_mean=bias_mean),
    ]


def bifpn_spec(cfg=None):
    """Ordered layer list of QuantizablePanopticBiFPNPR (models/panoptic_bifpn.py:22-67,
    decoders/bifpn.py:17-236, quantization/panoptic_bifpn.py:83-105).  ``alias`` lists the extra
    state-dict prefixes under which a SHARED module appears (one conv block instance serves every
    level of a direction, bifpn.py:41-42,97-98)."""
    cfg = dict(MITONET_MINI_CFG, **(cfg or {}))
    assert cfg['depthwise']
    F, ncls = cfg['fpn_dim'], cfg['num_classes']
    ew = encoder_widths(cfg)
    L = encoder_spec(cfg)
    L.append(_L('p2_resample.conv.0', (F, ew[0], 1, 1), bn='p2_resample.conv.1'))
    widths = ew[1:] + [F, F]                              # P3..P7 inputs of the first BiFPN layer
    for dec in (['semantic'] + (['instance'] if cfg['ins_decoder'] else [])):
        fp = f'{dec}_fpn'
        L.append(_L(f'{fp}.p6_resample.conv.0', (F, ew[3], 1, 1), bn=f'{fp}.p6_resample.conv.1'))
        for li in range(cfg['fpn_layers']):
            nins = widths if li == 0 else [F] * 5
            td_nins = nins[::-1][1:]                        # P6, P5, P4, P3
            bu_nins = nins[1:]                              # P4, P5, P6, P7
            for dname, dn in (('top_down_fpn', td_nins), ('bottom_up_fpn', bu_nins)):
                pre = f'{fp}.bifpns.{li}.{dname}'
                for i, nin in enumerate(dn):
                    if nin != F:
                        L.append(_L(f'{pre}.resamplings.{i}.conv.0', (F, nin, 1, 1), bn=f'{pre}.resamplings.{i}.conv.1'))
                al = [f'{pre}.after_combines.{i}' for i in range(1, 4)]
                d = _L(f'{pre}.after_combines.0.0.sepconv.0', (F, 1, 3, 3), kind='dw')
                d['alias'] = [a + '.0.sepconv.0' for a in al]
                L.append(d)
                d = _L(f'{pre}.after_combines.0.0.sepconv.1', (F, F, 1, 1), bn=f'{pre}.after_combines.0.1')
                d['alias'] = [a + '.0.sepconv.1' for a in al]
                d['bn_alias'] = [a + '.1' for a in al]
                L.append(d)
                L.append(_L(f'{pre}.weights', (5,), kind='fw'))
        dp = f'{dec}_decoder'
        for i in range(5):
            L.append(_L(f'{dp}.upsamplings.{i}.0', (F if i == 0 else 2 * F, F, 2, 2), bn=f'{dp}.upsamplings.{i}.1', kind='convT'))
        L.append(_L(f'{dp}.fusion.0.sepconv.0', (2 * F, 1, 5, 5), kind='dw'))
        L.append(_L(f'{dp}.fusion.0.sepconv.1', (F, 2 * F, 1, 1), bn=f'{dp}.fusion.1'))
    L += pdl_head_spec('semantic_head', F, ncls, out_std=0.8)
    L += pdl_head_spec('ins_center', F, 1, out_std=0.25, bias_mean=-0.5)
    L += pdl_head_spec('ins_xy', F, 2, out_std=3.0)
    fin = F + ncls
    for k in range(cfg['num_fc']):
        L.append(_L(f'semantic_pr.point_head.fc_layers.{k}.0', (F, fin, 1), bias=True, kind='fc'))
    L.append(_L('semantic_pr.point_head.predictor', (ncls, fin, 1), bias=True, kind='fc', gain=0.8, zero_mean=True))
    return L


def model_spec(cfg=None):
    arch = (cfg or {}).get('arch', 'PanopticDeepLabPR')
    return bifpn_spec(cfg) if 'BiFPN' in arch else pdl_spec(cfg)


def ins_projection_widths(cfg):
    """Projected low-level channels of the INSTANCE decoder, per stage.  The reference derives them as
    ``int(s * ins_ratio)`` (panoptic_deeplab.py: ``low_level_channels_project``, ``ins_ratio``); an export carries only the
    resulting widths, and ``int(p * (a / p)) != a`` for some ``(a, p)`` -- so ``infer_cfg`` stores the widths it READ under
    ``low_level_channels_project_ins`` and every consumer takes them from here instead of round-tripping through a float."""
    explicit = cfg.get('low_level_channels_project_ins')
    if explicit is not None:
        return [int(v) for v in explicit]
    return [int(s * cfg['ins_ratio']) for s in cfg['low_level_channels_project']]


def pdl_spec(cfg=None):
    """Ordered layer list of QuantizablePanopticDeepLabPR for ``cfg``."""
    cfg = dict(MITONET_PDL_CFG, **(cfg or {}))
    widths = encoder_widths(cfg)
    stages = cfg['low_level_stages']
    llc = [widths[s - 1] for s in stages]
    dec = cfg['decoder_channels']
    ncls = cfg['num_classes']
    L = encoder_spec(cfg)
    L += pdl_decoder_spec('semantic_decoder', widths[-1], dec, llc,
                          cfg['low_level_channels_project'], cfg['aspp_channels'])
    if cfg['ins_decoder']:
        L += pdl_decoder_spec('instance_decoder', widths[-1], dec, llc,
                              ins_projection_widths(cfg), cfg['aspp_channels'])
    L += pdl_head_spec('semantic_head', dec, ncls, out_std=0.8)
    L += pdl_head_spec('ins_center', dec, 1, out_std=0.25, bias_mean=-0.5)
    L += pdl_head_spec('ins_xy', dec, 2, out_std=3.0)
    fin = dec + ncls
    for k in range(cfg['num_fc']):
        L.append(_L(f'semantic_pr.point_head.fc_layers.{k}.0', (dec, fin, 1), bias=True, kind='fc'))
    L.append(_L('semantic_pr.point_head.predictor', (ncls, fin, 1), bias=True, kind='fc', gain=0.8, zero_mean=True))
    return L


def seeded_state_dict(cfg=None, seed=0):
    """Deterministic random-init parameters in the reference's unfused
    ``state_dict`` layout (numpy fp32 arrays; int64 for num_batches_tracked).

    Scales are chosen so every activation of the network stays O(1) and the
    three heads produce logits / heatmaps / offsets with a useful dynamic
    range (the reference's own init gives ~0 outputs, heads.py:21-26).
    """
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for L in model_spec(cfg):
        shp = L['shape']
        if L['kind'] == 'fw':   # fast-normalised fusion weights; one negative entry exercises the ReLU clip
            w = rng.uniform(0.4, 1.6, shp)
            w[rng.integers(0, shp[0])] = -0.2
            sd[L['name']] = w.astype(np.float32)
            continue
        fan_in = int(np.prod(shp[1:])) if L['kind'] != 'convT' else int(shp[0])
        std = L['gain'] * (np.sqrt(2.0 / fan_in) if L['bn'] or L['kind'] == 'fc' or L['kind'] == 'dw'
                           else np.sqrt(1.0 / fan_in))
        if L['kind'] == 'dw':
            std = np.sqrt(1.0 / fan_in) * 1.5
        w = rng.standard_normal(shp) * std
        if L['zero_mean']:  # keeps the output of a post-ReLU input centred
            w = w - w.mean(axis=1, keepdims=True)
        sd[L['name'] + '.weight'] = w.astype(np.float32)
        for a in L.get('alias', []):
            sd[a + '.weight'] = sd[L['name'] + '.weight']
        if L['bias']:
            sd[L['name'] + '.bias'] = (L['bias_mean'] + rng.standard_normal(shp[0]) * 0.1).astype(np.float32)
        if L['bn']:
            c = shp[1] if L['kind'] == 'convT' else shp[0]
            g = L['bn_gamma']
            sd[L['bn'] + '.weight'] = (g * rng.uniform(0.8, 1.2, c)).astype(np.float32)
            sd[L['bn'] + '.bias'] = (rng.standard_normal(c) * 0.05).astype(np.float32)
            sd[L['bn'] + '.running_mean'] = (rng.standard_normal(c) * 0.05).astype(np.float32)
            sd[L['bn'] + '.running_var'] = rng.uniform(0.8, 1.2, c).astype(np.float32)
            sd[L['bn'] + '.num_batches_tracked'] = np.array(0, dtype=np.int64)
            for a in L.get('bn_alias', []):
                for suf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    sd[a + '.' + suf] = sd[L['bn'] + '.' + suf]
    return sd


def _np(x):
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def fold_state_dict(sd, cfg=None):
    """-> OrderedDict name -> (w fp32 (Cout,Cin/g,kh,kw | Cout,Cin,1), b fp32 (Cout,)).

    Works on the unfused layout and on the ``fuse_model()`` layout of the
    exported TorchScript models (Conv+ReLU become ``<name>.0.weight/bias``,
    Conv+BN become ``<name>.weight/bias``, BatchNorm keys disappear).
    """
    out = OrderedDict()
    for L in model_spec(cfg):
        n = L['name']
        if L['kind'] == 'fw':
            v = _np(sd[n]).astype(np.float32)
            out[n] = (np.ascontiguousarray(v), np.zeros(v.shape[0], dtype=np.float32))
            continue
        if n + '.weight' in sd:
            w = _np(sd[n + '.weight']).astype(np.float32)
            b = _np(sd[n + '.bias']).astype(np.float32) if n + '.bias' in sd else None
        elif n + '.0.weight' in sd:  # ConvReLU2d / fused Sequential
            w = _np(sd[n + '.0.weight']).astype(np.float32)
            b = _np(sd[n + '.0.bias']).astype(np.float32) if n + '.0.bias' in sd else None
        else:
            raise KeyError(f'parameter for layer {n!r} not found in state dict')
        if tuple(w.shape) != L['shape']:
            raise ValueError(f'{n}: expected shape {L["shape"]}, got {tuple(w.shape)}')
        cdim = 1 if L['kind'] == 'convT' else 0      # ConvTranspose2d weight is (Cin, Cout, kh, kw)
        if b is None:
            b = np.zeros(w.shape[cdim], dtype=np.float32)
        bn = L['bn']
        if bn and bn + '.running_mean' in sd:
            gamma = _np(sd[bn + '.weight']).astype(np.float32)
            beta = _np(sd[bn + '.bias']).astype(np.float32)
            mean = _np(sd[bn + '.running_mean']).astype(np.float32)
            var = _np(sd[bn + '.running_var']).astype(np.float32)
            # same arithmetic as torch.nn.utils.fusion.fuse_conv_bn_weights (fp32)
            rstd = (1.0 / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
            scale = (gamma * rstd).astype(np.float32)
            bshape = [1] * w.ndim
            bshape[cdim] = -1
            w = (w * scale.reshape(bshape)).astype(np.float32)
            b = ((b - mean) * rstd * gamma + beta).astype(np.float32)
        out[n] = (np.ascontiguousarray(w), np.ascontiguousarray(b))
    return out


# ----------------------------------------------------------------------------
# architecture from the export itself
# ----------------------------------------------------------------------------
def _conv_attr(mod, attr):
    """attribute of a (possibly fused) conv of a loaded TorchScript module: ``Conv2d`` carries it itself, the fused
    ``ConvReLU2d`` / ``Sequential`` keeps the conv as child ``0``."""
    for _ in range(3):
        if mod is None:
            return None
        try:
            return getattr(mod, attr)
        except AttributeError:
            mod = dict(mod.named_children()).get('0')
    return None


def _child(mod, path):
    for part in path.split('.'):
        if mod is None:
            return None
        mod = dict(mod.named_children()).get(part)
    return mod


def infer_cfg(state_dict, module=None):
    """The model configuration (the keys of MITONET_PDL_CFG / MITONET_MINI_CFG) of an exported model, read from the export
    itself: the reference's YAML descriptors (empanada_napari/configs/*.yaml) carry no architecture, the widgets simply
    ``torch.jit.load`` the file (empanada_napari/utils.py:80-106) and call it.  Widths, class count, decoder layout and
    PointRend depth follow from the parameter shapes of ``state_dict`` (either key layout, see ``fold_state_dict``); what
    is not a parameter -- the ASPP dilations (decoders/aspp.py:51-94), the encoder's output stride
    (encoders/resnet.py:173-175) and PointRend's point budget (point_rend.py:146-176) -- is read from the attributes of
    the scripted submodules when ``module`` (the loaded TorchScript model) is given, else left at the training defaults
    (empanada_napari/training/*.yaml)."""
    keys = list(state_dict.keys())

    def shape(name):
        for suf in ('.weight', '.0.weight'):
            if name + suf in state_dict:
                return tuple(int(d) for d in state_dict[name + suf].shape)
        return None

    enc = {}
    if shape('encoder.stem.cbr.0') is not None:      # RegNet (encoders/regnet.py:38-160)
        widths, depths, groups = [], [], []
        for si in (1, 2, 3, 4):
            blocks = {k.split('.')[2] for k in keys if k.startswith(f'encoder.stage{si}.')}
            if not blocks or shape(f'encoder.stage{si}.block1.bottleneck.b.0') is None:
                raise NotImplementedError(f'RegNet export without stage {si}: only four-stage RegNets are built')
            w, gw = shape(f'encoder.stage{si}.block1.bottleneck.b.0')[:2]
            widths.append(w)
            depths.append(len(blocks))
            groups.append(w // gw)
        layout = dict(w_stem=shape('encoder.stem.cbr.0')[0], widths=widths, depths=depths, groups=groups,
                      use_se=any('.bottleneck.se.' in k for k in keys))
        name = next((n for n, prm in REGNET_PARAMS.items() if regnet_layout(*prm) == layout), 'regnet')
        enc = dict(encoder=name, regnet=layout)
        if module is not None:      # the strides as exported (regnet.py:141-142 can set the last one to 1)
            st = [_conv_attr(_child(module, f'encoder.stage{si}.block1.bottleneck.b'), 'stride') for si in (1, 2, 3, 4)]
            if all(v is not None for v in st) and [int(v[0]) for v in st] != [2, 2, 2, 2]:
                enc['regnet'] = dict(layout, strides=[int(v[0]) for v in st])
    elif shape('encoder.conv1') is None or shape('encoder.layer1.0.conv3') is None:
        stem = [k for k in keys if k.startswith('encoder.')][:3]
        raise NotImplementedError(f'only ResNet50 and RegNet encoders are built; encoder keys: {stem}')
    else:
        nblocks = tuple(len({k.split('.')[2] for k in keys if k.startswith(f'encoder.layer{li}.')}) for li in (1, 2, 3, 4))
        if nblocks != RESNET50_LAYERS or shape('encoder.layer1.0.conv3')[0] != 256:
            raise NotImplementedError(f'only the resnet50 encoder is built; this export has blocks {nblocks}')
    ncls = shape('semantic_head.head.1')[0]
    num_fc = len({k.split('.')[3] for k in keys if k.startswith('semantic_pr.point_head.fc_layers.')})
    npts = None
    if module is not None:
        pr = _child(module, 'semantic_pr')
        try:
            npts = int(pr.subdivision_num_points)
        except Exception:
            npts = None
    if any(k.startswith('semantic_fpn.') for k in keys):
        cfg = dict(MITONET_MINI_CFG, **enc)
        cfg.update(num_classes=ncls, fpn_dim=shape('p2_resample.conv.0')[0], num_fc=num_fc,
                   fpn_layers=len({k.split('.')[2] for k in keys if k.startswith('semantic_fpn.bifpns.')}),
                   ins_decoder=any(k.startswith('instance_fpn.') for k in keys),
                   depthwise=shape('semantic_fpn.bifpns.0.top_down_fpn.after_combines.0.0.sepconv.0') is not None)
        if not cfg['depthwise']:
            raise NotImplementedError('BiFPN exports without depthwise-separable node convolutions are not built')
    else:
        cfg = dict(MITONET_PDL_CFG, **enc)
        widths = encoder_widths(cfg)
        stages, proj = [], []
        while shape(f'semantic_decoder.project.{len(stages)}.0') is not None:
            lp, cin = shape(f'semantic_decoder.project.{len(stages)}.0')[:2]
            stages.append(widths.index(cin) + 1)
            proj.append(lp)
        ins = any(k.startswith('instance_decoder.') for k in keys)
        dec = shape('semantic_decoder.fuse.0.0.sepconv.1')[0]
        aspp = shape('semantic_decoder.aspp.convs.0.0')[0]
        cfg.update(num_classes=ncls, decoder_channels=dec, aspp_channels=None if aspp == dec else aspp, num_fc=num_fc,
                   low_level_stages=stages, low_level_channels_project=proj, ins_decoder=ins)
        if ins:
            # the widths as exported, stage by stage (ins_ratio is kept for display only: see ins_projection_widths)
            cfg['low_level_channels_project_ins'] = [shape(f'instance_decoder.project.{i}.0')[0] for i in range(len(stages))]
            cfg['ins_ratio'] = cfg['low_level_channels_project_ins'][0] / proj[0]
        if module is not None:
            rates = []
            for i in (1, 2, 3):
                d = _conv_attr(_child(module, f'semantic_decoder.aspp.convs.{i}'), 'dilation')
                if d is not None:
                    rates.append(int(d[0]))
            if len(rates) == 3:
                cfg['atrous_rates'] = rates
            d4 = _conv_attr(_child(module, 'encoder.layer4.0.conv2'), 'dilation')
            if d4 is not None:
                cfg['stage4_stride'] = 16 if int(d4[0]) == 2 else 32

    if npts:
        cfg['subdivision_num_points'] = npts
    return cfg
