// Panoptic-DeepLab + PointRend forward on gfx950: the layer schedule of
// QuantizablePanopticDeepLabPR.forward (eval) over the HIP kernels of this
// library.  Reference: empanada/models/quantization/panoptic_deeplab.py:103-115,
// 194-250; encoders/resnet.py:217-229; decoders/panoptic_deeplab.py:68-80;
// decoders/aspp.py:96-102; heads.py:12-19; point_rend.py:241-269.
//
// Data layout in HBM: activations NHWC fp16 in one arena (no aliasing: every
// intermediate stays addressable for the parity taps), conv weights fp16
// [Cout][KH*KW][Cin padded to 64], biases fp32, head outputs fp32 NCHW.
// Algebraic restructurings (all exact in real arithmetic):
//  * BatchNorm folded into the preceding conv by the host (weights.py);
//  * the ASPP image-pooling branch is a per-image constant after its 1x1 conv,
//    so it enters the 5-branch projection as a per-image bias instead of a
//    broadcast + concat (decoders/aspp.py:45-48,99-102);
//  * torch.cat is free: producers write into channel slices of one buffer;
//  * the unused first semantic_head call of _encode_decode (:107) is skipped.
#include <string.h>

#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace emp {

namespace {

struct HostParam {
  std::vector<int64_t> shape;
  std::vector<float> w, b;
  bool set = false;
};

struct Act {  // NHWC fp16 activation
  half_t* p = nullptr;
  int N = 0, H = 0, W = 0, C = 0, ld = 0;
  size_t off = 0;
};

struct DevConv {  // packed conv weights
  half_t* w = nullptr;
  half_t* w256 = nullptr;         // the 256 x 256 tile's image of w (conv256_pack_weights), made at the first launch that takes that tile
  float* b = nullptr;
  int cout = 0, cin = 0, cin_pad = 0, kh = 1, kw = 1;
  int cin2 = 0, cin2_pad = 0;     // K-concatenated second source (conv3 + projection shortcut)
  bool wsplit = false;            // the "second source" is the SAME input again, against the lo halves of an fp16 hi + lo weight pair
};

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace
}  // namespace emp

using namespace emp;

struct emp_pdl {
  emp_pdl_config cfg;
  int aspp_ch = 256, dec_ch = 256, ncls = 1;
  std::vector<std::string> param_names;
  std::map<std::string, HostParam> params;
  bool finalized = false;
  FILE* layer_log = [] { const char* e = getenv("EMP_LAYER_LOG"); return e ? fopen(e, "w") : (FILE*)nullptr; }();
  // fused separable convs (sepconv.hip); EMP_FUSE_SEPCONV=0 keeps the dwconv + 1x1 conv + head1x1 launches (A/B runs)
  bool fuse_stem = [] { const char* e = getenv("EMP_FUSE_STEM"); return !(e && e[0] == '0'); }();   // stem.hip
  bool fuse_ds = [] { const char* e = getenv("EMP_FUSE_DS"); return !(e && e[0] == '0'); }();       // conv3 + downsample in one GEMM
  bool fuse_b2b = [] { const char* e = getenv("EMP_FUSE_B2B"); return !(e && e[0] == '0'); }();     // conv3 + the next block's conv1
  bool pack256 = [] { const char* e = getenv("EMP_CONV256_PACK"); return !(e && e[0] == '0'); }();   // packed weight images for the 256 x 256 tile (A/B runs: 0)
  bool fuse_proj = [] { const char* e = getenv("EMP_FUSE_PROJ"); return !(e && e[0] == '0'); }();   // low-level projections + the next stage's conv1
  bool fuse_aspp = [] { const char* e = getenv("EMP_FUSE_ASPP"); return !(e && e[0] == '0'); }();   // the two decoders' ASPP branches as one conv each
  bool fuse_sepconv = [] { const char* e = getenv("EMP_FUSE_SEPCONV"); return !(e && e[0] == '0'); }();
  bool fuse_pr = [] { const char* e = getenv("EMP_FUSE_PR"); return !(e && e[0] == '0'); }();             // pointrend.hip
  // Separable blocks with an exact depthwise half (sepconv_precise.hip: fp32 taps, depthwise result as fp16 hi + lo, 2
  // MFMAs per product).  EMP_PRECISE_SEPCONV:
  //   1 (default) = the blocks the CENTRE heat-map and the offsets depend on: the last-stage fusion conv(s) of the decoder
  //                 that feeds ins_center, the ins_center head and -- BiFPN networks, round 4 -- the ins_xy head and every
  //                 3x3 node of the FPN that feeds that decoder.  The precise nodes ALWAYS run fused (no tile-count threshold): one kernel,
  //                 one rounding sequence at every batch size, so a tile's result does not depend on the batch it
  //                 arrives in;
  //   2 = every fused block and node; 0 = none (round-2 numerics);
  //   A/B switches: 3 = the ins_center head only, 4 = the decoder's fusion convs only, 5 = 1 + the nodes of BOTH FPNs,
  //   6 = 1 + the ins_xy head, 7 = round 3's default (head + fusion convs, no nodes).
  int precise_sepconv = [] { const char* e = getenv("EMP_PRECISE_SEPCONV"); return e ? atoi(e) : 1; }();
  // BiFPN networks (round 4): the weights of the layers the centre heat-map is most sensitive to as fp16 hi + lo PAIRS --
  // the pointwise convs of the precise 128-cout blocks (a third MFMA per product, sepconv_precise.hip WS) and the
  // transposed convs of the decoder that feeds the centre head (the lo halves ride as a K-concatenated "second source"
  // on the same input: ConvParams::in2).  tools/error_budget.py --arch bifpn: those roundings are 80 % of the weight-side
  // variance; 512^2 tile, ctr rms / scale 1.13e-3 -> 0.96e-3.  EMP_PRECISE_WSPLIT=0 switches it off (A/B).
  bool precise_wsplit = [] { const char* e = getenv("EMP_PRECISE_WSPLIT"); return !(e && e[0] == '0'); }();
  // ... and the FUSED MAPS of those nodes (the fast-normalised sum a node's separable conv reads) as fp16 hi + lo pairs too:
  // fuse_combine writes channels [hi | lo] of a 2F-wide buffer and the node runs with 2F input channels, duplicated
  // depthwise taps and pointwise weights (depthwise and pointwise are linear: dw(hi) W + dw(lo) W = dw(hi + lo) W).  The
  // 24 fused-map roundings of an FPN were worth more than their share of the variance: 512^2 tile, ctr rms / scale
  // 0.96e-3 -> 0.78e-3 (format-emulating oracle).  EMP_PRECISE_FSPLIT=0 switches it off (A/B).
  bool precise_fsplit = [] { const char* e = getenv("EMP_PRECISE_FSPLIT"); return !(e && e[0] == '0'); }();
  // BiFPN nodes run fused once the map has this many 8 x 16 tiles (a tile per CU); EMP_SEPCONV_MIN_TILES for A/B runs
  int sepconv_min_tiles = [] { const char* e = getenv("EMP_SEPCONV_MIN_TILES"); return e ? atoi(e) : 256; }();

  // fp32 reference mode (emp_pdl_set_precision / EMP_PRECISION=fp32; run32 below): fp32 weights, fp32 activation pool
  // precision 2 = the fp16x3 mode (round 5): the fp32 mode's graph, maps and weights, its convolutions on the fp16 matrix
  // pipe with split operands (conv16x3.hip: three MFMAs per product into an fp32 accumulator)
  // Round 6: the DEFAULT is 2 -- the mode that meets the north star's tolerance (1e-3 of the reference's fp32 forward in the
  // max norm) on every network; the fp16 engine (0) is the explicit throughput opt-in (emp_pdl_set_precision(net, 0) /
  // EMP_PRECISION=fp16: ~5e-3 in the max norm)
  int precision = [] {
    const char* e = getenv("EMP_PRECISION");
    if (e && (!strcmp(e, "fp16") || !strcmp(e, "16"))) return 0;
    if (e && (!strcmp(e, "fp32") || !strcmp(e, "32"))) return 1;
    return 2;
  }();
  bool fp32_graph() const { return precision != 0; }
  // RegNet on the fp16 engine: the grouped 3x3 as ONE launch (blockIdx.y = group, conv_igemm_grouped.hip); EMP_REGNET_GROUPED=0:
  // one launch per group with its couts padded to 64 / 128 for the register-weight kernels (round 4; A/B)
  // fp16x3 mode: the heads' 1x1 fused into the pointwise conv (EMP_X3_FUSE_HEAD=0: the separate head1x1_32 launch; A/B)
  bool x3_fuse_ds = [] { const char* e = getenv("EMP_X3_FUSE_DS"); return !(e && e[0] == '0'); }();      // conv3 + projection shortcut as one K-concatenated conv (A/B)
  bool x3_fuse_head = [] { const char* e = getenv("EMP_X3_FUSE_HEAD"); return !(e && e[0] == '0'); }();
  // fp16x3 mode, round 6: the stride-16 region of a ResNet50 network (layer3, layer4, ASPP) as hl32 maps on conv16x3p_kernel's
  // 256 x 256 tile once a layer3 map has this many pixel tiles (one tile of 256 couts per pixel tile is a whole launch of the
  // 256-channel layers of layer3; the ASPP branches of the two decoders run merged, 512 couts per launch.  Whole step, planes vs
  // round 5's kernels: batch 4 (64 tiles) 413.8 vs 439.2 tiles/s, batch 8 (128) 520.1 vs 469.7, batch 16 545.3 vs 472.0;
  // profiles/r06_x3p.txt).
  // EMP_X3_PLANES=0: never (A/B); EMP_X3_PLANES_MIN_TILES=n
  bool x3_planes = [] { const char* e = getenv("EMP_X3_PLANES"); return !(e && e[0] == '0'); }();
  int x3_planes_min_tiles = [] { const char* e = getenv("EMP_X3_PLANES_MIN_TILES"); return e ? atoi(e) : 128; }();
  bool x3_planes_ready = false;      // set by finalize32: every layer of the region has its packed image
  // fp16x3 mode, round 6 (late): split-K for the long-K launches that fill less than half the chip (ONE 1024^2 tile: each 3x3 ASPP branch
  // is 64 workgroups over K = 18 432) -- Conv32::kpart; EMP_X3_KSPLIT=0: never (A/B)
  bool x3_ksplit = [] { const char* e = getenv("EMP_X3_KSPLIT"); return !(e && e[0] == '0'); }();
  float* x3_kpart = nullptr;      // X3_KPART_BYTES of scratch, made by finalize32
  bool x3_small_aspp = [] { const char* e = getenv("EMP_X3_SMALL_ASPP"); return !(e && e[0] == '0'); }();      // below the plane region's threshold the ASPP branches still run merged on the plane kernel, K-split (A/B)
  bool x3_fuse_stem = [] { const char* e = getenv("EMP_X3_FUSE_STEM"); return !(e && e[0] == '0'); }();      // stem + max-pool as one MFMA launch (A/B)
  bool x3_merge_proj = [] { const char* e = getenv("EMP_X3_MERGE_PROJ"); return !(e && e[0] == '0'); }();      // both decoders' low-level projections as one launch (A/B)
  bool x3_merge_aspp = [] { const char* e = getenv("EMP_X3_MERGE_ASPP"); return !(e && e[0] == '0'); }();      // both decoders' ASPP branches as one launch (A/B)
  // fp16x3 mode, round 6: a separable block (depthwise KxK -> pointwise -> act [-> head 1x1]) as ONE launch (sepconv_x3.hip) once
  // the map has this many 8 x 16 tiles (a persistent workgroup per CU); EMP_X3_FUSE_SEP=0: the depthwise launch + conv16x3 (A/B)
  bool x3_fuse_sep = [] { const char* e = getenv("EMP_X3_FUSE_SEP"); return !(e && e[0] == '0'); }();
  int x3_sep_min_tiles = [] { const char* e = getenv("EMP_X3_SEP_MIN_TILES"); return e ? atoi(e) : 1; }();      // (256 until finding 75: fewer launches win at every size measured)
  struct SepX3 { float* dw = nullptr; half_t* pw = nullptr; int C = 0, Cout = 0, ks = 0; };
  std::map<std::string, SepX3> sepx3;      // by the block's name ("... .sepconv" without the .0 / .1)
  bool regnet_grouped = [] { const char* e = getenv("EMP_REGNET_GROUPED"); return !(e && e[0] == '0'); }();
  int64_t regnet_group_tiles = [] { const char* e = getenv("EMP_REGNET_GROUP_TILES"); return e ? atoll(e) : 2048ll; }();
  struct W32 { float* w = nullptr; float* b = nullptr; int cout = 0, cin = 0, cin16 = 0, kh = 1, kw = 1; uint32_t* wp = nullptr; int cin2 = 0, cin2_16 = 0; half_t* wimg = nullptr; half_t* wimgp = nullptr; int x3p_kg = 0; };
  std::map<std::string, W32> w32;
  std::map<std::string, std::pair<float*, size_t>> pool32;      // name -> (device buffer, floats)
  std::map<std::string, std::array<int, 4>> geom32;             // zero-tailed RegNet maps: the geometry a buffer was last cleared for

  // device parameters
  std::map<std::string, DevConv> convs;
  std::map<std::string, float*> f32w;  // fp32 device blobs (stem, gemv, heads)
  std::map<std::string, half_t*> f16w;  // fp16 device blobs (depthwise taps)
  std::map<std::string, std::vector<float>> fusew;  // BiFPN fast-fusion weights after relu / (sum + eps)
  std::vector<void*> owned;

  // arena
  char* arena = nullptr;
  size_t arena_cap = 0, arena_used = 0;
  int pN = 0, pH = 0, pW = 0, pRS = 0;  // planned shape (pN: the batch of the current forward)
  int capN = 0;                         // batch the arena layout was planned for (pN <= capN)
  std::map<std::string, Act> acts;
  std::vector<std::string> act_order;
  std::map<std::string, std::pair<size_t, size_t>> raw;  // name -> (offset, bytes)
  double flops = 0.0;
  // live timing of the dominant kernel class (256x256 conv tile): HIP event pairs on the launch stream, summed by
  // emp_pdl_profile_read (bench.py's roofline block)
  size_t image_bytes = 0;      // packed 256 x 256 weight images made at finalize (fp16 engine)
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  double prof_flops = 0.0;
  // The two decoders (+ their heads) are independent after the encoder: for small problems, whose launches leave CUs
  // idle (one 1024^2 tile: <= 256 workgroups per launch), the instance side runs on a second stream -- batch-1 call
  // 2.10 -> 1.85 ms, 4 tiles 4.70 -> 3.83 ms.  EMP_PAR_DECODERS = pixel count N*H*W up to which this is done (0 = never).
  // At the bench size it would still buy 1.7 % (24.74 -> 24.34 ms per 32 tiles) but time-slices CUs between launches of
  // the two streams, so that per-kernel durations (and the roofline of the dominant kernel: 0.45 -> 0.30) stop
  // describing the kernels: large problems stay on one stream.
  int64_t par_limit = [] { const char* e = getenv("EMP_PAR_DECODERS"); return e ? atoll(e) : (int64_t)4 << 20; }();
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;

  ~emp_pdl() {
    for (void* p : owned) (void)hipFree(p);
    for (auto& kv : pool32) (void)hipFree(kv.second.first);
    if (arena) (void)hipFree(arena);
    if (layer_log) fclose(layer_log);
    for (auto& e : prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (aux) (void)hipStreamDestroy(aux);
  }
};

namespace {

const int kLayers[4] = {3, 4, 6, 3};
const int kPlanes[4] = {64, 128, 256, 512};

// does the separable block `pre` (its parameters are pre.sepconv.0 / pre.sepconv.1) run with the exact depthwise half (sepconv_precise.hip)?
bool precise_layer(const emp_pdl* n, const std::string& pre) {
  const int mode = n->precise_sepconv;
  if (mode <= 0) return false;
  if (mode == 2) return true;
  const char* dec = n->cfg.ins_decoder ? "instance_decoder." : "semantic_decoder.";
  const bool head = pre.compare(0, 11, "ins_center.") == 0, fuse = pre.compare(0, strlen(dec), dec) == 0;
  if (mode == 3) return head;      // A/B switches: only the head / only the decoder's fusion convs
  if (mode == 4) return fuse;
  // the offsets head: mode 6, and by default on BiFPN networks (there the offsets sat at 1.1e-3 of their scale with it in
  // fp16 and the block is a 128-channel one: +0.5 % of a forward; the Panoptic-DeepLab offsets are at 7e-4 without it)
  if ((mode == 6 || (mode == 1 && n->cfg.arch == 1)) && pre.compare(0, 7, "ins_xy.") == 0) return true;
  return head || fuse;
}

// does the 3x3 separable node block of the BiFPN whose parameter names start with `name` ("semantic_fpn..." /
// "instance_fpn...") run with the exact depthwise half?  Default: the FPN that feeds the centre / offset heads.
bool precise_node(const emp_pdl* n, const std::string& name) {
  const int mode = n->precise_sepconv;
  if (mode == 2 || mode == 5) return true;
  if (mode != 1 && mode != 6) return false;
  const char* fp = n->cfg.ins_decoder ? "instance_fpn." : "semantic_fpn.";
  return name.compare(0, strlen(fp), fp) == 0;
}

// hi + lo weight pairs (emp_pdl::precise_wsplit): BiFPN networks, the modes with precise nodes
bool wsplit_on(const emp_pdl* n) {
  const int mode = n->precise_sepconv;
  return n->cfg.arch == 1 && n->precise_wsplit && (mode == 1 || mode == 2 || mode == 5 || mode == 6);
}

// the fused maps of this FPN's nodes travel as hi + lo pairs (emp_pdl::precise_fsplit): name starts with "<dec>_fpn."
bool fsplit_on(const emp_pdl* n, const std::string& name) {
  const int F = n->cfg.fpn_dim;
  return wsplit_on(n) && n->precise_fsplit && n->fuse_sepconv && precise_node(n, name) && sepconvp_supported(2 * F, F, 0);
}

void expect(emp_pdl* n, const std::string& name) {
  n->param_names.push_back(name);
  n->params[name] = HostParam();
}

// RegNet block (regnet.py:51-97): does block b (1-based) of stage si (1-based) carry a shortcut convolution?
bool regnet_has_shortcut(const emp_pdl_config& c, int si, int b) {
  if (b > 1) return false;
  const int w_in = si == 1 ? c.rn_stem : c.rn_widths[si - 2];
  return w_in != c.rn_widths[si - 1] || c.rn_strides[si - 1] > 1;
}

void build_param_list(emp_pdl* n) {
  const emp_pdl_config& c = n->cfg;
  if (c.encoder == 1) {
    expect(n, "encoder.stem.cbr.0");
    for (int si = 1; si <= 4; ++si)
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        expect(n, p + ".bottleneck.a.0");
        expect(n, p + ".bottleneck.b.0");
        if (c.rn_se) {
          expect(n, p + ".bottleneck.se.se.0");
          expect(n, p + ".bottleneck.se.se.2");
        }
        expect(n, p + ".bottleneck.c.0");
        if (regnet_has_shortcut(c, si, b)) expect(n, p + ".downsample.conv.0");
      }
  } else {
    expect(n, "encoder.conv1");
    for (int li = 1; li <= 4; ++li)
      for (int b = 0; b < kLayers[li - 1]; ++b) {
        std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
        expect(n, p + ".conv1");
        expect(n, p + ".conv2");
        expect(n, p + ".conv3");
        if (b == 0) expect(n, p + ".downsample.0");
      }
  }
  if (c.arch == 1) {
    expect(n, "p2_resample.conv.0");
    const int F = c.fpn_dim;
    const bool rn = c.encoder == 1;
    const int w0[5] = {rn ? c.rn_widths[1] : 512, rn ? c.rn_widths[2] : 1024, rn ? c.rn_widths[3] : 2048, F, F};
    const char* dn[2] = {"semantic", "instance"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      std::string fp = std::string(dn[d]) + "_fpn";
      expect(n, fp + ".p6_resample.conv.0");
      for (int li = 0; li < c.fpn_layers; ++li) {
        int nins[5];
        for (int i = 0; i < 5; ++i) nins[i] = li == 0 ? w0[i] : F;
        const int td[4] = {nins[3], nins[2], nins[1], nins[0]};
        const int bu[4] = {nins[1], nins[2], nins[3], nins[4]};
        for (int dir = 0; dir < 2; ++dir) {
          std::string pre = fp + ".bifpns." + std::to_string(li) + (dir == 0 ? ".top_down_fpn" : ".bottom_up_fpn");
          for (int i = 0; i < 4; ++i)
            if ((dir == 0 ? td[i] : bu[i]) != F) expect(n, pre + ".resamplings." + std::to_string(i) + ".conv.0");
          expect(n, pre + ".after_combines.0.0.sepconv.0");
          expect(n, pre + ".after_combines.0.0.sepconv.1");
          expect(n, pre + ".weights");
        }
      }
      std::string dp = std::string(dn[d]) + "_decoder";
      for (int i = 0; i < 5; ++i) expect(n, dp + ".upsamplings." + std::to_string(i) + ".0");
      expect(n, dp + ".fusion.0.sepconv.0");
      expect(n, dp + ".fusion.0.sepconv.1");
    }
  }
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  for (int d = 0; d < ((c.arch == 0 && c.ins_decoder) ? 2 : (c.arch == 0 ? 1 : 0)); ++d) {
    std::string p = decs[d];
    expect(n, p + ".aspp.convs.0.0");
    for (int i = 1; i <= 3; ++i) expect(n, p + ".aspp.convs." + std::to_string(i) + ".0");
    expect(n, p + ".aspp.convs.4.aspp_pooling.1");
    expect(n, p + ".aspp.project.0");
    for (int i = 0; i < c.n_stages; ++i) expect(n, p + ".project." + std::to_string(i) + ".0");
    for (int i = 0; i < c.n_stages; ++i) {
      expect(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.0");
      expect(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1");
    }
  }
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  for (int h = 0; h < 3; ++h) {
    std::string p = heads[h];
    expect(n, p + ".head.0.0.sepconv.0");
    expect(n, p + ".head.0.0.sepconv.1");
    expect(n, p + ".head.1");
  }
  for (int k = 0; k < c.num_fc; ++k) expect(n, "semantic_pr.point_head.fc_layers." + std::to_string(k) + ".0");
  expect(n, "semantic_pr.point_head.predictor");
}

int dev_upload(emp_pdl* n, const void* h, size_t bytes, void** out) {
  void* d = nullptr;
  EMP_CHECK_HIP(hipMalloc(&d, bytes ? bytes : 16));
  n->owned.push_back(d);
  if (bytes) EMP_CHECK_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
  *out = d;
  return EMP_OK;
}

// OIHW fp32 -> [O][KH*KW][Ipad] fp16 (+ fp32 bias)
int pack_conv(emp_pdl* n, const std::string& name, int cin_pad_to = 0) {
  const HostParam& hp = n->params.at(name);
  EMP_REQUIRE(hp.shape.size() == 4 || hp.shape.size() == 3, "%s: conv weight must be 3-d or 4-d", name.c_str());
  DevConv dc;
  dc.cout = (int)hp.shape[0];
  dc.cin = (int)hp.shape[1];
  dc.kh = hp.shape.size() == 4 ? (int)hp.shape[2] : 1;
  dc.kw = hp.shape.size() == 4 ? (int)hp.shape[3] : 1;
  dc.cin_pad = cin_pad_to ? cin_pad_to : round_up(dc.cin, 64);
  EMP_REQUIRE(dc.cin_pad >= dc.cin, "%s: bad padding", name.c_str());
  const int kt = dc.kh * dc.kw;
  std::vector<half_t> pk((size_t)dc.cout * kt * dc.cin_pad, (half_t)0.f);
  for (int o = 0; o < dc.cout; ++o)
    for (int i = 0; i < dc.cin; ++i)
      for (int t = 0; t < kt; ++t)
        pk[((size_t)o * kt + t) * dc.cin_pad + i] = (half_t)hp.w[((size_t)o * dc.cin + i) * kt + t];
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(half_t), &d);
  if (rc) return rc;
  dc.w = (half_t*)d;
  rc = dev_upload(n, hp.b.data(), hp.b.size() * sizeof(float), &d);
  if (rc) return rc;
  dc.b = (float*)d;
  n->convs[name] = dc;
  return EMP_OK;
}

// conv3 and the projection shortcut of a bottleneck as one weight matrix [Cout][cin3_pad | cin_ds_pad] with the summed
// (folded BN) bias: relu(conv3(c2) + downsample(x)) becomes a single GEMM over the concatenated K (ConvParams::in2)
int pack_conv3_ds(emp_pdl* n, const std::string& block) {
  const HostParam& h3 = n->params.at(block + ".conv3");
  const HostParam& hd = n->params.at(block + ".downsample.0");
  EMP_REQUIRE(h3.shape.size() == 4 && hd.shape.size() == 4 && h3.shape[0] == hd.shape[0] && h3.shape[2] == 1 &&
                  hd.shape[2] == 1 && h3.b.size() == hd.b.size(), "%s: conv3 / downsample shapes do not match", block.c_str());
  DevConv dc;
  dc.cout = (int)h3.shape[0];
  dc.cin = (int)h3.shape[1];
  dc.cin_pad = round_up(dc.cin, 64);
  dc.cin2 = (int)hd.shape[1];
  dc.cin2_pad = round_up(dc.cin2, 64);
  const size_t K = (size_t)dc.cin_pad + dc.cin2_pad;
  std::vector<half_t> pk((size_t)dc.cout * K, (half_t)0.f);
  std::vector<float> b((size_t)dc.cout);
  for (int o = 0; o < dc.cout; ++o) {
    for (int i = 0; i < dc.cin; ++i) pk[o * K + i] = (half_t)h3.w[(size_t)o * dc.cin + i];
    for (int i = 0; i < dc.cin2; ++i) pk[o * K + dc.cin_pad + i] = (half_t)hd.w[(size_t)o * dc.cin2 + i];
    b[o] = h3.b[o] + hd.b[o];
  }
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(half_t), &d);
  if (rc) return rc;
  dc.w = (half_t*)d;
  rc = dev_upload(n, b.data(), b.size() * sizeof(float), &d);
  if (rc) return rc;
  dc.b = (float*)d;
  n->convs[block + ".conv3+ds"] = dc;
  return EMP_OK;
}

int upload_f32(emp_pdl* n, const std::string& key, const std::vector<float>& v);

// RegNet stem (W,1,3,3) -> [9][W] fp32 (both precisions compute it in fp32)
int upload_regnet_stem(emp_pdl* n) {
  const emp_pdl_config& c = n->cfg;
  const HostParam& hp = n->params["encoder.stem.cbr.0"];
  EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[0] == c.rn_stem && hp.shape[1] == 1 && hp.shape[2] == 3 && hp.shape[3] == 3,
              "encoder.stem.cbr.0 must be (%d,1,3,3)", c.rn_stem);
  std::vector<float> w((size_t)9 * c.rn_stem);
  for (int o = 0; o < c.rn_stem; ++o)
    for (int t = 0; t < 9; ++t) w[(size_t)t * c.rn_stem + o] = hp.w[(size_t)o * 9 + t];
  int rc = upload_f32(n, "rn.stem.w", w);
  if (rc) return rc;
  return upload_f32(n, "rn.stem.b", hp.b);
}

// fragment-ordered copy of a pointwise weight for the fused separable conv (sepconv.hip), when its shape qualifies
int pack_sepconv_pw(emp_pdl* n, const std::string& name) {
  const DevConv& dc = n->convs.at(name);
  if (dc.kh != 1 || dc.kw != 1 || !sepconv5_supported(dc.cin_pad, dc.cout, 0)) return EMP_OK;
  void* d = nullptr;
  EMP_CHECK_HIP(hipMalloc(&d, (size_t)dc.cin_pad * dc.cout * sizeof(half_t)));
  n->owned.push_back(d);
  int rc = launch_sepconv5_pack_pw(dc.w, dc.cin_pad, dc.cin_pad, dc.cout, (half_t*)d, nullptr);
  if (rc) return rc;
  EMP_CHECK_HIP(hipStreamSynchronize(nullptr));
  n->f16w[name + ".packed"] = (half_t*)d;
  return EMP_OK;
}

// fragment-ordered fp16 copy of a pointwise weight for the fused separable conv with the exact depthwise half
// (sepconv_precise.hip), when its shape qualifies
// dup: the weight rows twice, [W | W] -- for an input that arrives as channels [hi | lo] (fsplit_on)
int pack_sepconvp_pw(emp_pdl* n, const std::string& name, bool dup = false) {
  const DevConv& dc = n->convs.at(name);
  const int cin_eff = (dup ? 2 : 1) * dc.cin_pad;
  if (dc.kh != 1 || dc.kw != 1 || !sepconvp_supported(cin_eff, dc.cout, 0)) return EMP_OK;
  const HostParam& hp = n->params.at(name);
  std::vector<float> w32((size_t)dc.cout * cin_eff, 0.f);
  for (int o = 0; o < dc.cout; ++o)
    for (int i = 0; i < dc.cin; ++i) {
      w32[(size_t)o * cin_eff + i] = hp.w[(size_t)o * dc.cin + i];
      if (dup) w32[(size_t)o * cin_eff + dc.cin_pad + i] = hp.w[(size_t)o * dc.cin + i];
    }
  void* tmp = nullptr;
  EMP_CHECK_HIP(hipMalloc(&tmp, w32.size() * sizeof(float)));
  hipError_t e = hipMemcpy(tmp, w32.data(), w32.size() * sizeof(float), hipMemcpyHostToDevice);
  void* d = nullptr;
  const bool ws = wsplit_on(n) && sepconvp_wsplit_supported(dc.cout);
  if (e == hipSuccess) e = hipMalloc(&d, (size_t)(ws ? 2 : 1) * cin_eff * dc.cout * sizeof(half_t));
  if (e != hipSuccess) {
    (void)hipFree(tmp);
    set_error("%s: packing the pointwise weights: %s", name.c_str(), hipGetErrorString(e));
    return EMP_ERR_HIP;
  }
  n->owned.push_back(d);
  int rc = launch_sepconvp_pack_pw((const float*)tmp, cin_eff, cin_eff, dc.cout, (half_t*)d, nullptr, ws ? 1 : 0);
  n->convs[name].wsplit = ws;
  e = hipStreamSynchronize(nullptr);
  (void)hipFree(tmp);
  if (rc) return rc;
  EMP_CHECK_HIP(e);
  n->f16w[name + (dup ? ".packedpd" : ".packedp")] = (half_t*)d;
  return EMP_OK;
}

int upload_f32(emp_pdl* n, const std::string& key, const std::vector<float>& v) {
  void* d;
  int rc = dev_upload(n, v.data(), v.size() * sizeof(float), &d);
  if (rc) return rc;
  n->f32w[key] = (float*)d;
  return EMP_OK;
}

// depthwise (C,1,5,5) -> [25][Cpad] fp16
int pack_dw(emp_pdl* n, const std::string& name, int cpad, bool dup = false) {
  const HostParam& hp = n->params.at(name);
  const int C = (int)hp.shape[0];
  EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[1] == 1 && hp.shape[2] == hp.shape[3] &&
                  (hp.shape[2] == 5 || hp.shape[2] == 3) && cpad >= C,
              "%s: expected a (C,1,k,k) depthwise weight with k in {3,5}", name.c_str());
  EMP_REQUIRE(cpad % 64 == 0, "%s: padded channel count must be a multiple of 64", name.c_str());
  const int KK = (int)(hp.shape[2] * hp.shape[3]);
  std::vector<half_t> pk((size_t)KK * cpad, (half_t)0.f);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < KK; ++t) pk[(size_t)t * cpad + c] = (half_t)hp.w[(size_t)c * KK + t];
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(half_t), &d);
  if (rc) return rc;
  n->f16w[name] = (half_t*)d;
  // the fp32-accurate fused separable conv (sepconv_precise.hip) keeps the taps in fp32 (they only meet the fp32 vector pipe), chunk-major
  // [cpad/64][KK][64]: the order of launch_sepconvp_pack_dw
  std::vector<float> pk32((size_t)KK * cpad, 0.f);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < KK; ++t) pk32[((size_t)(c >> 6) * KK + t) * 64 + (c & 63)] = hp.w[(size_t)c * KK + t];
  if (dup) {      // the same taps for channels [0, cpad) and [cpad, 2 cpad): the input arrives as [hi | lo] (fsplit_on)
    std::vector<float> d2((size_t)2 * KK * cpad, 0.f);
    std::copy(pk32.begin(), pk32.end(), d2.begin());
    std::copy(pk32.begin(), pk32.end(), d2.begin() + (size_t)KK * cpad);
    const int rc2 = upload_f32(n, name + ".f32d", d2);
    if (rc2) return rc2;
  }
  return upload_f32(n, name + ".f32", pk32);
}

// ConvTranspose2d(k=2,s=2) weight (Cin,Cout,2,2) -> 1x1 conv with 4*Cout outputs [(dy*2+dx)*Cout + co][Cin]
// split: rows [hi | lo] -- the fp16 residuals of the weights behind them, multiplied with the same input as a
// K-concatenated second source (ConvParams::in2 = in): w*x = hi*x + lo*x in one fp32 accumulation
int pack_convT(emp_pdl* n, const std::string& name, bool split = false) {
  const HostParam& hp = n->params.at(name);
  EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[2] == 2 && hp.shape[3] == 2, "%s: expected (Cin,Cout,2,2)", name.c_str());
  DevConv dc;
  dc.cin = (int)hp.shape[0];
  const int co = (int)hp.shape[1];
  dc.cout = 4 * co;
  dc.cin_pad = round_up(dc.cin, 64);
  dc.kh = dc.kw = 1;
  if (split) { dc.cin2 = dc.cin; dc.cin2_pad = dc.cin_pad; dc.wsplit = true; }
  const size_t K = (size_t)dc.cin_pad + dc.cin2_pad;
  std::vector<half_t> pk((size_t)dc.cout * K, (half_t)0.f);
  std::vector<float> b((size_t)dc.cout, 0.f);
  for (int q = 0; q < 4; ++q)
    for (int o = 0; o < co; ++o) {
      b[(size_t)q * co + o] = hp.b[o];
      for (int i = 0; i < dc.cin; ++i) {
        const float w = hp.w[(((size_t)i * co + o) * 2 + (q >> 1)) * 2 + (q & 1)];
        const half_t hi = (half_t)w;
        pk[((size_t)q * co + o) * K + i] = hi;
        if (split) pk[((size_t)q * co + o) * K + dc.cin_pad + i] = (half_t)(w - (float)hi);
      }
    }
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(half_t), &d);
  if (rc) return rc;
  dc.w = (half_t*)d;
  rc = dev_upload(n, b.data(), b.size() * sizeof(float), &d);
  if (rc) return rc;
  dc.b = (float*)d;
  n->convs[name] = dc;
  return EMP_OK;
}

// ---- arena planning ------------------------------------------------------
struct Planner {
  size_t off = 0;
  size_t take(size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  }
};

void add_act(emp_pdl* n, Planner& pl, const std::string& name, int N, int H, int W, int C, int ld = 0) {
  Act a;
  a.N = N; a.H = H; a.W = W; a.C = C; a.ld = ld ? ld : C;
  a.off = pl.take((size_t)N * H * W * a.ld * sizeof(half_t));
  n->acts[name] = a;
  n->act_order.push_back(name);
}
void add_raw(emp_pdl* n, Planner& pl, const std::string& name, size_t bytes) {
  n->raw[name] = {pl.take(bytes), bytes};
}



int plan(emp_pdl* n, int N, int H, int W, int RS, hipStream_t stream) {
  const emp_pdl_config& c = n->cfg;
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0,
              "forward: H=%d W=%d must be positive multiples of 16 (factor_pad first)", H, W);
  EMP_REQUIRE(RS >= 1 && RS <= 6, "render_steps=%d out of range", RS);
  n->acts.clear();
  n->act_order.clear();
  n->raw.clear();
  Planner pl;
  add_raw(n, pl, "zero", 2048);
  int ph[5], pw[5];
  if (c.encoder == 1) {
    // RegNet: widths are multiples of 8, the kernels read channels in slabs of 64 -- every map gets rows of
    // round_up(C, 64) channels (the input and the output of the grouped 3x3 64 more: its last group's padded slab starts
    // at channel (G - 1) * group width); the tails are zeroed with the arena below and only ever written with zeros
    int h = H / 2, w = W / 2;
    add_act(n, pl, "stem", N, h, w, c.rn_stem, round_up(c.rn_stem, 64));
    ph[0] = h; pw[0] = w;
    for (int si = 1; si <= 4; ++si) {
      const int cw = c.rn_widths[si - 1], ld = round_up(cw, 64);
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const int sb = b == 1 ? c.rn_strides[si - 1] : 1;
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        add_act(n, pl, p + ".a", N, h, w, cw, ld + 64);
        const int ho = (h - 1) / sb + 1, wo = (w - 1) / sb + 1;
        add_act(n, pl, p + ".b", N, ho, wo, cw, ld + 64);
        if (c.rn_se) {
          add_act(n, pl, p + ".se1", N, ho, wo, round_up(cw / 4, 8), round_up(cw / 4, 64));
          add_act(n, pl, p + ".se2", N, ho, wo, cw, ld);
        }
        if (regnet_has_shortcut(c, si, b)) add_act(n, pl, p + ".ds", N, ho, wo, cw, ld);
        add_act(n, pl, p, N, ho, wo, cw, ld);
        h = ho; w = wo;
      }
      ph[si] = h; pw[si] = w;
    }
  } else {
  add_act(n, pl, "stem", N, H / 2, W / 2, 64);
  add_act(n, pl, "p1", N, H / 4, W / 4, 64);
  int h = H / 4, w = W / 4, inpl = 64;
  ph[0] = h; pw[0] = w;
  for (int li = 1; li <= 4; ++li) {
    int stride = li == 1 ? 1 : 2;
    if (li == 4 && c.stage4_stride == 16) stride = 1;
    for (int b = 0; b < kLayers[li - 1]; ++b) {
      const int s = b == 0 ? stride : 1;
      const int planes = kPlanes[li - 1];
      std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
      add_act(n, pl, p + ".c1", N, h, w, planes);
      const int ho = (h - 1) / s + 1, wo = (w - 1) / s + 1;
      add_act(n, pl, p + ".c2", N, ho, wo, planes);
      if (b == 0) add_act(n, pl, p + ".ds", N, ho, wo, planes * 4);
      add_act(n, pl, p, N, ho, wo, planes * 4);
      h = ho; w = wo; inpl = planes * 4;
    }
    ph[li] = h; pw[li] = w;
  }
  (void)inpl;
  }
  int hq = 0, wq = 0;
  if (c.arch == 0) {
    const int h5 = ph[4], w5 = pw[4];
    add_raw(n, pl, "pooled", (size_t)N * 2048 * 4);
    add_raw(n, pl, "pool_part", (size_t)avgpool_scratch_floats(N, 2048) * 4);
    const char* decs[2] = {"semantic_decoder", "instance_decoder"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      std::string p = decs[d];
      add_raw(n, pl, p + ".poolfeat", (size_t)N * n->aspp_ch * 4);
      add_raw(n, pl, p + ".bias_n", (size_t)N * n->aspp_ch * 4);
      add_act(n, pl, p + ".aspp.cat", N, h5, w5, 4 * n->aspp_ch);
      add_act(n, pl, p + ".aspp", N, h5, w5, n->aspp_ch);
      int xch = n->aspp_ch;
      for (int i = 0; i < c.n_stages; ++i) {
        const int st = c.low_level_stages[i];
        const int lp = d == 0 ? c.low_level_proj_sem[i] : c.low_level_proj_ins[i];
        const int cpad = round_up(xch + lp, 64);
        std::string q = p + ".stage" + std::to_string(i);
        add_act(n, pl, q + ".cat", N, ph[st], pw[st], cpad);
        add_act(n, pl, q + ".dw", N, ph[st], pw[st], cpad);
        add_act(n, pl, q + ".out", N, ph[st], pw[st], n->dec_ch);
        xch = n->dec_ch;
      }
    }
    const int st_last = c.low_level_stages[c.n_stages - 1];
    hq = ph[st_last]; wq = pw[st_last];  // resolution of semantic_x (1/4 for MitoNet)
    EMP_REQUIRE(hq * 4 == H && wq * 4 == W, "the last decoder stage must be at 1/4 resolution (got %dx%d)", hq, wq);

  } else {
    EMP_REQUIRE(H % 128 == 0 && W % 128 == 0, "BiFPN forward: H=%d W=%d must be multiples of 128", H, W);
    const int F = c.fpn_dim;
    hq = ph[1]; wq = pw[1];
    add_act(n, pl, "p2f", N, ph[1], pw[1], F);
    // level sizes P3..P7
    int lh[5], lw[5];
    for (int i = 0; i < 3; ++i) { lh[i] = ph[2 + i]; lw[i] = pw[2 + i]; }
    lh[3] = lh[2] / 2; lw[3] = lw[2] / 2; lh[4] = lh[3] / 2; lw[4] = lw[3] / 2;
    const char* dn[2] = {"semantic", "instance"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      std::string fp = std::string(dn[d]) + "_fpn";
      add_act(n, pl, fp + ".p6pre", N, lh[2], lw[2], F);
      add_act(n, pl, fp + ".in.P6", N, lh[3], lw[3], F);
      add_act(n, pl, fp + ".in.P7", N, lh[4], lw[4], F);
      for (int li = 0; li < c.fpn_layers; ++li) {
        std::string L = fp + ".l" + std::to_string(li);
        for (int lv = 0; lv < 5; ++lv) {
          std::string q = L + ".P" + std::to_string(3 + lv);
          if (li == 0 && lv < 3) { add_act(n, pl, q + ".rtd", N, lh[lv], lw[lv], F); add_act(n, pl, q + ".rbu", N, lh[lv], lw[lv], F); }
          const int FZ = fsplit_on(n, fp + ".") ? 2 * F : F;      // [hi | lo] pair of the fused map (emp_pdl::precise_fsplit)
          add_act(n, pl, q + ".fuse", N, lh[lv], lw[lv], FZ);
          if (lv > 0) add_act(n, pl, q + ".fuseb", N, lh[lv], lw[lv], FZ);   // bottom-up node's fused input (kept for taps)
          add_act(n, pl, q + ".dw", N, lh[lv], lw[lv], F);
          add_act(n, pl, q + ".td", N, lh[lv], lw[lv], F);
          add_act(n, pl, q + ".bu", N, lh[lv], lw[lv], F);
        }
      }
      std::string dp = std::string(dn[d]) + "_decoder";
      int hh = lh[4], ww = lw[4];
      for (int i = 0; i < 5; ++i) {
        hh *= 2; ww *= 2;
        add_act(n, pl, dp + ".cat" + std::to_string(i), N, hh, ww, 2 * F);
      }
      add_act(n, pl, dp + ".dw", N, hq, wq, 2 * F);
      add_act(n, pl, dp + ".out", N, hq, wq, F);
    }
  }
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  const int hc[3] = {n->ncls, 1, 2};
  for (int k = 0; k < 3; ++k) {
    std::string p = heads[k];
    add_act(n, pl, p + ".dw", N, hq, wq, n->dec_ch);
    add_act(n, pl, p + ".pw", N, hq, wq, n->dec_ch);
    add_raw(n, pl, p + ".out", (size_t)N * hc[k] * hq * wq * 4);
  }
  // PointRend
  const int P = c.subdivision_num_points;
  const int ldp = round_up(n->dec_ch + n->ncls, 64);
  int64_t plane_max = 0;
  {
    int hh = hq, ww = wq;
    for (int s = 0; s < RS; ++s) {
      hh *= 2; ww *= 2;
      if (s + 1 < RS) add_raw(n, pl, "pr.sem" + std::to_string(s), (size_t)N * n->ncls * hh * ww * 4);
      plane_max = (int64_t)hh * ww;
    }
  }
  add_raw(n, pl, "pr.keys", (size_t)N * plane_max * 4);
  add_raw(n, pl, "pr.topk", topk_work_bytes(N, plane_max));
  add_raw(n, pl, "pr.idx", (size_t)N * P * 4);
  add_raw(n, pl, "pr.x0", (size_t)N * P * ldp * 2);
  add_raw(n, pl, "pr.x1", (size_t)N * P * ldp * 2);

  const size_t need = pl.off;
  if (need > n->arena_cap) {
    if (n->arena) EMP_CHECK_HIP(hipFree(n->arena));
    n->arena = nullptr;
    n->arena_cap = 0;
    hipError_t e = hipMalloc((void**)&n->arena, need);
    if (e != hipSuccess) {
      set_error("arena: hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
      return EMP_ERR_NOMEM;
    }
    n->arena_cap = need;
  }
  // channel-padding regions must read as exact zeros; ordered on the caller's stream behind whatever still runs on
  // the previous layout (a hipFree above has synchronised the device already)
  EMP_CHECK_HIP(hipMemsetAsync(n->arena, 0, need, stream));
  n->arena_used = need;
  for (auto& kv : n->acts) kv.second.p = (half_t*)(n->arena + kv.second.off);
  n->pN = N; n->pH = H; n->pW = W; n->pRS = RS;
  n->capN = N;
  return EMP_OK;
}

// A smaller batch at the planned image size runs in the SAME layout: every buffer is batch-major, so the first N
// images are a prefix of it and the zero padding stays valid -- no re-plan, no re-zeroing of a multi-GB arena for
// the last partial slice chunk of a stack (and none when the full batch comes back).
void rebatch(emp_pdl* n, int N) {
  for (auto& kv : n->acts) kv.second.N = N;
  n->pN = N;
}

template <typename T>
T* rawp(emp_pdl* n, const std::string& name) {
  return (T*)(n->arena + n->raw.at(name).first);
}

// conv helper: in (channels [0,Cin_pad) of `in`), out channels [coff, coff+Cout) of `out`
int conv(emp_pdl* n, const std::string& wname, const Act& in, int in_coff, const Act& out, int out_coff, int stride,
         int pad, int dil, int act, const Act* res, const float* bias_n, hipStream_t s, int ps_cout = 0,
         const Act* in2 = nullptr, int stride2 = 1, const Act* out2 = nullptr, int out2_coff = 0, int split = 0,
         const std::string* next_name = nullptr, const Act* next_out = nullptr, bool* fused_next = nullptr,
         const Act* out3 = nullptr, int out3_coff = 0, int split3 = 0) {
  DevConv& dc = n->convs.at(wname);
  ConvParams p{};
  if (out3) {     // couts [split3, Cout) go to a third tensor (ConvParams::out3)
    EMP_REQUIRE(out2 && out3->N == out.N && out3->H == out.H && out3->W == out.W && out3_coff + dc.cout - split3 <= out3->ld,
                "%s: third destination mismatch", wname.c_str());
    p.out3 = out3->p + out3_coff; p.out3_ld = out3->ld; p.split3 = split3;
  }
  if (out2) {     // couts [split, Cout) go to a second tensor (ConvParams::out2)
    EMP_REQUIRE(out2->N == out.N && out2->H == out.H && out2->W == out.W &&
                    out2_coff + (out3 ? split3 : dc.cout) - split <= out2->ld,
                "%s: second destination mismatch", wname.c_str());
    p.out2 = out2->p + out2_coff; p.out2_ld = out2->ld; p.split = split;
  }
  if (in2) {      // K-concatenated second source (ConvParams::in2)
    EMP_REQUIRE(dc.cin2_pad > 0 && dc.cin2_pad <= in2->ld && in2->N == in.N, "%s: second source mismatch", wname.c_str());
    p.in2 = in2->p; p.Cin2 = dc.cin2_pad; p.in2_ld = in2->ld; p.H2 = in2->H; p.W2 = in2->W; p.stride2 = stride2;
  } else {
    EMP_REQUIRE(dc.cin2_pad == 0, "%s: packed for two sources", wname.c_str());
  }
  p.in = in.p + in_coff;
  p.wgt = dc.w;
  p.bias = dc.b;
  p.bias_n = bias_n;
  p.res = res ? res->p : nullptr;
  p.res_ld = res ? res->ld : 0;
  p.out = out.p + out_coff;
  p.zero = rawp<half_t>(n, "zero");
  p.N = in.N; p.H = in.H; p.W = in.W; p.Cin = dc.cin_pad; p.in_ld = in.ld;
  p.Cout = dc.cout; p.KH = dc.kh; p.KW = dc.kw; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (in.H + 2 * pad - dil * (dc.kh - 1) - 1) / stride + 1;
  p.Wo = (in.W + 2 * pad - dil * (dc.kw - 1) - 1) / stride + 1;
  const int up = ps_cout ? 2 : 1;
  EMP_REQUIRE(p.Ho * up == out.H && p.Wo * up == out.W && in.N == out.N, "%s: output shape mismatch (%dx%d vs %dx%d)",
              wname.c_str(), p.Ho * up, p.Wo * up, out.H, out.W);
  EMP_REQUIRE(in_coff + dc.cin_pad <= in.ld && out_coff + (ps_cout ? ps_cout : (out2 ? split : dc.cout)) <= out.ld,
              "%s: channel slice out of range", wname.c_str());
  p.out_ld = out.ld;
  p.act = act;
  p.ps_cout = ps_cout;
  p.M = p.N * p.Ho * p.Wo;
  double next_flops = 0.0;
  if (fused_next) *fused_next = false;
  if (next_name && n->fuse_b2b) {
    // the next bottleneck's conv1 (1x1, stride 1, ReLU) computed from this launch's tile while it is in LDS
    const DevConv& nd = n->convs.at(*next_name);
    p.next_w = nd.w; p.next_b = nd.b; p.next_out = next_out->p; p.next_cout = nd.cout; p.next_ld = next_out->ld;
    if (nd.kh == 1 && nd.kw == 1 && nd.cin_pad == dc.cout && nd.cin2_pad == 0 && next_out->N == out.N && next_out->H == out.H &&
        next_out->W == out.W && nd.cout <= next_out->ld && out_coff == 0 && conv_b2b_supported(p)) {
      *fused_next = true;
      next_flops = 2.0 * (double)p.M * nd.cout * (double)nd.cin;
      n->flops += next_flops;
    } else {
      p.next_w = nullptr; p.next_b = nullptr; p.next_out = nullptr; p.next_cout = 0; p.next_ld = 0;
    }
  }
  const double kflop = (double)(dc.cin * dc.kh * dc.kw + (dc.wsplit ? 0 : dc.cin2));      // a hi + lo weight pair is one product
  n->flops += 2.0 * (double)p.M * dc.cout * kflop;
  if (n->layer_log)   // EMP_LAYER_LOG=<file>: one line per MFMA launch, in launch order (tools/layer_roofline.py)
    fprintf(n->layer_log, "conv,%s,%d,%d,%d,%d,%d,%d,%d,%d\n", wname.c_str(), p.M, dc.cin_pad + dc.cin2_pad, dc.cout, dc.kh,
            stride, dil, res ? 1 : 0, in.N * in.H * in.W);
  int variant = 0;
  if (n->pack256 && !p.next_w && conv_uses_256(p)) {
    // the 256 x 256 tile reads its weights from a packed image (whole 128-byte lines per LDS-DMA instruction): made once,
    // at the first launch of this layer that takes the tile (batch-dependent), on the launch's stream
    // (round 6, ADVICE r05: the images are made by emp_pdl_finalize for every layer whose SHAPE the tile takes -- no allocation,
    // no device synchronisation and no failure mode inside a forward; a layer without one runs the tile on its plain weights)
    if (dc.w256) {
      p.wgt = dc.w256;
      variant = 1 << 20;
    }
  }
  if (n->profile && !p.next_w && conv_uses_256(p)) {     // the fused back-to-back launches are another kernel symbol
    if (n->prof_used == n->prof_events.size()) {
      hipEvent_t a, b;
      EMP_CHECK_HIP(hipEventCreate(&a));
      EMP_CHECK_HIP(hipEventCreate(&b));
      n->prof_events.emplace_back(a, b);
    }
    auto& ev = n->prof_events[n->prof_used++];
    EMP_CHECK_HIP(hipEventRecord(ev.first, s));
    const int rc = launch_conv_igemm(p, variant, s);
    EMP_CHECK_HIP(hipEventRecord(ev.second, s));
    n->prof_flops += 2.0 * (double)p.M * dc.cout * kflop;
    return rc;
  }
  return launch_conv_igemm(p, variant, s);
}

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

int run(emp_pdl* n, const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw, int RS,
        int interp, float* o_sem, float* o_ctr, float* o_off, hipStream_t s) {
  const emp_pdl_config& c = n->cfg;
  n->flops = 0.0;
  auto A = [&](const std::string& k) -> Act& { return n->acts.at(k); };
  hipStream_t const s_main = s;
  const bool par = c.ins_decoder && (int64_t)N * H * W <= n->par_limit;
  if (par && !n->aux) {
    EMP_CHECK_HIP(hipStreamCreateWithFlags(&n->aux, hipStreamNonBlocking));
    EMP_CHECK_HIP(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
    EMP_CHECK_HIP(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
  }
  auto fork = [&]() -> int {      // everything both decoders read is enqueued on s_main: the second stream starts here
    if (!par) return EMP_OK;
    EMP_CHECK_HIP(hipEventRecord(n->ev_fork, s_main));
    EMP_CHECK_HIP(hipStreamWaitEvent(n->aux, n->ev_fork, 0));
    return EMP_OK;
  };

  // ---- encoder ----
  std::string xname, pyr[5];
  bool proj_done[8] = {false, false, false, false, false, false, false, false};   // decoder projections written by the encoder
  if (c.encoder == 1) {
    // RegNet (regnet.py:160-166) on the generic conv: a (1x1) -> b (grouped 3x3: one launch per group on its channel
    // slice) -> [per-pixel squeeze-excite gate] -> c (1x1) + shortcut, ReLU
    const Act& st = A("stem");
    RC(launch_stem3x3s2_f16(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("rn.stem.w"), n->f32w.at("rn.stem.b"), c.rn_stem, st.p,
                            st.ld, s));
    n->flops += 2.0 * N * (H / 2) * (W / 2) * (double)c.rn_stem * 9.0;
    xname = "stem";
    pyr[0] = "stem";
    for (int si = 1; si <= 4; ++si) {
      const int cw = c.rn_widths[si - 1], g = c.rn_groups[si - 1], gw = cw / g;
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        const int sb = b == 1 ? c.rn_strides[si - 1] : 1;
        RC(conv(n, p + ".bottleneck.a.0", A(xname), 0, A(p + ".a"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
        // one grouped launch, unless every group alone already fills the chip with the register-weight 3x3 kernel's tiles
        // (stride 1, >= 2048 tiles of 8 x 16 pixels: the big maps of a big batch, where that kernel is the faster one)
        const Act& ain = A(p + ".a");
        const bool per_group = !n->regnet_grouped || (sb == 1 && (int64_t)N * cdiv(ain.H, 8) * cdiv(ain.W, 16) >= n->regnet_group_tiles);
        if (!per_group) {
          // one launch, blockIdx.y = group (conv_igemm_grouped.hip): Cout = the group width exactly, Cin = it padded to 64
          const DevConv& dc = n->convs.at(p + ".bottleneck.b.0#grouped");
          const Act& in = A(p + ".a");
          const Act& out = A(p + ".b");
          ConvParams q{};
          q.in = in.p; q.wgt = dc.w; q.bias = dc.b; q.out = out.p; q.zero = rawp<half_t>(n, "zero");
          q.N = in.N; q.H = in.H; q.W = in.W; q.Cin = dc.cin_pad; q.in_ld = in.ld;
          q.Cout = gw; q.KH = 3; q.KW = 3; q.stride = sb; q.pad = 1; q.dil = 1;
          q.Ho = (in.H + 2 - 3) / sb + 1;
          q.Wo = (in.W + 2 - 3) / sb + 1;
          EMP_REQUIRE(q.Ho == out.H && q.Wo == out.W && in.N == out.N, "%s: grouped 3x3 output shape mismatch", p.c_str());
          q.out_ld = out.ld; q.act = 1; q.M = q.N * q.Ho * q.Wo;
          n->flops += 2.0 * (double)q.M * cw * (double)(gw * 9);
          RC(launch_conv_igemm_grouped(q, g, gw, (long long)gw * 9 * dc.cin_pad, gw, s));
        } else {
          for (int gi = 0; gi < g; ++gi)
            RC(conv(n, p + ".bottleneck.b.0#" + std::to_string(gi), A(p + ".a"), gi * gw, A(p + ".b"), gi * gw, sb, 1, 1, 1, nullptr,
                    nullptr, s));
        }
        if (c.rn_se) {
          RC(conv(n, p + ".bottleneck.se.se.0#pad", A(p + ".b"), 0, A(p + ".se1"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
          RC(conv(n, p + ".bottleneck.se.se.2", A(p + ".se1"), 0, A(p + ".se2"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
          const Act& xb = A(p + ".b");      // the gated map lands in .se2 (the gate's own buffer); .b keeps the pre-gate map
          RC(launch_gate_mul_f16(xb.p, xb.ld, A(p + ".se2").p, A(p + ".se2").ld, (int64_t)N * xb.H * xb.W, cw, s));
        }
        const Act* idn = &A(xname);
        if (regnet_has_shortcut(c, si, b)) {
          RC(conv(n, p + ".downsample.conv.0", A(xname), 0, A(p + ".ds"), 0, sb, 0, 1, 0, nullptr, nullptr, s));
          idn = &A(p + ".ds");
        }
        RC(conv(n, p + ".bottleneck.c.0", A(p + (c.rn_se ? ".se2" : ".b")), 0, A(p), 0, 1, 0, 1, 1, idn, nullptr, s));
        xname = p;
      }
      pyr[si] = xname;
    }
  } else {
    if (n->fuse_stem) {
      // conv1 + bn1 + relu + maxpool in one launch on the matrix pipe; the half-resolution map is never written
      RC(launch_stem_pool(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("stem.w"), n->f32w.at("stem.b"), A("p1").p, s));
    } else {
      RC(launch_stem7x7(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("stem.w"), n->f32w.at("stem.b"), A("stem").p, s));
      RC(launch_maxpool3x3s2(A("stem").p, N, H / 2, W / 2, 64, A("p1").p, s));
    }
    n->flops += 2.0 * N * (H / 2) * (W / 2) * 64.0 * 49.0;
    xname = "p1";
    pyr[0] = "p1";
    bool c1_done = false;      // the coming block's conv1 was computed by the previous block's last launch
    for (int li = 1; li <= 4; ++li) {
      int stride = li == 1 ? 1 : 2, dil = 1;
      if (li == 4 && c.stage4_stride == 16) { stride = 1; dil = 2; }
      for (int b = 0; b < kLayers[li - 1]; ++b) {
        const int sb = b == 0 ? stride : 1;
        std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
        if (!c1_done) {
          // first block of a stage whose input the decoders project too: one conv, three destinations (see create)
          int pi = -1;
          if (b == 0 && c.arch == 0)
            for (int i = 0; i < c.n_stages; ++i)
              if (c.low_level_stages[i] == li - 1 && n->convs.count(p + ".conv1+project." + std::to_string(i))) pi = i;
          if (pi >= 0) {
            const int xch = pi == 0 ? n->aspp_ch : n->dec_ch;      // where the projected channels sit in the concat buffers
            const Act& cb1 = A("semantic_decoder.stage" + std::to_string(pi) + ".cat");
            const Act& cb2 = A("instance_decoder.stage" + std::to_string(pi) + ".cat");
            const int c1 = n->convs.at(p + ".conv1").cout;
            RC(conv(n, p + ".conv1+project." + std::to_string(pi), A(xname), 0, A(p + ".c1"), 0, 1, 0, 1, true, nullptr, nullptr, s,
                    0, nullptr, 1, &cb1, xch, c1, nullptr, nullptr, nullptr, &cb2, xch, c1 + c.low_level_proj_sem[pi]));
            proj_done[pi] = true;
          } else {
            RC(conv(n, p + ".conv1", A(xname), 0, A(p + ".c1"), 0, 1, 0, 1, true, nullptr, nullptr, s));
          }
        }
        c1_done = false;
        RC(conv(n, p + ".conv2", A(p + ".c1"), 0, A(p + ".c2"), 0, sb, dil, dil, true, nullptr, nullptr, s));
        // the conv1 of the block that follows (same layer, or the first block of the next one): a candidate for the
        // back-to-back fusion into this block's last launch (ConvParams::next_*; conv() decides)
        std::string nxt;
        if (b + 1 < kLayers[li - 1]) nxt = "encoder.layer" + std::to_string(li) + "." + std::to_string(b + 1);
        else if (li < 4) nxt = "encoder.layer" + std::to_string(li + 1) + ".0";
        const std::string nxt_w = nxt + ".conv1";
        const bool has_next = !nxt.empty() && n->convs.count(nxt_w);
        const Act* idn = &A(xname);
        if (b == 0 && n->fuse_ds) {
          // relu(bn3(conv3(c2)) + bn(downsample(x))) as ONE GEMM whose K runs over c2's channels and then over x's
          // (sampled with the block's stride): the shortcut map is neither written nor read back
          RC(conv(n, p + ".conv3+ds", A(p + ".c2"), 0, A(p), 0, 1, 0, 1, true, nullptr, nullptr, s, 0, &A(xname), sb, nullptr, 0, 0,
                  has_next ? &nxt_w : nullptr, has_next ? &A(nxt + ".c1") : nullptr, &c1_done));
          xname = p;
          continue;
        }
        if (b == 0) {
          RC(conv(n, p + ".downsample.0", A(xname), 0, A(p + ".ds"), 0, sb, 0, 1, false, nullptr, nullptr, s));
          idn = &A(p + ".ds");
        }
        RC(conv(n, p + ".conv3", A(p + ".c2"), 0, A(p), 0, 1, 0, 1, true, idn, nullptr, s, 0, nullptr, 1, nullptr, 0, 0,
                has_next ? &nxt_w : nullptr, has_next ? &A(nxt + ".c1") : nullptr, &c1_done));
        xname = p;
      }
      pyr[li] = xname;
    }
  }
  std::string dec_out[2];
  if (c.arch == 1) {
    // ---- BiFPN decoders (bifpn.py:185-236, panoptic_bifpn.py:70-82) ----
    const int F = c.fpn_dim;
    const half_t* zero = rawp<half_t>(n, "zero");
    RC(conv(n, "p2_resample.conv.0", A(pyr[1]), 0, A("p2f"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
    RC(fork());
    const char* dn[2] = {"semantic", "instance"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      hipStream_t s = (par && d == 1) ? n->aux : s_main;
      const std::string fp = std::string(dn[d]) + "_fpn";
      // P6 / P7 (bifpn.py:187-188)
      RC(conv(n, fp + ".p6_resample.conv.0", A(pyr[4]), 0, A(fp + ".p6pre"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
      {
        const Act& a6 = A(fp + ".p6pre");
        RC(launch_maxpool3x3s2(a6.p, N, a6.H, a6.W, F, A(fp + ".in.P6").p, s));
        const Act& b6 = A(fp + ".in.P6");
        RC(launch_maxpool3x3s2(b6.p, N, b6.H, b6.W, F, A(fp + ".in.P7").p, s));
      }
      std::string feat[5] = {pyr[2], pyr[3], pyr[4], fp + ".in.P6", fp + ".in.P7"};   // P3..P7
      for (int li = 0; li < c.fpn_layers; ++li) {
        const std::string L = fp + ".l" + std::to_string(li);
        const std::string pre = fp + ".bifpns." + std::to_string(li);
        auto node = [&](const std::string& dirpre, const std::string& q, const half_t* a, const half_t* b2,
                        const half_t* c3, float ca, float cb, float cc, int mode, const std::string& outname) -> int {
          const Act& fz = A(q + (mode ? ".fuseb" : ".fuse"));
          const std::string pwn = dirpre + ".after_combines.0.0.sepconv.1";
          const DevConv& pwc = n->convs.at(pwn);
          const Act& on = A(outname);
          n->flops += 2.0 * 9.0 * (double)N * fz.H * fz.W * F;
          if (fz.ld == 2 * F && n->f16w.count(pwn + ".packedpd") && on.ld == pwc.cout) {
            // fused map as an fp16 hi + lo pair in channels [0, F) | [F, 2F); the node's block reads all 2F with duplicated
            // taps and pointwise weights (fsplit_on): always fused, one rounding sequence at every batch size
            RC(launch_fuse_combine(a, b2, c3, ca, cb, cc, mode, N, fz.H, fz.W, F, fz.p, s, fz.p + F, 2 * F));
            RC(launch_sepconvp(fz.p, N, fz.H, fz.W, 2 * F, fz.ld, n->f32w.at(dirpre + ".after_combines.0.0.sepconv.0.f32d"),
                               n->f16w.at(pwn + ".packedpd"), pwc.b, pwc.cout, 2, on.p, on.ld, nullptr, nullptr, 0, nullptr, 0,
                               zero, s, 3, pwc.wsplit ? 1 : 0));
            n->flops += 2.0 * (double)N * fz.H * fz.W * pwc.cout * (double)pwc.cin;
            if (n->layer_log) fprintf(n->layer_log, "sepconvp,%s,%d,%d,%d,3,1,1,0,%d\n", pwn.c_str(), N * fz.H * fz.W, 2 * F, pwc.cout, N * fz.H * fz.W);
            return EMP_OK;
          }
          RC(launch_fuse_combine(a, b2, c3, ca, cb, cc, mode, N, fz.H, fz.W, F, fz.p, s));
          // the node's 3x3 block with the exact depthwise half (precise_node): fused at EVERY size -- the kernel is the
          // only implementation of this rounding sequence, so no tile-count threshold may pick another one
          if (n->fuse_sepconv && precise_node(n, pwn) && n->f16w.count(pwn + ".packedp") &&
              pwc.cin_pad == F && fz.ld == F && on.ld == pwc.cout && sepconvp_supported(F, pwc.cout, 0)) {
            RC(launch_sepconvp(fz.p, N, fz.H, fz.W, F, fz.ld, n->f32w.at(dirpre + ".after_combines.0.0.sepconv.0.f32"),
                               n->f16w.at(pwn + ".packedp"), pwc.b, pwc.cout, 2, on.p, on.ld, nullptr, nullptr, 0, nullptr, 0,
                               zero, s, 3, pwc.wsplit ? 1 : 0));
            n->flops += 2.0 * (double)N * fz.H * fz.W * pwc.cout * (double)pwc.cin;
            if (n->layer_log) fprintf(n->layer_log, "sepconvp,%s,%d,%d,%d,3,1,1,0,%d\n", pwn.c_str(), N * fz.H * fz.W, F, pwc.cout, N * fz.H * fz.W);
            return EMP_OK;
          }
          // depthwise 3x3 -> pointwise -> BN -> SiLU in one launch (sepconv.hip, KS = 3) once the map has a tile per CU
          if (n->fuse_sepconv && n->f16w.count(pwn + ".packed") && pwc.cin_pad == F && fz.ld == F && on.ld == pwc.cout &&
              sepconv5_supported(F, pwc.cout, 0) && (int64_t)N * ((fz.H + 7) / 8) * ((fz.W + 15) / 16) >= n->sepconv_min_tiles) {
            RC(launch_sepconv5(fz.p, N, fz.H, fz.W, F, fz.ld, n->f16w.at(dirpre + ".after_combines.0.0.sepconv.0"),
                               n->f16w.at(pwn + ".packed"), pwc.b, pwc.cout, 2, on.p, on.ld, nullptr, nullptr, 0, nullptr, 0,
                               zero, s, 3));
            n->flops += 2.0 * (double)N * fz.H * fz.W * pwc.cout * (double)pwc.cin;
            if (n->layer_log) fprintf(n->layer_log, "sepconv,%s,%d,%d,%d,3,1,1,0,%d\n", pwn.c_str(), N * fz.H * fz.W, F, pwc.cout, N * fz.H * fz.W);
            return EMP_OK;
          }
          RC(launch_dwconv(fz.p, N, fz.H, fz.W, F, F, n->f16w.at(dirpre + ".after_combines.0.0.sepconv.0"), 3,
                           A(q + ".dw").p, F, zero, s));
          return conv(n, pwn, A(q + ".dw"), 0, on, 0, 1, 0, 1, 2, nullptr, nullptr, s);
        };
        // top-down: P7 -> P3 (bifpn.py:47-69); level index lv: 0=P3 .. 4=P7
        {
          const std::string dp = pre + ".top_down_fpn";
          const float* w = n->fusew.at(dp + ".weights").data();
          std::string td_prev = feat[4];
          for (int i = 0; i < 4; ++i) {
            const int lv = 3 - i;
            const std::string q = L + ".P" + std::to_string(3 + lv);
            std::string hi = feat[lv];
            const std::string rk = dp + ".resamplings." + std::to_string(i) + ".conv.0";
            if (n->convs.count(rk)) {
              RC(conv(n, rk, A(feat[lv]), 0, A(q + ".rtd"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
              hi = q + ".rtd";
            }
            const float den = w[i] + w[i + 1] + 1e-4f;
            RC(node(dp, q, A(td_prev).p, A(hi).p, nullptr, w[i] / den, w[i + 1] / den, 0.f, 0, q + ".td"));
            td_prev = q + ".td";
          }
        }
        // bottom-up: P3' -> P7 (bifpn.py:103-137)
        {
          const std::string dp = pre + ".bottom_up_fpn";
          const float* w = n->fusew.at(dp + ".weights").data();
          std::string bu_prev = L + ".P3.td";
          std::string newfeat[5];
          newfeat[0] = bu_prev;
          for (int i = 0; i < 4; ++i) {
            const int lv = i + 1;
            const std::string q = L + ".P" + std::to_string(3 + lv);
            std::string lo = feat[lv];
            const std::string rk = dp + ".resamplings." + std::to_string(i) + ".conv.0";
            if (n->convs.count(rk)) {
              RC(conv(n, rk, A(feat[lv]), 0, A(q + ".rbu"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
              lo = q + ".rbu";
            }
            if (i < 3) {
              const float den = w[i] + w[i + 1] + w[i + 2] + 1e-4f;
              RC(node(dp, q, A(bu_prev).p, A(lo).p, A(q + ".td").p, w[i] / den, w[i + 1] / den, w[i + 2] / den, 1,
                      q + ".bu"));
            } else {
              const float den = w[i] + w[i + 1] + 1e-4f;
              RC(node(dp, q, A(bu_prev).p, A(lo).p, nullptr, w[i] / den, w[i + 1] / den, 0.f, 1, q + ".bu"));
            }
            bu_prev = q + ".bu";
            newfeat[lv] = bu_prev;
          }
          for (int lv = 0; lv < 5; ++lv) feat[lv] = newfeat[lv];
        }
      }
      // decoder: 5 x (ConvTranspose k2 s2 + BN + ReLU, concat skip), 5x5 separable fusion (bifpn.py:226-236)
      const std::string dp = std::string(dn[d]) + "_decoder";
      const std::string skips[5] = {feat[3], feat[2], feat[1], feat[0], "p2f"};
      std::string x = feat[4];
      for (int i = 0; i < 5; ++i) {
        const Act& cat = A(dp + ".cat" + std::to_string(i));
        const std::string un = dp + ".upsamplings." + std::to_string(i) + ".0";
        RC(conv(n, un, A(x), 0, cat, 0, 1, 0, 1, 1, nullptr, nullptr, s, F, n->convs.at(un).wsplit ? &A(x) : nullptr, 1));
        const Act& sk = A(skips[i]);
        RC(launch_bilinear_ac(sk.p, N, sk.H, sk.W, F, sk.ld, cat.p + F, cat.H, cat.W, cat.ld, s));  // same size: strided copy
        x = dp + ".cat" + std::to_string(i);
      }
      const Act& cat = A(x);
      n->flops += 2.0 * 25.0 * (double)N * cat.H * cat.W * 2 * F;
      {
        const DevConv& pwc = n->convs.at(dp + ".fusion.0.sepconv.1");
        const Act& so = A(dp + ".out");
        if (n->fuse_sepconv && precise_layer(n, dp + ".fusion.0") && n->f16w.count(dp + ".fusion.0.sepconv.1.packedp") &&
            cat.ld == 2 * F && pwc.cin_pad == 2 * F && so.ld == pwc.cout && sepconvp_supported(2 * F, pwc.cout, 0)) {
          // the decoder that feeds the centre heat-map: the block with the exact depthwise half (sepconv_precise.hip)
          RC(launch_sepconvp(cat.p, N, cat.H, cat.W, 2 * F, cat.ld, n->f32w.at(dp + ".fusion.0.sepconv.0.f32"),
                             n->f16w.at(dp + ".fusion.0.sepconv.1.packedp"), pwc.b, pwc.cout, 1, so.p, so.ld, nullptr, nullptr,
                             0, nullptr, 0, zero, s, 5, pwc.wsplit ? 1 : 0));
          n->flops += 2.0 * (double)N * cat.H * cat.W * pwc.cout * (double)pwc.cin;
          if (n->layer_log) fprintf(n->layer_log, "sepconvp,%s,%d,%d,%d,5,1,1,0,%d\n", (dp + ".fusion.0").c_str(), N * cat.H * cat.W, 2 * F, pwc.cout, N * cat.H * cat.W);
        } else if (n->fuse_sepconv && n->f16w.count(dp + ".fusion.0.sepconv.1.packed") && cat.ld == 2 * F && pwc.cin_pad == 2 * F &&
            so.ld == pwc.cout && sepconv5_supported(2 * F, pwc.cout, 0)) {
          RC(launch_sepconv5(cat.p, N, cat.H, cat.W, 2 * F, cat.ld, n->f16w.at(dp + ".fusion.0.sepconv.0"),
                             n->f16w.at(dp + ".fusion.0.sepconv.1.packed"), pwc.b, pwc.cout, 1, so.p, so.ld, nullptr, nullptr,
                             0, nullptr, 0, zero, s));
          n->flops += 2.0 * (double)N * cat.H * cat.W * pwc.cout * (double)pwc.cin;
          if (n->layer_log) fprintf(n->layer_log, "sepconv,%s,%d,%d,%d,5,1,1,0,%d\n", (dp + ".fusion.0").c_str(), N * cat.H * cat.W, 2 * F, pwc.cout, N * cat.H * cat.W);
        } else {
          RC(launch_dwconv(cat.p, N, cat.H, cat.W, 2 * F, cat.ld, n->f16w.at(dp + ".fusion.0.sepconv.0"), 5, A(dp + ".dw").p,
                           2 * F, zero, s));
          RC(conv(n, dp + ".fusion.0.sepconv.1", A(dp + ".dw"), 0, so, 0, 1, 0, 1, 1, nullptr, nullptr, s));
        }
      }
      dec_out[d] = dp + ".out";
    }
  } else {
  const Act& p5 = A(pyr[4]);

  // ---- decoders ----
  RC(launch_avgpool(p5.p, N, p5.H * p5.W, p5.C, p5.ld, rawp<float>(n, "pooled"), rawp<float>(n, "pool_part"), s));
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  // ASPP branches of BOTH decoders, one conv per branch (couts [0, aspp_ch) -> semantic concat buffer, the rest ->
  // instance concat buffer) when the merged weights exist (finalize) and the launch goes to the 256x256 tile
  bool aspp_done[4] = {false, false, false, false};
  const bool aspp_merged = c.ins_decoder && (int64_t)((N * p5.H * p5.W + 255) / 256) * (2 * n->aspp_ch / 256) >= 192;
  if (aspp_merged)      // smaller problems keep the separate launches (deep-ring 64x64 tile, two streams)
    for (int i = 0; i <= 3; ++i) {
      const std::string pm = "decoders.aspp.convs." + std::to_string(i) + ".0";
      if (!n->convs.count(pm)) continue;
      const int r = i == 0 ? 1 : c.atrous_rates[i - 1];
      const Act& cat0 = A(std::string(decs[0]) + ".aspp.cat");
      const Act& cat1 = A(std::string(decs[1]) + ".aspp.cat");
      RC(conv(n, pm, p5, 0, cat0, i * n->aspp_ch, 1, i == 0 ? 0 : r, r, true, nullptr, nullptr, s, 0, nullptr, 1, &cat1,
              i * n->aspp_ch, n->aspp_ch));
      aspp_done[i] = true;
    }
  // The merged low-level projections ("decoders.project.i": one pass over a pyramid level writes BOTH decoders' concat
  // buffers) that the encoder did not already run: they read encoder maps only and must be on s_main BEFORE the
  // instance side forks onto the second stream -- after the fork nothing would order the instance decoder's read of
  // its concat buffer behind this launch (EMP_FUSE_PROJ=0, low_level_stages with stage 0, odd conv1 shapes).
  if (c.ins_decoder) {
    int xch0 = n->aspp_ch;
    for (int i = 0; i < c.n_stages; ++i) {
      const std::string pm = "decoders.project." + std::to_string(i);
      if (!proj_done[i] && n->convs.count(pm)) {
        const Act& cb1 = A(std::string(decs[0]) + ".stage" + std::to_string(i) + ".cat");
        const Act& cb2 = A(std::string(decs[1]) + ".stage" + std::to_string(i) + ".cat");
        RC(conv(n, pm, A(pyr[c.low_level_stages[i]]), 0, cb1, xch0, 1, 0, 1, true, nullptr, nullptr, s, 0, nullptr, 1, &cb2, xch0,
                c.low_level_proj_sem[i]));
        proj_done[i] = true;
      }
      xch0 = n->dec_ch;
    }
  }
  RC(fork());
  for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
    hipStream_t s = (par && d == 1) ? n->aux : s_main;
    std::string p = decs[d];
    float* poolfeat = rawp<float>(n, p + ".poolfeat");
    float* bias_n = rawp<float>(n, p + ".bias_n");
    RC(launch_gemv(rawp<float>(n, "pooled"), N, p5.C, n->f32w.at(p + ".pool.w"), nullptr, n->aspp_ch, 1, poolfeat, s));
    RC(launch_gemv(poolfeat, N, n->aspp_ch, n->f32w.at(p + ".projpool.w"), nullptr, n->aspp_ch, 0, bias_n, s));
    const Act& cat = A(p + ".aspp.cat");
    if (!aspp_done[0]) RC(conv(n, p + ".aspp.convs.0.0", p5, 0, cat, 0, 1, 0, 1, true, nullptr, nullptr, s));
    for (int i = 1; i <= 3; ++i) {
      if (aspp_done[i]) continue;
      const int r = c.atrous_rates[i - 1];
      RC(conv(n, p + ".aspp.convs." + std::to_string(i) + ".0", p5, 0, cat, i * n->aspp_ch, 1, r, r, true, nullptr,
              nullptr, s));
    }
    RC(conv(n, p + ".aspp.project.0", cat, 0, A(p + ".aspp"), 0, 1, 0, 1, true, nullptr, bias_n, s));
    std::string x = p + ".aspp";
    int xch = n->aspp_ch;
    for (int i = 0; i < c.n_stages; ++i) {
      const int st = c.low_level_stages[i];
      std::string q = p + ".stage" + std::to_string(i);
      const Act& cb = A(q + ".cat");
      const Act& xa = A(x);
      RC(launch_bilinear_ac(xa.p, N, xa.H, xa.W, xch, xa.ld, cb.p, cb.H, cb.W, cb.ld, s));
      const std::string pm = "decoders.project." + std::to_string(i);
      if (proj_done[i]) {
        // written by the encoder's merged launch (conv1 of the next stage + both projections) or by the merged
        // projection in front of the fork
      } else {
        RC(conv(n, p + ".project." + std::to_string(i) + ".0", A(pyr[st]), 0, cb, xch, 1, 0, 1, true, nullptr, nullptr, s));
      }
      const std::string fz = p + ".fuse." + std::to_string(i) + ".0.sepconv.";
      n->flops += 2.0 * 25.0 * (double)N * cb.H * cb.W * (xch + n->convs.at(p + ".project." + std::to_string(i) + ".0").cout);
      {
        const DevConv& pwc = n->convs.at(fz + "1");
        const Act& so = A(q + ".out");
        if (n->fuse_sepconv && precise_layer(n, p + ".fuse." + std::to_string(i) + ".0") && n->f16w.count(fz + "1.packedp") &&
            pwc.cin_pad == cb.ld && so.ld == pwc.cout && sepconvp_supported(cb.ld, pwc.cout, 0)) {
          // the decoder that feeds the centre heat-map: the block with the exact depthwise half (sepconv_precise.hip)
          RC(launch_sepconvp(cb.p, N, cb.H, cb.W, cb.ld, cb.ld, n->f32w.at(fz + "0.f32"), n->f16w.at(fz + "1.packedp"), pwc.b,
                             pwc.cout, 1, so.p, so.ld, nullptr, nullptr, 0, nullptr, 0, rawp<half_t>(n, "zero"), s));
          n->flops += 2.0 * (double)N * cb.H * cb.W * pwc.cout * (double)pwc.cin;
          if (n->layer_log) fprintf(n->layer_log, "sepconvp,%s,%d,%d,%d,5,1,1,0,%d\n", fz.c_str(), N * cb.H * cb.W, cb.ld, pwc.cout, N * cb.H * cb.W);
        } else if (n->fuse_sepconv && n->f16w.count(fz + "1.packed") && pwc.cin_pad == cb.ld && so.ld == pwc.cout &&
            sepconv5_supported(cb.ld, pwc.cout, 0)) {
          // depthwise 5x5 -> pointwise -> bias -> ReLU in one launch (sepconv.hip); the depthwise map stays in LDS
          RC(launch_sepconv5(cb.p, N, cb.H, cb.W, cb.ld, cb.ld, n->f16w.at(fz + "0"), n->f16w.at(fz + "1.packed"), pwc.b,
                             pwc.cout, 1, so.p, so.ld, nullptr, nullptr, 0, nullptr, 0, rawp<half_t>(n, "zero"), s));
          n->flops += 2.0 * (double)N * cb.H * cb.W * pwc.cout * (double)pwc.cin;
          if (n->layer_log) fprintf(n->layer_log, "sepconv,%s,%d,%d,%d,5,1,1,0,%d\n", fz.c_str(), N * cb.H * cb.W, cb.ld, pwc.cout, N * cb.H * cb.W);
        } else {
          RC(launch_dwconv(cb.p, N, cb.H, cb.W, cb.ld, cb.ld, n->f16w.at(fz + "0"), 5, A(q + ".dw").p, cb.ld,
                           rawp<half_t>(n, "zero"), s));
          RC(conv(n, fz + "1", A(q + ".dw"), 0, so, 0, 1, 0, 1, true, nullptr, nullptr, s));
        }
      }
      x = q + ".out";
      xch = n->dec_ch;
    }
    dec_out[d] = x;
  }
  }
  if (!c.ins_decoder) dec_out[1] = dec_out[0];
  const Act& semx = A(dec_out[0]);
  const Act& insx = A(dec_out[1]);
  const int hq = semx.H, wq = semx.W;

  // ---- heads ----
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  const int hc[3] = {n->ncls, 1, 2};
  float* head_out[3];
  for (int k = 0; k < 3; ++k) {
    hipStream_t s = (par && k > 0) ? n->aux : s_main;
    std::string p = heads[k];
    const Act& xin = k == 0 ? semx : insx;
    n->flops += 2.0 * 25.0 * (double)N * hq * wq * n->dec_ch;
    float* dst = rawp<float>(n, p + ".out");
    if (k == 1 && !interp) dst = o_ctr;
    if (k == 2 && !interp) dst = o_off;
    head_out[k] = dst;
    const DevConv& pwc = n->convs.at(p + ".head.0.0.sepconv.1");
    if (n->fuse_sepconv && precise_layer(n, p + ".head.0.0") && n->f16w.count(p + ".head.0.0.sepconv.1.packedp") &&
        xin.C == n->dec_ch && pwc.cin_pad == n->dec_ch && pwc.cout == n->dec_ch && hc[k] <= 2 &&
        sepconvp_supported(n->dec_ch, pwc.cout, hc[k])) {
      // the centre head with the exact depthwise half (sepconv_precise.hip)
      RC(launch_sepconvp(xin.p, N, hq, wq, n->dec_ch, xin.ld, n->f32w.at(p + ".head.0.0.sepconv.0.f32"),
                         n->f16w.at(p + ".head.0.0.sepconv.1.packedp"), pwc.b, pwc.cout, 1, nullptr, 0, n->f32w.at(p + ".head.1.w"),
                         n->f32w.at(p + ".head.1.b"), hc[k], dst, (int64_t)hq * wq, rawp<half_t>(n, "zero"), s, 5, pwc.wsplit ? 1 : 0));
      n->flops += 2.0 * (double)N * hq * wq * pwc.cout * (double)pwc.cin;
      if (n->layer_log) fprintf(n->layer_log, "sepheadp,%s,%d,%d,%d,5,1,1,0,%d\n", p.c_str(), N * hq * wq, n->dec_ch, pwc.cout, N * hq * wq);
    } else if (n->fuse_sepconv && n->f16w.count(p + ".head.0.0.sepconv.1.packed") && xin.C == n->dec_ch &&
        pwc.cin_pad == n->dec_ch && pwc.cout == n->dec_ch && hc[k] <= 2 &&
        sepconv5_supported(n->dec_ch, pwc.cout, hc[k])) {
      // head.0 (depthwise 5x5 -> pointwise -> ReLU) and head.1 (1x1 -> hc planes) in one launch: neither the
      // depthwise nor the dec_ch-channel map reaches HBM (sepconv.hip)
      RC(launch_sepconv5(xin.p, N, hq, wq, n->dec_ch, xin.ld, n->f16w.at(p + ".head.0.0.sepconv.0"),
                         n->f16w.at(p + ".head.0.0.sepconv.1.packed"), pwc.b, pwc.cout, 1, nullptr, 0, n->f32w.at(p + ".head.1.w"), n->f32w.at(p + ".head.1.b"), hc[k],
                         dst, (int64_t)hq * wq, rawp<half_t>(n, "zero"), s));
      n->flops += 2.0 * (double)N * hq * wq * pwc.cout * (double)pwc.cin;
      if (n->layer_log) fprintf(n->layer_log, "sephead,%s,%d,%d,%d,5,1,1,0,%d\n", p.c_str(), N * hq * wq, n->dec_ch, pwc.cout, N * hq * wq);
    } else {
      RC(launch_dwconv(xin.p, N, hq, wq, n->dec_ch, xin.ld, n->f16w.at(p + ".head.0.0.sepconv.0"), 5, A(p + ".dw").p,
                       n->dec_ch, rawp<half_t>(n, "zero"), s));
      RC(conv(n, p + ".head.0.0.sepconv.1", A(p + ".dw"), 0, A(p + ".pw"), 0, 1, 0, 1, true, nullptr, nullptr, s));
      RC(launch_head1x1(A(p + ".pw").p, N, hq * wq, n->dec_ch, n->dec_ch, n->f32w.at(p + ".head.1.w"),
                        n->f32w.at(p + ".head.1.b"), hc[k], dst, (int64_t)hq * wq, nullptr, s));
    }
    n->flops += 2.0 * (double)N * hq * wq * n->dec_ch * hc[k];
  }
  if (interp) {
    hipStream_t s = par ? n->aux : s_main;
    RC(launch_bilinear_ac_f32_nchw(head_out[1], N * 1, hq, wq, o_ctr, 4, s));
    RC(launch_bilinear_ac_f32_nchw(head_out[2], N * 2, hq, wq, o_off, 4, s));
  }
  if (par) EMP_CHECK_HIP(hipEventRecord(n->ev_join, n->aux));      // joined at the end: PointRend overlaps the instance heads

  // ---- PointRend subdivision ----
  const int P = c.subdivision_num_points;
  const int ldp = round_up(n->dec_ch + n->ncls, 64);
  const float* coarse = head_out[0];
  const float* cur = coarse;
  int hh = hq, ww = wq;
  uint32_t* keys = rawp<uint32_t>(n, "pr.keys");
  int32_t* idx = rawp<int32_t>(n, "pr.idx");
  half_t* X[2] = {rawp<half_t>(n, "pr.x0"), rawp<half_t>(n, "pr.x1")};
  for (int st = 0; st < RS; ++st) {
    float* nxt = (st + 1 < RS) ? rawp<float>(n, "pr.sem" + std::to_string(st)) : o_sem;
    RC(launch_upsample2x_keys(cur, N, n->ncls, hh, ww, nxt, keys, s));
    hh *= 2; ww *= 2;
    const int64_t plane = (int64_t)hh * ww;
    const int k = (int)(plane < P ? plane : P);
    RC(launch_topk_smallest(keys, N, plane, k, rawp<char>(n, "pr.topk"), n->raw.at("pr.topk").second, idx, s));
    if (n->fuse_pr && pr_mlp_supported(n->dec_ch, ldp, n->ncls, c.num_fc)) {
      // sampling + fc layers + predictor + scatter in one launch: the (points x ldp) rows stay in LDS
      const half_t* fw_[4];
      const float* fb_[4];
      for (int f = 0; f < c.num_fc; ++f) {
        const DevConv& dc = n->convs.at("semantic_pr.point_head.fc_layers." + std::to_string(f) + ".0");
        fw_[f] = dc.w; fb_[f] = dc.b;
        n->flops += 2.0 * (double)N * k * dc.cout * (double)dc.cin;
      }
      RC(launch_pr_mlp(semx.p, N, hq, wq, n->dec_ch, semx.ld, coarse, n->ncls, idx, k, hh, ww, fw_, fb_, c.num_fc, ldp,
                       n->f32w.at("pr.predictor.w"), n->f32w.at("pr.predictor.b"), nxt, plane, s));
      n->flops += 2.0 * (double)N * k * ldp * n->ncls;
      cur = nxt;
      continue;
    }
    RC(launch_point_features(semx.p, N, hq, wq, n->dec_ch, semx.ld, coarse, n->ncls, idx, k, hh, ww, X[0], X[1], ldp, s));
    Act xa[2];
    for (int j = 0; j < 2; ++j) {
      xa[j].p = X[j]; xa[j].N = 1; xa[j].H = 1; xa[j].W = N * k; xa[j].C = ldp; xa[j].ld = ldp;
    }
    int curx = 0;
    for (int f = 0; f < c.num_fc; ++f) {
      RC(conv(n, "semantic_pr.point_head.fc_layers." + std::to_string(f) + ".0", xa[curx], 0, xa[curx ^ 1], 0, 1, 0, 1,
              true, nullptr, nullptr, s));
      curx ^= 1;
    }
    RC(launch_head1x1(X[curx], N, k, ldp, ldp, n->f32w.at("pr.predictor.w"), n->f32w.at("pr.predictor.b"), n->ncls, nxt,
                      plane, idx, s));
    n->flops += 2.0 * (double)N * k * ldp * n->ncls;
    cur = nxt;
  }
  // the caller's stream owns every output again once the instance side has finished
  if (par) EMP_CHECK_HIP(hipStreamWaitEvent(s_main, n->ev_join, 0));
  return EMP_OK;
}

// ======================================================================================================================
// fp32 reference mode (round 4).  The same layer schedule as run() -- panoptic_deeplab.py:194-250 / panoptic_bifpn.py:147-161
// in eval -- with every map and weight in fp32 and no layer fusion: one generic exact-fp32 MFMA conv (ref32.hip), the
// depthwise / pooling / resampling layers on the fp32 vector pipe.  The ASPP pooling branch still enters the projection
// as a per-image bias (exact algebra, aspp.py:45-48,99-102).  Slow by design: the device-side fp32 comparator.
// ======================================================================================================================
struct T32 { float* p = nullptr; int N = 0, H = 0, W = 0, C = 0, ld = 0; int fmt = 0; };      // fmt 1: an hl32 map (conv16x3p.hip), same bytes

int buf32(emp_pdl* n, const std::string& key, size_t floats, float** out) {
  auto it = n->pool32.find(key);
  if (it == n->pool32.end() || it->second.second < floats) {
    if (it != n->pool32.end()) { EMP_CHECK_HIP(hipFree(it->second.first)); n->pool32.erase(it); }
    float* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, floats * sizeof(float) + 64);
    if (e != hipSuccess) {
      set_error("fp32 mode: hipMalloc(%zu bytes) for '%s' failed: %s", floats * sizeof(float), key.c_str(), hipGetErrorString(e));
      return EMP_ERR_NOMEM;
    }
    EMP_CHECK_HIP(hipMemset(d, 0, floats * sizeof(float)));      // channel pads must read as zeros
    n->pool32[key] = {d, floats};
    *out = d;
    return EMP_OK;
  }
  *out = it->second.first;
  return EMP_OK;
}

int t32(emp_pdl* n, const std::string& key, int N, int H, int W, int C, T32* t) {
  t->N = N; t->H = H; t->W = W; t->C = C; t->ld = C;
  return buf32(n, key, (size_t)N * H * W * C, &t->p);
}

// OIHW / OIW fp32 -> [O][KH*KW][I16] fp32 (+ bias); cin_to: pad the input channels to this many (PointRend rows)
int pack32(emp_pdl* n, const std::string& name, int cin_to = 0, const HostParam* src = nullptr) {
  const HostParam& hp = src ? *src : n->params.at(name);
  EMP_REQUIRE(hp.shape.size() == 4 || hp.shape.size() == 3, "%s: conv weight must be 3-d or 4-d", name.c_str());
  emp_pdl::W32 w;
  w.cout = (int)hp.shape[0];
  w.cin = (int)hp.shape[1];
  w.kh = hp.shape.size() == 4 ? (int)hp.shape[2] : 1;
  w.kw = hp.shape.size() == 4 ? (int)hp.shape[3] : 1;
  w.cin16 = cin_to ? cin_to : round_up(w.cin, 16);
  EMP_REQUIRE(w.cin16 >= w.cin && w.cin16 % 16 == 0, "%s: bad channel padding", name.c_str());
  const int kt = w.kh * w.kw;
  std::vector<float> pk((size_t)w.cout * kt * w.cin16, 0.f);
  for (int o = 0; o < w.cout; ++o)
    for (int i = 0; i < w.cin; ++i)
      for (int t = 0; t < kt; ++t) pk[((size_t)o * kt + t) * w.cin16 + i] = hp.w[((size_t)o * w.cin + i) * kt + t];
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(float), &d);
  if (rc) return rc;
  w.w = (float*)d;
  rc = dev_upload(n, hp.b.data(), hp.b.size() * sizeof(float), &d);
  if (rc) return rc;
  w.b = (float*)d;
  n->w32[name] = w;
  return EMP_OK;
}

// fp16x3 mode: conv3 and the projection shortcut of a bottleneck as one fp32 weight matrix [Cout][cin3_16 | cin_ds_16] with
// the summed (folded BN) bias -- relu(conv3(c2) + downsample(x)) as a single convolution over the concatenated K
// (Conv32::in2; the fp16 engine's pack_conv3_ds): the shortcut map is neither written nor read back
int pack32_conv3_ds(emp_pdl* n, const std::string& block) {
  const HostParam& h3 = n->params.at(block + ".conv3");
  const HostParam& hd = n->params.at(block + ".downsample.0");
  EMP_REQUIRE(h3.shape.size() == 4 && hd.shape.size() == 4 && h3.shape[0] == hd.shape[0] && h3.shape[2] == 1 && hd.shape[2] == 1 &&
                  h3.b.size() == hd.b.size(), "%s: conv3 / downsample shapes do not match", block.c_str());
  emp_pdl::W32 w;
  w.cout = (int)h3.shape[0];
  w.cin = (int)h3.shape[1];
  w.cin16 = round_up(w.cin, 16);
  w.cin2 = (int)hd.shape[1];
  w.cin2_16 = round_up(w.cin2, 16);
  const size_t K = (size_t)w.cin16 + w.cin2_16;
  std::vector<float> pk((size_t)w.cout * K, 0.f), b((size_t)w.cout);
  for (int o = 0; o < w.cout; ++o) {
    for (int i = 0; i < w.cin; ++i) pk[o * K + i] = h3.w[(size_t)o * w.cin + i];
    for (int i = 0; i < w.cin2; ++i) pk[o * K + w.cin16 + i] = hd.w[(size_t)o * w.cin2 + i];
    b[o] = h3.b[o] + hd.b[o];
  }
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(float), &d);
  if (rc) return rc;
  w.w = (float*)d;
  rc = dev_upload(n, b.data(), b.size() * sizeof(float), &d);
  if (rc) return rc;
  w.b = (float*)d;
  n->w32[block + ".conv3+ds"] = w;
  return EMP_OK;
}

// depthwise (C,1,k,k) -> [k*k][C] fp32
int pack32_dw(emp_pdl* n, const std::string& name, int cpad) {
  const HostParam& hp = n->params.at(name);
  const int C = (int)hp.shape[0], KK = (int)(hp.shape[2] * hp.shape[3]);
  EMP_REQUIRE(cpad >= C, "%s: bad depthwise padding", name.c_str());
  std::vector<float> pk((size_t)KK * cpad, 0.f);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < KK; ++t) pk[(size_t)t * cpad + c] = hp.w[(size_t)c * KK + t];
  return upload_f32(n, name + ".dw32", pk);
}

// ConvTranspose2d(k=2,s=2) weight (Cin,Cout,2,2) -> 1x1 conv with 4*Cout outputs, fp32 (pack_convT)
int pack32_convT(emp_pdl* n, const std::string& name) {
  const HostParam& hp = n->params.at(name);
  EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[2] == 2 && hp.shape[3] == 2, "%s: expected (Cin,Cout,2,2)", name.c_str());
  const int cin = (int)hp.shape[0], co = (int)hp.shape[1];
  HostParam t;
  t.shape = {4 * co, cin, 1, 1};
  t.w.resize((size_t)4 * co * cin);
  t.b.resize((size_t)4 * co);
  for (int q = 0; q < 4; ++q)
    for (int o = 0; o < co; ++o) {
      t.b[(size_t)q * co + o] = hp.b[o];
      for (int i = 0; i < cin; ++i) t.w[((size_t)q * co + o) * cin + i] = hp.w[(((size_t)i * co + o) * 2 + (q >> 1)) * 2 + (q & 1)];
    }
  return pack32(n, name, 0, &t);
}

#define RC32(x)          \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

int finalize32(emp_pdl* n) {
  const emp_pdl_config& c = n->cfg;
  if (c.encoder == 1) {
    RC32(upload_regnet_stem(n));
    for (int si = 1; si <= 4; ++si)
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        const int w = c.rn_widths[si - 1], g = c.rn_groups[si - 1];
        RC32(pack32(n, p + ".bottleneck.a.0"));
        RC32(pack32(n, p + ".bottleneck.b.0"));      // (w, w / g, 3, 3): rows [o][tap][(w / g) padded to 16], group-major in o
        const emp_pdl::W32& wb = n->w32.at(p + ".bottleneck.b.0");
        EMP_REQUIRE(wb.cout == w && wb.cin * g == w && wb.kh == 3 && wb.kw == 3, "%s.bottleneck.b.0 must be (%d,%d,3,3)", p.c_str(), w,
                    w / g);
        if (c.rn_se) {
          RC32(pack32(n, p + ".bottleneck.se.se.0"));
          RC32(pack32(n, p + ".bottleneck.se.se.2"));
        }
        RC32(pack32(n, p + ".bottleneck.c.0"));
        if (regnet_has_shortcut(c, si, b)) RC32(pack32(n, p + ".downsample.conv.0"));
      }
  } else {
    for (int li = 1; li <= 4; ++li)
      for (int b = 0; b < kLayers[li - 1]; ++b) {
        const std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
        RC32(pack32(n, p + ".conv1"));
        RC32(pack32(n, p + ".conv2"));
        RC32(pack32(n, p + ".conv3"));
        if (b == 0) RC32(pack32(n, p + ".downsample.0"));
        if (b == 0 && n->precision == 2) RC32(pack32_conv3_ds(n, p));
      }
  }
  if (c.arch == 1) {
    RC32(pack32(n, "p2_resample.conv.0"));
    for (const auto& nm : n->param_names) {
      const bool fpn = nm.find("_fpn.") != std::string::npos, dec = nm.find("_decoder.") != std::string::npos;
      if (!fpn && !dec) continue;
      if (nm.size() > 8 && nm.compare(nm.size() - 8, 8, ".weights") == 0) continue;      // fusew: host side (finalize)
      if (nm.find(".sepconv.0") != std::string::npos) RC32(pack32_dw(n, nm, (int)n->params[nm].shape[0]));
      else if (nm.find(".upsamplings.") != std::string::npos) RC32(pack32_convT(n, nm));
      else RC32(pack32(n, nm));
    }
  } else {
    const char* decs[2] = {"semantic_decoder", "instance_decoder"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      const std::string p = decs[d];
      for (int i = 0; i <= 3; ++i) RC32(pack32(n, p + ".aspp.convs." + std::to_string(i) + ".0"));
      {   // projection: first 4A input channels -> conv; the pooled branch's A channels -> per-image bias (projpool.w)
        const HostParam& hp = n->params[p + ".aspp.project.0"];
        const int A = n->aspp_ch;
        HostParam head;
        head.shape = {A, 4 * A, 1, 1};
        head.w.resize((size_t)A * 4 * A);
        head.b = hp.b;
        for (int o = 0; o < A; ++o)
          for (int i = 0; i < 4 * A; ++i) head.w[(size_t)o * 4 * A + i] = hp.w[(size_t)o * 5 * A + i];
        RC32(pack32(n, p + ".aspp.project.0", 0, &head));
      }
      int xch = n->aspp_ch;
      for (int i = 0; i < c.n_stages; ++i) {
        const int lp = d == 0 ? c.low_level_proj_sem[i] : c.low_level_proj_ins[i];
        RC32(pack32(n, p + ".project." + std::to_string(i) + ".0"));
        const int cpad = round_up(xch + lp, n->precision == 2 ? 32 : 16);      // (the fused block walks the channels in chunks of 32)
        RC32(pack32_dw(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.0", cpad));
        RC32(pack32(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1", cpad));
        xch = n->dec_ch;
      }
    }
  }
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  for (int k = 0; k < 3; ++k) {
    const std::string p = heads[k];
    RC32(pack32_dw(n, p + ".head.0.0.sepconv.0", n->dec_ch));
    RC32(pack32(n, p + ".head.0.0.sepconv.1"));
  }
  const int ldp = round_up(n->dec_ch + n->ncls, 16);
  for (int k = 0; k < c.num_fc; ++k) RC32(pack32(n, "semantic_pr.point_head.fc_layers." + std::to_string(k) + ".0", ldp));
  {
    const HostParam& hp = n->params["semantic_pr.point_head.predictor"];
    const int K = (int)hp.shape[1];
    std::vector<float> w((size_t)n->ncls * ldp, 0.f);
    for (int o = 0; o < n->ncls; ++o)
      for (int i = 0; i < K; ++i) w[(size_t)o * ldp + i] = hp.w[(size_t)o * K + i];
    RC32(upload_f32(n, "pr.predictor.w32", w));
  }
  if (n->precision == 2 && n->x3_ksplit && !n->x3_kpart) {
    void* d = nullptr;
    EMP_CHECK_HIP(hipMalloc(&d, (size_t)X3_KPART_BYTES));
    n->owned.push_back(d);
    n->x3_kpart = (float*)d;
  }
  if (n->precision == 2 && n->x3_fuse_sep) {
    // fp16x3 mode, round 6: the separable blocks' weights in the fused kernel's orders (sepconv_x3.hip)
    for (auto& kv : n->w32) {
      const std::string& nm = kv.first;
      if (nm.size() < 10 || nm.compare(nm.size() - 10, 10, ".sepconv.1") != 0) continue;
      const std::string base = nm.substr(0, nm.size() - 2);      // "... .sepconv"
      auto dwi = n->f32w.find(base + ".0.dw32");
      auto hpi = n->params.find(base + ".0");
      if (dwi == n->f32w.end() || hpi == n->params.end() || hpi->second.shape.size() != 4) continue;
      const emp_pdl::W32& w = kv.second;
      const int ks = (int)hpi->second.shape[2], C = w.cin16;
      if (w.kh != 1 || w.kw != 1 || !sepconv_x3_supported(C, w.cout, 0, ks)) continue;
      emp_pdl::SepX3 sx;
      sx.C = C; sx.Cout = w.cout; sx.ks = ks;
      void* d = nullptr;
      EMP_CHECK_HIP(hipMalloc(&d, (size_t)ks * ks * C * sizeof(float)));
      n->owned.push_back(d);
      sx.dw = (float*)d;
      RC32(launch_sepx3_pack_dw(dwi->second, ks, C, C, sx.dw, nullptr));
      EMP_CHECK_HIP(hipMalloc(&d, (size_t)sepx3_pw_halfs(C, w.cout) * sizeof(half_t)));
      n->owned.push_back(d);
      sx.pw = (half_t*)d;
      RC32(launch_sepx3_pack_pw(w.w, C, C, w.cout, sx.pw, nullptr));
      n->sepx3[base] = sx;
    }
    EMP_CHECK_HIP(hipStreamSynchronize(nullptr));
  }
  if (n->precision == 2 && c.arch == 0 && c.ins_decoder && n->x3_merge_proj) {
    // fp16x3 mode, round 6: the two decoders' low-level projections of stage i read the same encoder map: weights stacked along
    // Cout, one launch with two destinations (conv16x3.hip store4) -- the widest map of the network is read once instead of twice
    for (int i = 0; i < c.n_stages; ++i) {
      const std::string a = "semantic_decoder.project." + std::to_string(i) + ".0", b = "instance_decoder.project." + std::to_string(i) + ".0";
      auto ia = n->w32.find(a), ib = n->w32.find(b);
      if (ia == n->w32.end() || ib == n->w32.end()) continue;
      const emp_pdl::W32 &wa = ia->second, &wb = ib->second;
      if (wa.cin16 != wb.cin16 || wa.kh != 1 || wb.kh != 1 || wa.kw != 1 || wb.kw != 1 || wa.cout % 4 || wb.cout % 4 || wa.cin16 >= 1024) continue;
      emp_pdl::W32 m = wa;
      m.cout = wa.cout + wb.cout; m.wp = nullptr; m.wimg = nullptr; m.wimgp = nullptr;
      const size_t na = (size_t)wa.cout * wa.cin16, nb = (size_t)wb.cout * wb.cin16;
      void* d = nullptr;
      EMP_CHECK_HIP(hipMalloc(&d, (na + nb) * sizeof(float)));
      n->owned.push_back(d);
      m.w = (float*)d;
      EMP_CHECK_HIP(hipMemcpy(m.w, wa.w, na * sizeof(float), hipMemcpyDeviceToDevice));
      EMP_CHECK_HIP(hipMemcpy(m.w + na, wb.w, nb * sizeof(float), hipMemcpyDeviceToDevice));
      EMP_CHECK_HIP(hipMalloc(&d, (size_t)m.cout * sizeof(float)));
      n->owned.push_back(d);
      m.b = (float*)d;
      EMP_CHECK_HIP(hipMemcpy(m.b, wa.b, (size_t)wa.cout * sizeof(float), hipMemcpyDeviceToDevice));
      EMP_CHECK_HIP(hipMemcpy(m.b + wa.cout, wb.b, (size_t)wb.cout * sizeof(float), hipMemcpyDeviceToDevice));
      n->w32["decoders.project." + std::to_string(i) + ".0"] = m;
    }
  }
  if (n->precision == 2) {
    // fp16x3 mode: every convolution weight once more as fp16 pairs (hi | lo << 16, the fp32 blob's layout), so that the
    // kernel's weight staging is a lane permutation instead of five vector operations per element (conv16x3.hip)
    for (auto& kv : n->w32) {
      emp_pdl::W32& w = kv.second;
      const int64_t cnt = (int64_t)w.cout * (w.kh * w.kw * w.cin16 + w.cin2_16);
      void* d = nullptr;
      EMP_CHECK_HIP(hipMalloc(&d, (size_t)(cnt > 0 ? cnt : 4) * sizeof(uint32_t)));
      n->owned.push_back(d);
      w.wp = (uint32_t*)d;
      RC32(launch_split_pairs(w.w, w.wp, cnt, nullptr));
      // long-K layers (the split-role kernel's): the weights once more as that kernel's LDS image, fetched by LDS-DMA
      const int K = w.kh * w.kw * w.cin16;
      const int64_t ih = (w.cin2_16 == 0 && K >= 1024) ? x3_weight_image_halfs(w.cout, K, w.cin16) : 0;
      if (ih > 0) {
        EMP_CHECK_HIP(hipMalloc(&d, (size_t)ih * sizeof(half_t)));
        n->owned.push_back(d);
        w.wimg = (half_t*)d;
        RC32(launch_x3_weight_image(w.w, w.wimg, w.cout, K, nullptr));
      }
      // round 6: the plane region's layers (ResNet layer3 / layer4 and the ASPP convolutions: Cout % 256 == 0, Cin % 32 == 0)
      // once more as conv16x3p_kernel's packed hi / lo image
      const bool region = kv.first.find("encoder.layer3.") == 0 || kv.first.find("encoder.layer4.") == 0 || kv.first.find(".aspp.") != std::string::npos;
      const int64_t ip = (n->x3_planes && n->cfg.encoder == 0 && region && w.cin2_16 == 0 && w.cin16 % 32 == 0 && K >= 128) ? x3p_image_halfs(w.cout, K) : 0;
      if (ip > 0) {
        EMP_CHECK_HIP(hipMalloc(&d, (size_t)ip * sizeof(half_t)));
        n->owned.push_back(d);
        w.wimgp = (half_t*)d;
        w.x3p_kg = x3p_kgroup(w.kh * w.kw, w.cin16);
        RC32(launch_x3p_pack(w.w, w.wimgp, w.cout, K, nullptr, w.kh * w.kw, w.cin16, w.x3p_kg));
      }
    }
    if (n->x3_planes && c.encoder == 0 && c.arch == 0 && c.ins_decoder && n->x3_merge_aspp) {
      // the two decoders' ASPP branch i reads the same p5: [semantic ; instance] weights stacked along Cout, ONE launch with two
      // destinations (twice the workgroups per launch: a batch of 8 tiles of 1024^2 fills the chip with the 256-channel branches)
      for (int i = 0; i <= 3; ++i) {
        const std::string a = "semantic_decoder.aspp.convs." + std::to_string(i) + ".0", b = "instance_decoder.aspp.convs." + std::to_string(i) + ".0";
        auto ia = n->w32.find(a), ib = n->w32.find(b);
        if (ia == n->w32.end() || ib == n->w32.end()) continue;
        const emp_pdl::W32 &wa = ia->second, &wb = ib->second;
        if (!wa.wimgp || !wb.wimgp || wa.cout != wb.cout || wa.cin16 != wb.cin16 || wa.kh != wb.kh || wa.kw != wb.kw) continue;
        emp_pdl::W32 m = wa;
        m.cout = wa.cout + wb.cout; m.wp = nullptr; m.wimg = nullptr; m.wimgp = nullptr;
        const size_t K = (size_t)wa.kh * wa.kw * wa.cin16, na = (size_t)wa.cout * K, nb = (size_t)wb.cout * K;
        void* d = nullptr;
        EMP_CHECK_HIP(hipMalloc(&d, (na + nb) * sizeof(float)));
        n->owned.push_back(d);
        m.w = (float*)d;
        EMP_CHECK_HIP(hipMemcpy(m.w, wa.w, na * sizeof(float), hipMemcpyDeviceToDevice));
        EMP_CHECK_HIP(hipMemcpy(m.w + na, wb.w, nb * sizeof(float), hipMemcpyDeviceToDevice));
        EMP_CHECK_HIP(hipMalloc(&d, (size_t)m.cout * sizeof(float)));
        n->owned.push_back(d);
        m.b = (float*)d;
        EMP_CHECK_HIP(hipMemcpy(m.b, wa.b, (size_t)wa.cout * sizeof(float), hipMemcpyDeviceToDevice));
        EMP_CHECK_HIP(hipMemcpy(m.b + wa.cout, wb.b, (size_t)wb.cout * sizeof(float), hipMemcpyDeviceToDevice));
        const int64_t ip = x3p_image_halfs(m.cout, (int)K);
        if (ip <= 0) continue;
        EMP_CHECK_HIP(hipMalloc(&d, (size_t)ip * sizeof(half_t)));
        n->owned.push_back(d);
        m.wimgp = (half_t*)d;
        m.x3p_kg = x3p_kgroup(m.kh * m.kw, m.cin16);
        RC32(launch_x3p_pack(m.w, m.wimgp, m.cout, (int)K, nullptr, m.kh * m.kw, m.cin16, m.x3p_kg));
        n->w32["decoders.aspp.convs." + std::to_string(i) + ".0"] = m;
      }
    }
    EMP_CHECK_HIP(hipStreamSynchronize(nullptr));
    if (n->x3_planes && c.encoder == 0) {
      bool ok = true;
      for (auto& kv : n->w32) {
        const bool region = kv.first.find("encoder.layer3.") == 0 || kv.first.find("encoder.layer4.") == 0 || kv.first.find(".aspp.") != std::string::npos;
        const bool boundary = kv.first == "encoder.layer3.0.conv1" || kv.first == "encoder.layer3.0.downsample.0" || kv.first.find("conv3+ds") != std::string::npos;
        if (region && !boundary && !kv.second.wimgp) ok = false;      // (the boundary layers read the fp32 layer2 map: round 5's kernels)
      }
      for (int i = 0; i < (c.arch == 0 ? c.n_stages : 0); ++i) ok = ok && c.low_level_stages[i] <= 2;      // a low-level skip out of the region would need a conversion
      n->x3_planes_ready = ok;
    }
  }
  return EMP_OK;
}

// out[:, :, :, out_coff : out_coff + Cout) = act(conv(in[:, :, :, in_coff : in_coff + Cin16)) + bias (+ bias_n) (+ res))
int c32(emp_pdl* n, const std::string& wname, const T32& in, int in_coff, const T32& out, int out_coff, int stride, int pad,
        int dil, int act, const T32* res, const float* bias_n, hipStream_t s, int ps_cout = 0, int groups = 1,
        const float* head_w = nullptr, float* head_part = nullptr, int head_c = 0, const T32* in2 = nullptr, int stride2 = 1,
        const T32* out2 = nullptr, int out2_coff = 0, int split2 = 0) {
  const emp_pdl::W32& w = n->w32.at(wname);
  Conv32 p{};
  if (out2) {      // couts [split2, Cout) -> out2 (the merged ASPP branches on conv16x3p; the merged low-level projections on conv16x3's vector epilogue)
    EMP_REQUIRE(n->precision == 2 && out2->fmt == out.fmt && (!out2->fmt || out2_coff % 32 == 0) && (in.fmt || !out.fmt), "%s: bad second destination", wname.c_str());
    p.out2 = out2->p + out2_coff; p.out2_ld = out2->ld; p.split2 = split2;
  }
  if (groups > 1) {      // grouped 3x3 of a RegNet block: w.cin is the group width, w.cout all output channels
    EMP_REQUIRE(w.cout % groups == 0 && in_coff == 0 && out_coff == 0 && !res && !bias_n && !ps_cout, "%s (fp32): bad grouped call",
                wname.c_str());
    p.groups = groups;
    p.cin_g = w.cin;
  }
  // an hl32 map (T32::fmt, the fp16x3 mode's plane region): halfs, channel c of a row at (c / 32) * 64 + c % 32 -- a channel
  // slice starts at a multiple of 32 channels = coff * 2 halfs = coff floats into the row
  EMP_REQUIRE((!in.fmt || in_coff % 32 == 0) && (!out.fmt || out_coff % 32 == 0), "%s: hl32 channel slices start at multiples of 32", wname.c_str());
  EMP_REQUIRE(!in.fmt || (w.wimgp && groups <= 1 && !in2 && !head_w && !ps_cout), "%s: an hl32 input needs the packed image of a plain convolution", wname.c_str());
  p.in = in.p + in_coff; p.in_ld = in.ld; p.in_fmt = in.fmt;
  p.wimgp = in.fmt ? w.wimgp : nullptr;
  p.x3p_kg = w.x3p_kg;
  p.w = w.w; p.bias = w.b; p.bias_n = bias_n;
  p.res = res ? res->p : nullptr; p.res_ld = res ? res->ld : 0; p.res_fmt = res ? res->fmt : 0;
  p.out = out.p + out_coff; p.out_ld = out.ld; p.out_fmt = out.fmt;
  p.N = in.N; p.H = in.H; p.W = in.W; p.Cin = w.cin16; p.Cout = w.cout / groups; p.KH = w.kh; p.KW = w.kw;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (in.H + 2 * pad - dil * (w.kh - 1) - 1) / stride + 1;
  p.Wo = (in.W + 2 * pad - dil * (w.kw - 1) - 1) / stride + 1;
  p.act = act; p.ps_cout = ps_cout;
  p.x3 = n->precision == 2;
  p.wpair = p.x3 ? w.wp : nullptr;
  p.wimg = (p.x3 && groups <= 1) ? w.wimg : nullptr;
  p.head_w = head_w; p.head_part = head_part; p.head_c = head_c;      // (fp16x3 only: the map `out` is then not written)
  if (p.x3 && n->x3_ksplit && n->x3_kpart && !head_w && !in2 && (in.fmt || !out2)) { p.kpart = n->x3_kpart; p.kpart_bytes = X3_KPART_BYTES; }
  if (in2) {      // K-concatenated second source (fp16x3 only: weights packed by pack32_conv3_ds)
    EMP_REQUIRE(p.x3 && w.cin2_16 > 0 && w.cin2_16 <= in2->ld && in2->N == in.N, "%s: second source mismatch", wname.c_str());
    p.in2 = in2->p; p.in2_ld = in2->ld; p.Cin2 = w.cin2_16; p.H2 = in2->H; p.W2 = in2->W; p.stride2 = stride2;
    n->flops += 2.0 * (double)p.N * p.Ho * p.Wo * w.cout * (double)w.cin2;
  } else {
    EMP_REQUIRE(w.cin2_16 == 0, "%s: packed for two sources", wname.c_str());
  }
  EMP_REQUIRE(!head_w || p.x3, "%s: the fused head exists in the fp16x3 mode only", wname.c_str());
  const int up = ps_cout ? 2 : 1;
  EMP_REQUIRE(p.Ho * up == out.H && p.Wo * up == out.W && in.N == out.N, "%s (fp32): output shape mismatch", wname.c_str());
  EMP_REQUIRE(in_coff + (groups - 1) * w.cin + w.cin16 <= in.ld && out_coff + (ps_cout ? ps_cout : (out2 ? split2 : w.cout)) <= out.ld,
              "%s (fp32): channel slice out of range (needs %d of a row of %d)", wname.c_str(), in_coff + (groups - 1) * w.cin + w.cin16, in.ld);
  n->flops += 2.0 * (double)p.N * p.Ho * p.Wo * w.cout * (double)(w.cin * w.kh * w.kw);
  if (n->profile && p.x3 && p.in_fmt) {     // emp_pdl_profile: HIP events around the plane region's launches (the fp16x3 mode's dominant kernel)
    if (n->prof_used == n->prof_events.size()) {
      hipEvent_t a, b;
      EMP_CHECK_HIP(hipEventCreate(&a));
      EMP_CHECK_HIP(hipEventCreate(&b));
      n->prof_events.emplace_back(a, b);
    }
    auto& ev = n->prof_events[n->prof_used++];
    EMP_CHECK_HIP(hipEventRecord(ev.first, s));
    const int rc = launch_conv32(p, s);
    EMP_CHECK_HIP(hipEventRecord(ev.second, s));
    n->prof_flops += 3.0 * 2.0 * (double)p.N * p.Ho * p.Wo * w.cout * (double)(w.cin16 * w.kh * w.kw);      // fp16 MFMA flops: three products per MAC
    return rc;
  }
  return launch_conv32(p, s);
}

int run32(emp_pdl* n, const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw, int RS, int interp,
          float* o_sem, float* o_ctr, float* o_off, hipStream_t s) {
  const emp_pdl_config& c = n->cfg;
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "forward: H=%d W=%d must be positive multiples of 16", H, W);
  EMP_REQUIRE(RS >= 1 && RS <= 6, "render_steps=%d out of range", RS);
  EMP_REQUIRE(c.arch == 0 || (H % 128 == 0 && W % 128 == 0), "BiFPN forward: H=%d W=%d must be multiples of 128", H, W);
  n->flops = 0.0;
  std::map<std::string, T32> T;
  auto mk = [&](const std::string& k, int H_, int W_, int C_, int fmt = 0) -> int { T32 t; int rc = t32(n, k, N, H_, W_, C_, &t); t.fmt = fmt; T[k] = t; return rc; };
  // fp16x3 mode, round 6: layer3 / layer4 / ASPP maps as hl32 planes on conv16x3p_kernel (emp_pdl members x3_planes*)
  const bool hlr = n->precision == 2 && n->x3_planes_ready && c.encoder == 0 &&
                      ((int64_t)N * (H / 16) * (W / 16)) / 256 >= n->x3_planes_min_tiles * (c.stage4_stride == 32 ? 2 : 1);
  // (an encoder at output stride 32 -- the BiFPN networks -- has a quarter of those tiles in layer4: BiFPN-PR with the region on / off
  //  492 / 513 tiles/s at batch 8, 594 / 581 at 16, 651 / 631 at 32: its threshold is twice the stride-16 one)
  auto A = [&](const std::string& k) -> T32& { return T.at(k); };
  auto dw = [&](const T32& in, const std::string& wname, int K, const T32& out) -> int {
    n->flops += 2.0 * K * K * (double)N * in.H * in.W * in.C;
    return launch_dwconv_f32(in.p, N, in.H, in.W, in.C, in.ld, n->f32w.at(wname + ".dw32"), K, out.p, out.ld, s);
  };
  // fp16x3 mode: the separable block `base` (".sepconv") of `in` as one launch, if the fused kernel takes it; *done says so
  auto sep_x3 = [&](const std::string& base, const T32& in, int act, float* out, int out_ld, const float* hw, const float* hb, int hcn,
                    float* hout, bool* done) -> int {
    *done = false;
    auto it = n->sepx3.find(base);
    if (n->precision != 2 || it == n->sepx3.end()) return EMP_OK;
    const emp_pdl::SepX3& sx = it->second;
    const int64_t tiles = (int64_t)N * ((in.H + 7) / 8) * ((in.W + 15) / 16);
    if (in.fmt || in.ld < sx.C || tiles < n->x3_sep_min_tiles || !sepconv_x3_supported(sx.C, sx.Cout, hcn, sx.ks)) return EMP_OK;
    *done = true;
    n->flops += 2.0 * sx.ks * sx.ks * (double)N * in.H * in.W * in.C + 2.0 * (double)N * in.H * in.W * sx.Cout * (double)n->w32.at(base + ".1").cin;
    return launch_sepconv_x3(in.p, N, in.H, in.W, sx.C, in.ld, sx.dw, sx.pw, n->w32.at(base + ".1").b, sx.Cout, act, out, out_ld, hw, hb, hcn,
                             hout, (int64_t)in.H * in.W, s, sx.ks);
  };
  // ---- encoder ----
  std::string x, pyr[5];
  if (c.encoder == 1) {
    // RegNet (regnet.py:160-166).  Its widths are multiples of 8, not of 16: every map gets a row of round_up(C, 16) + 16
    // floats whose tail stays zero (buf32 clears a buffer when it allocates it and no kernel writes beyond C), so that a
    // consumer reading its input channels padded to 16 -- from a group's first channel, in the grouped 3x3 -- stays
    // inside the row and meets zeros (or the next group's finite values) under zero weights.
    auto mkp = [&](const std::string& k, int H_, int W_, int C_) -> int {
      T32 t;
      t.N = N; t.H = H_; t.W = W_; t.C = C_; t.ld = round_up(C_, 16) + 16;
      const size_t had = n->pool32.count(k) ? n->pool32[k].second : 0;
      const int rc = buf32(n, k, (size_t)N * H_ * W_ * t.ld, &t.p);
      if (rc) return rc;
      // a buffer kept from a forward of another shape holds that forward's values where this one's row tails are
      const std::array<int, 4> geo = {N, H_, W_, t.ld};
      auto gi = n->geom32.find(k);
      if (gi == n->geom32.end() || gi->second != geo) {
        if (had >= (size_t)N * H_ * W_ * t.ld && gi != n->geom32.end())
          EMP_CHECK_HIP(hipMemsetAsync(t.p, 0, (size_t)N * H_ * W_ * t.ld * sizeof(float), s));
        n->geom32[k] = geo;
      }
      T[k] = t;
      return EMP_OK;
    };
    RC32(mkp("stem", H / 2, W / 2, c.rn_stem));
    RC32(launch_stem3x3s2_f32(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("rn.stem.w"), n->f32w.at("rn.stem.b"), c.rn_stem,
                              A("stem").p, A("stem").ld, s));
    n->flops += 2.0 * N * (H / 2) * (W / 2) * (double)c.rn_stem * 9.0;
    x = "stem";
    pyr[0] = "stem";
    for (int si = 1; si <= 4; ++si) {
      const int w = c.rn_widths[si - 1], g = c.rn_groups[si - 1];
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        const int sb = b == 1 ? c.rn_strides[si - 1] : 1;
        const T32 xin = A(x);
        const int ho = (xin.H - 1) / sb + 1, wo = (xin.W - 1) / sb + 1;
        RC32(mkp(p + ".a", xin.H, xin.W, w));
        RC32(c32(n, p + ".bottleneck.a.0", xin, 0, A(p + ".a"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
        RC32(mkp(p + ".b", ho, wo, w));
        RC32(c32(n, p + ".bottleneck.b.0", A(p + ".a"), 0, A(p + ".b"), 0, sb, 1, 1, 1, nullptr, nullptr, s, 0, g));
        if (c.rn_se) {      // per-pixel gate: x * sigmoid(W2 relu(W1 x)) (blocks.py:35-50: the pool is 1 x 1)
          RC32(mkp(p + ".se1", ho, wo, w / 4));
          RC32(c32(n, p + ".bottleneck.se.se.0", A(p + ".b"), 0, A(p + ".se1"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
          RC32(mkp(p + ".se2", ho, wo, w));
          RC32(c32(n, p + ".bottleneck.se.se.2", A(p + ".se1"), 0, A(p + ".se2"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
          RC32(launch_gate_mul_f32(A(p + ".b").p, A(p + ".b").ld, A(p + ".se2").p, A(p + ".se2").ld, (int64_t)N * ho * wo, w, s));
        }
        const T32* idn = &A(x);
        if (regnet_has_shortcut(c, si, b)) {
          RC32(mkp(p + ".ds", ho, wo, w));
          RC32(c32(n, p + ".downsample.conv.0", xin, 0, A(p + ".ds"), 0, sb, 0, 1, 0, nullptr, nullptr, s));
          idn = &A(p + ".ds");
        }
        RC32(mkp(p, ho, wo, w));
        RC32(c32(n, p + ".bottleneck.c.0", A(p + ".b"), 0, A(p), 0, 1, 0, 1, 1, idn, nullptr, s));
        x = p;
      }
      pyr[si] = x;
    }
  } else {
    n->flops += 2.0 * N * (H / 2) * (W / 2) * 64.0 * 49.0;
    RC32(mk("p1", H / 4, W / 4, 64));
    if (n->precision == 2 && n->x3_fuse_stem) {
      // fp16x3 mode, round 6: conv1 + bn1 + relu + maxpool as one launch on the matrix pipe (stem.hip stem_pool32_kernel: the three-MFMA
      // split product of every other convolution of the mode, fp32 tile and output); the half-resolution map is never written
      RC32(launch_stem_pool_f32(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("stem.w"), n->f32w.at("stem.b"), A("p1").p, s));
    } else {
      RC32(mk("stem", H / 2, W / 2, 64));
      RC32(launch_stem7x7_f32(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("stem.w"), n->f32w.at("stem.b"), A("stem").p, s));
      RC32(launch_maxpool3x3s2_f32(A("stem").p, N, H / 2, W / 2, 64, A("p1").p, s));
    }
    x = "p1";
    pyr[0] = "p1";
    for (int li = 1; li <= 4; ++li) {
      int stride = li == 1 ? 1 : 2, dil = 1;
      if (li == 4 && c.stage4_stride == 16) { stride = 1; dil = 2; }
      for (int b = 0; b < kLayers[li - 1]; ++b) {
        const int sb = b == 0 ? stride : 1, planes = kPlanes[li - 1];
        const std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
        const T32 xin = A(x);
        const int ho = (xin.H - 1) / sb + 1, wo = (xin.W - 1) / sb + 1;
        // plane region (li >= 3): every map hl32 except where round 5's kernels still read it -- layer3.0's conv1 and
        // conv3 + shortcut take the fp32 layer2 map (and the fp32 c2) and WRITE hl32 (Conv32::out_fmt)
        const int pl = (hlr && li >= 3) ? 1 : 0;
        const bool fuse_ds = b == 0 && n->precision == 2 && n->x3_fuse_ds && !(pl && xin.fmt);
        RC32(mk(p + ".c1", xin.H, xin.W, planes, pl));
        RC32(c32(n, p + ".conv1", xin, 0, A(p + ".c1"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
        RC32(mk(p + ".c2", ho, wo, planes, (pl && !fuse_ds) ? 1 : 0));
        RC32(c32(n, p + ".conv2", A(p + ".c1"), 0, A(p + ".c2"), 0, sb, dil, dil, 1, nullptr, nullptr, s));
        const T32* idn = &A(x);
        RC32(mk(p, ho, wo, planes * 4, pl));
        if (fuse_ds) {
          // fp16x3 mode: relu(conv3(c2) + downsample(x)) as one convolution over the concatenated K (the shortcut map is
          // neither written nor read back)
          RC32(c32(n, p + ".conv3+ds", A(p + ".c2"), 0, A(p), 0, 1, 0, 1, 1, nullptr, nullptr, s, 0, 1, nullptr, nullptr, 0, &xin, sb));
          x = p;
          continue;
        }
        if (b == 0) {
          RC32(mk(p + ".ds", ho, wo, planes * 4, pl));
          RC32(c32(n, p + ".downsample.0", xin, 0, A(p + ".ds"), 0, sb, 0, 1, 0, nullptr, nullptr, s));
          idn = &A(p + ".ds");
        }
        RC32(c32(n, p + ".conv3", A(p + ".c2"), 0, A(p), 0, 1, 0, 1, 1, idn, nullptr, s));
        x = p;
      }
      pyr[li] = x;
    }
    if (hlr && c.arch == 1) {
      // the BiFPN reads P4 / P5 with round 5's kernels (128-channel nodes): fp32 copies of the two pyramid levels
      for (int li = 3; li <= 4; ++li) {
        const T32 src = A(pyr[li]);
        const std::string k = pyr[li] + ".f32";
        RC32(mk(k, src.H, src.W, src.C));
        RC32(launch_hl32_to_f32(reinterpret_cast<const half_t*>(src.p), A(k).p, (int64_t)N * src.H * src.W, src.C, src.ld, src.C, s));
        pyr[li] = k;
      }
    }
  }
  std::string dec_out[2];
  if (c.arch == 1) {
    // ---- BiFPN decoders (bifpn.py:185-236) ----
    const int F = c.fpn_dim;
    const T32 p2 = A(pyr[1]);
    RC32(mk("p2f", p2.H, p2.W, F));
    RC32(c32(n, "p2_resample.conv.0", p2, 0, A("p2f"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
    const char* dn[2] = {"semantic", "instance"};
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      const std::string fp = std::string(dn[d]) + "_fpn";
      const T32 p5 = A(pyr[4]);
      RC32(mk(fp + ".p6pre", p5.H, p5.W, F));
      RC32(c32(n, fp + ".p6_resample.conv.0", p5, 0, A(fp + ".p6pre"), 0, 1, 0, 1, 0, nullptr, nullptr, s));
      RC32(mk(fp + ".in.P6", p5.H / 2, p5.W / 2, F));
      RC32(launch_maxpool3x3s2_f32(A(fp + ".p6pre").p, N, p5.H, p5.W, F, A(fp + ".in.P6").p, s));
      RC32(mk(fp + ".in.P7", p5.H / 4, p5.W / 4, F));
      RC32(launch_maxpool3x3s2_f32(A(fp + ".in.P6").p, N, p5.H / 2, p5.W / 2, F, A(fp + ".in.P7").p, s));
      std::string feat[5] = {pyr[2], pyr[3], pyr[4], fp + ".in.P6", fp + ".in.P7"};
      for (int li = 0; li < c.fpn_layers; ++li) {
        const std::string L = fp + ".l" + std::to_string(li), pre = fp + ".bifpns." + std::to_string(li);
        auto node = [&](const std::string& dirpre, const std::string& q, const float* a, const float* b2, const float* c3,
                        float ca, float cb, float cc, int mode, int h_, int w_, const std::string& outname) -> int {
          RC32(mk(q + ".fz", h_, w_, F));
          RC32(launch_fuse_combine_f32(a, b2, c3, ca, cb, cc, mode, N, h_, w_, F, A(q + ".fz").p, s));
          {
            bool fused = false;
            RC32(mk(outname, h_, w_, F));
            RC32(sep_x3(dirpre + ".after_combines.0.0.sepconv", A(q + ".fz"), 2, A(outname).p, A(outname).ld, nullptr, nullptr, 0, nullptr, &fused));
            if (fused) return EMP_OK;
          }
          RC32(mk(q + ".dw", h_, w_, F));
          RC32(dw(A(q + ".fz"), dirpre + ".after_combines.0.0.sepconv.0", 3, A(q + ".dw")));
          RC32(mk(outname, h_, w_, F));
          return c32(n, dirpre + ".after_combines.0.0.sepconv.1", A(q + ".dw"), 0, A(outname), 0, 1, 0, 1, 2, nullptr, nullptr, s);
        };
        auto resampled = [&](const std::string& rk, const std::string& src, const std::string& dst, std::string* name) -> int {
          *name = src;
          if (!n->w32.count(rk)) return EMP_OK;
          const T32 in = A(src);
          RC32(mk(dst, in.H, in.W, F));
          *name = dst;
          return c32(n, rk, in, 0, A(dst), 0, 1, 0, 1, 0, nullptr, nullptr, s);
        };
        {
          const std::string dp = pre + ".top_down_fpn";
          const float* w = n->fusew.at(dp + ".weights").data();
          std::string td_prev = feat[4];
          for (int i = 0; i < 4; ++i) {
            const int lv = 3 - i;
            const std::string q = L + ".P" + std::to_string(3 + lv);
            std::string hi;
            RC32(resampled(dp + ".resamplings." + std::to_string(i) + ".conv.0", feat[lv], q + ".rtd", &hi));
            const float den = w[i] + w[i + 1] + 1e-4f;
            const T32 hi_t = A(hi);
            RC32(node(dp, q + ".tdn", A(td_prev).p, hi_t.p, nullptr, w[i] / den, w[i + 1] / den, 0.f, 0, hi_t.H, hi_t.W, q + ".td"));
            td_prev = q + ".td";
          }
        }
        {
          const std::string dp = pre + ".bottom_up_fpn";
          const float* w = n->fusew.at(dp + ".weights").data();
          std::string bu_prev = L + ".P3.td", newfeat[5];
          newfeat[0] = bu_prev;
          for (int i = 0; i < 4; ++i) {
            const int lv = i + 1;
            const std::string q = L + ".P" + std::to_string(3 + lv);
            std::string lo;
            RC32(resampled(dp + ".resamplings." + std::to_string(i) + ".conv.0", feat[lv], q + ".rbu", &lo));
            const T32 lo_t = A(lo);
            if (i < 3) {
              const float den = w[i] + w[i + 1] + w[i + 2] + 1e-4f;
              RC32(node(dp, q + ".bun", A(bu_prev).p, lo_t.p, A(q + ".td").p, w[i] / den, w[i + 1] / den, w[i + 2] / den, 1, lo_t.H,
                        lo_t.W, q + ".bu"));
            } else {
              const float den = w[i] + w[i + 1] + 1e-4f;
              RC32(node(dp, q + ".bun", A(bu_prev).p, lo_t.p, nullptr, w[i] / den, w[i + 1] / den, 0.f, 1, lo_t.H, lo_t.W, q + ".bu"));
            }
            bu_prev = q + ".bu";
            newfeat[lv] = bu_prev;
          }
          for (int lv = 0; lv < 5; ++lv) feat[lv] = newfeat[lv];
        }
      }
      const std::string dp = std::string(dn[d]) + "_decoder";
      const std::string skips[5] = {feat[3], feat[2], feat[1], feat[0], "p2f"};
      std::string xx = feat[4];
      for (int i = 0; i < 5; ++i) {
        const T32 in = A(xx);
        const std::string cn = dp + ".cat" + std::to_string(i);
        RC32(mk(cn, in.H * 2, in.W * 2, 2 * F));
        RC32(c32(n, dp + ".upsamplings." + std::to_string(i) + ".0", in, 0, A(cn), 0, 1, 0, 1, 1, nullptr, nullptr, s, F));
        const T32 sk = A(skips[i]);
        RC32(launch_bilinear_ac_f32_nhwc(sk.p, N, sk.H, sk.W, F, sk.ld, A(cn).p + F, sk.H, sk.W, 2 * F, s));      // same size: copy
        xx = cn;
      }
      const T32 cat = A(xx);
      RC32(mk(dp + ".out", cat.H, cat.W, F));
      bool fused = false;
      RC32(sep_x3(dp + ".fusion.0.sepconv", cat, 1, A(dp + ".out").p, A(dp + ".out").ld, nullptr, nullptr, 0, nullptr, &fused));
      if (!fused) {
        RC32(mk(dp + ".dw", cat.H, cat.W, 2 * F));
        RC32(dw(cat, dp + ".fusion.0.sepconv.0", 5, A(dp + ".dw")));
        RC32(c32(n, dp + ".fusion.0.sepconv.1", A(dp + ".dw"), 0, A(dp + ".out"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
      }
      dec_out[d] = dp + ".out";
    }
  } else {
    // ---- Panoptic-DeepLab decoders (decoders/panoptic_deeplab.py:68-80, aspp.py:96-102) ----
    const T32 p5 = A(pyr[4]);
    float* pooled;
    RC32(buf32(n, "pooled", (size_t)N * p5.C, &pooled));
    if (p5.fmt) RC32(launch_avgpool_hl32(reinterpret_cast<const half_t*>(p5.p), N, p5.H * p5.W, p5.C, p5.ld, pooled, s));
    else RC32(launch_avgpool_f32(p5.p, N, p5.H * p5.W, p5.C, p5.ld, pooled, s));
    const char* decs[2] = {"semantic_decoder", "instance_decoder"};
    // plane region: branch i of BOTH decoders as one launch (weights stacked along Cout at finalize, two destinations)
    const bool merged_aspp = p5.fmt && c.ins_decoder && n->w32.count("decoders.aspp.convs.0.0") && n->w32.count("decoders.aspp.convs.3.0");
    // below the plane region's threshold (small batches; round 6, late): the branches still run merged on the plane kernel -- K-split
    // (Conv32::kpart: 32 workgroups x 8 splits for ONE 1024^2 tile) -- from an hl32 copy of p5, into fp32 concat buffers
    bool small_aspp = false;
    if (!p5.fmt && n->precision == 2 && n->x3_planes_ready && n->x3_ksplit && n->x3_small_aspp && n->x3_kpart && c.ins_decoder &&
        n->w32.count("decoders.aspp.convs.0.0") && n->w32.count("decoders.aspp.convs.3.0") && p5.C % 32 == 0) {
      small_aspp = true;
      RC32(mk("p5.hl32", p5.H, p5.W, p5.C, 1));
      const T32& ph = A("p5.hl32");
      RC32(launch_hl32_from_f32(p5.p, reinterpret_cast<half_t*>(ph.p), (int64_t)N * p5.H * p5.W, p5.C, p5.ld, ph.ld, s));
      for (int d = 0; d < 2; ++d) RC32(mk(std::string(decs[d]) + ".aspp.cat", p5.H, p5.W, 4 * n->aspp_ch, 0));
      const T32 &c0 = A("semantic_decoder.aspp.cat"), &c1 = A("instance_decoder.aspp.cat");
      for (int i = 0; i <= 3; ++i) {
        const int r = i ? c.atrous_rates[i - 1] : 1;
        RC32(c32(n, "decoders.aspp.convs." + std::to_string(i) + ".0", ph, 0, c0, i * n->aspp_ch, 1, i ? r : 0, r, 1, nullptr, nullptr, s, 0, 1,
                 nullptr, nullptr, 0, nullptr, 1, &c1, i * n->aspp_ch, n->aspp_ch));
      }
    }
    if (merged_aspp) {
      for (int d = 0; d < 2; ++d) RC32(mk(std::string(decs[d]) + ".aspp.cat", p5.H, p5.W, 4 * n->aspp_ch, 1));
      const T32 &c0 = A("semantic_decoder.aspp.cat"), &c1 = A("instance_decoder.aspp.cat");
      for (int i = 0; i <= 3; ++i) {
        const int r = i ? c.atrous_rates[i - 1] : 1;
        RC32(c32(n, "decoders.aspp.convs." + std::to_string(i) + ".0", p5, 0, c0, i * n->aspp_ch, 1, i ? r : 0, r, 1, nullptr, nullptr, s, 0, 1,
                 nullptr, nullptr, 0, nullptr, 1, &c1, i * n->aspp_ch, n->aspp_ch));
      }
    }
    // the decoders' low-level projections, both decoders in one launch where finalize32 stacked their weights: the .cat buffers of
    // both decoders exist before the loop below fills their up-sampled halves
    bool proj_done[3] = {false, false, false};
    if (n->precision == 2 && c.ins_decoder) {
      int xch0 = n->aspp_ch;
      for (int i = 0; i < c.n_stages; ++i) {
        const std::string wn = "decoders.project." + std::to_string(i) + ".0";
        if (n->w32.count(wn)) {
          const T32 low = A(pyr[c.low_level_stages[i]]);
          const int cps = round_up(xch0 + c.low_level_proj_sem[i], 32), cpi = round_up(xch0 + c.low_level_proj_ins[i], 32);
          const std::string qs = std::string(decs[0]) + ".stage" + std::to_string(i), qi = std::string(decs[1]) + ".stage" + std::to_string(i);
          RC32(mk(qs + ".cat", low.H, low.W, cps));
          RC32(mk(qi + ".cat", low.H, low.W, cpi));
          if (!low.fmt && xch0 % 4 == 0) {
            RC32(c32(n, wn, low, 0, A(qs + ".cat"), xch0, 1, 0, 1, 1, nullptr, nullptr, s, 0, 1, nullptr, nullptr, 0, nullptr, 1, &A(qi + ".cat"), xch0,
                     c.low_level_proj_sem[i]));
            proj_done[i] = true;
          }
        }
        xch0 = n->dec_ch;
      }
    }
    for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
      const std::string p = decs[d];
      float *poolfeat, *bias_n;
      RC32(buf32(n, p + ".poolfeat", (size_t)N * n->aspp_ch, &poolfeat));
      RC32(buf32(n, p + ".bias_n", (size_t)N * n->aspp_ch, &bias_n));
      RC32(launch_gemv(pooled, N, p5.C, n->f32w.at(p + ".pool.w"), nullptr, n->aspp_ch, 1, poolfeat, s));
      RC32(launch_gemv(poolfeat, N, n->aspp_ch, n->f32w.at(p + ".projpool.w"), nullptr, n->aspp_ch, 0, bias_n, s));
      if (!merged_aspp && !small_aspp) {
        RC32(mk(p + ".aspp.cat", p5.H, p5.W, 4 * n->aspp_ch, p5.fmt));      // (the projection reads it as it was written; its output is fp32: the up-sampler's input)
        RC32(c32(n, p + ".aspp.convs.0.0", p5, 0, A(p + ".aspp.cat"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
        for (int i = 1; i <= 3; ++i) {
          const int r = c.atrous_rates[i - 1];
          RC32(c32(n, p + ".aspp.convs." + std::to_string(i) + ".0", p5, 0, A(p + ".aspp.cat"), i * n->aspp_ch, 1, r, r, 1, nullptr, nullptr, s));
        }
      }
      RC32(mk(p + ".aspp", p5.H, p5.W, n->aspp_ch));
      RC32(c32(n, p + ".aspp.project.0", A(p + ".aspp.cat"), 0, A(p + ".aspp"), 0, 1, 0, 1, 1, nullptr, bias_n, s));
      std::string xx = p + ".aspp";
      int xch = n->aspp_ch;
      for (int i = 0; i < c.n_stages; ++i) {
        const T32 low = A(pyr[c.low_level_stages[i]]);
        const int lp = d == 0 ? c.low_level_proj_sem[i] : c.low_level_proj_ins[i];
        const int cpad = round_up(xch + lp, n->precision == 2 ? 32 : 16);      // (finalize32 packed the block's weights for this width)
        const std::string q = p + ".stage" + std::to_string(i);
        RC32(mk(q + ".cat", low.H, low.W, cpad));
        const T32 xa = A(xx);
        RC32(launch_bilinear_ac_f32_nhwc(xa.p, N, xa.H, xa.W, xch, xa.ld, A(q + ".cat").p, low.H, low.W, cpad, s));
        if (!proj_done[i]) RC32(c32(n, p + ".project." + std::to_string(i) + ".0", low, 0, A(q + ".cat"), xch, 1, 0, 1, 1, nullptr, nullptr, s));
        RC32(mk(q + ".out", low.H, low.W, n->dec_ch));
        bool fused = false;
        RC32(sep_x3(p + ".fuse." + std::to_string(i) + ".0.sepconv", A(q + ".cat"), 1, A(q + ".out").p, A(q + ".out").ld, nullptr, nullptr, 0, nullptr, &fused));
        if (!fused) {
          RC32(mk(q + ".dw", low.H, low.W, cpad));
          RC32(dw(A(q + ".cat"), p + ".fuse." + std::to_string(i) + ".0.sepconv.0", 5, A(q + ".dw")));
          RC32(c32(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1", A(q + ".dw"), 0, A(q + ".out"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
        }
        xx = q + ".out";
        xch = n->dec_ch;
      }
      dec_out[d] = xx;
    }
  }
  if (!c.ins_decoder) dec_out[1] = dec_out[0];
  const T32 semx = A(dec_out[0]), insx = A(dec_out[1]);
  const int hq = semx.H, wq = semx.W;
  EMP_REQUIRE(hq * 4 == H && wq * 4 == W, "the decoder output must be at 1/4 resolution (got %dx%d)", hq, wq);
  // ---- heads (heads.py:12-19) ----
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  const int hc[3] = {n->ncls, 1, 2};
  float* head_out[3];
  for (int k = 0; k < 3; ++k) {
    const std::string p = heads[k];
    const T32& xin = k == 0 ? semx : insx;
    float* dst;
    RC32(buf32(n, p + ".out", (size_t)N * hc[k] * hq * wq, &dst));
    if (k == 1 && !interp) dst = o_ctr;
    if (k == 2 && !interp) dst = o_off;
    head_out[k] = dst;
    {   // fp16x3 mode, round 6: depthwise + pointwise + ReLU + the head's 1x1 in one launch (sepconv_x3.hip, head_c <= 2)
      bool fused = false;
      RC32(sep_x3(p + ".head.0.0.sepconv", xin, 1, nullptr, 0, n->f32w.at(p + ".head.1.w"), n->f32w.at(p + ".head.1.b"), hc[k], dst, &fused));
      if (fused) {
        n->flops += 2.0 * (double)N * hq * wq * n->dec_ch * hc[k];
        continue;
      }
    }
    RC32(mk(p + ".dw", hq, wq, n->dec_ch));
    RC32(dw(xin, p + ".head.0.0.sepconv.0", 5, A(p + ".dw")));
    if (n->precision == 2 && n->x3_fuse_head && hc[k] <= 4) {
      // fp16x3 mode: the head's 1x1 inside the pointwise conv's epilogue (conv16x3.hip HEAD): the dec_ch-wide map is
      // neither written nor read back; every cout tile leaves per-pixel partial sums, added in ascending order
      const int tiles = conv16x3_cout_tiles(n->dec_ch);
      float* part;
      RC32(buf32(n, p + ".part", (size_t)tiles * N * hq * wq * hc[k], &part));
      RC32(c32(n, p + ".head.0.0.sepconv.1", A(p + ".dw"), 0, A(p + ".dw"), 0, 1, 0, 1, 1, nullptr, nullptr, s, 0, 1,
               n->f32w.at(p + ".head.1.w"), part, hc[k]));
      RC32(launch_head_finish_f32(part, tiles, N, hq * wq, hc[k], n->f32w.at(p + ".head.1.b"), dst, s));
    } else {
      RC32(mk(p + ".pw", hq, wq, n->dec_ch));
      RC32(c32(n, p + ".head.0.0.sepconv.1", A(p + ".dw"), 0, A(p + ".pw"), 0, 1, 0, 1, 1, nullptr, nullptr, s));
      RC32(launch_head1x1_f32(A(p + ".pw").p, N, hq * wq, n->dec_ch, n->dec_ch, n->f32w.at(p + ".head.1.w"), n->f32w.at(p + ".head.1.b"),
                              hc[k], dst, (int64_t)hq * wq, nullptr, s));
    }
    n->flops += 2.0 * (double)N * hq * wq * n->dec_ch * hc[k];
  }
  if (interp) {
    RC32(launch_bilinear_ac_f32_nchw(head_out[1], N * 1, hq, wq, o_ctr, 4, s));
    RC32(launch_bilinear_ac_f32_nchw(head_out[2], N * 2, hq, wq, o_off, 4, s));
  }
  // ---- PointRend subdivision (point_rend.py:241-269, eval) ----
  const int P = c.subdivision_num_points;
  const int ldp = round_up(n->dec_ch + n->ncls, 16);
  const float* coarse = head_out[0];
  const float* cur = coarse;
  int hh = hq, ww = wq;
  int64_t plane_max = (int64_t)hq * wq;
  for (int st = 0; st < RS; ++st) plane_max *= 4;
  float* fkeys;
  RC32(buf32(n, "pr.keys", (size_t)N * plane_max, &fkeys));
  float* ftopk;
  const size_t topk_bytes = topk_work_bytes(N, plane_max);
  RC32(buf32(n, "pr.topk", (topk_bytes + 3) / 4, &ftopk));
  float* fidx;
  RC32(buf32(n, "pr.idx", (size_t)N * P, &fidx));
  T32 X[2];
  for (int j = 0; j < 2; ++j) {
    X[j].N = 1; X[j].H = 1; X[j].W = N * P; X[j].C = ldp; X[j].ld = ldp;
    RC32(buf32(n, j ? "pr.x1" : "pr.x0", (size_t)N * P * ldp, &X[j].p));
  }
  for (int st = 0; st < RS; ++st) {
    float* nxt = o_sem;
    if (st + 1 < RS) RC32(buf32(n, "pr.sem" + std::to_string(st), (size_t)N * n->ncls * hh * ww * 4, &nxt));
    RC32(launch_upsample2x_keys(cur, N, n->ncls, hh, ww, nxt, (uint32_t*)fkeys, s));
    hh *= 2; ww *= 2;
    const int64_t plane = (int64_t)hh * ww;
    const int k = (int)(plane < P ? plane : P);
    RC32(launch_topk_smallest((const uint32_t*)fkeys, N, plane, k, (char*)ftopk, topk_bytes, (int32_t*)fidx, s));
    RC32(launch_point_features_f32(semx.p, N, hq, wq, n->dec_ch, semx.ld, coarse, n->ncls, (const int32_t*)fidx, k, hh, ww, X[0].p,
                                   X[1].p, ldp, s));
    T32 xa[2] = {X[0], X[1]};
    xa[0].W = xa[1].W = N * k;
    int curx = 0;
    for (int f = 0; f < c.num_fc; ++f) {
      RC32(c32(n, "semantic_pr.point_head.fc_layers." + std::to_string(f) + ".0", xa[curx], 0, xa[curx ^ 1], 0, 1, 0, 1, 1, nullptr,
               nullptr, s));
      curx ^= 1;
    }
    RC32(launch_head1x1_f32(xa[curx].p, N, k, ldp, ldp, n->f32w.at("pr.predictor.w32"), n->f32w.at("pr.predictor.b"), n->ncls, nxt,
                            plane, (const int32_t*)fidx, s));
    n->flops += 2.0 * (double)N * k * ldp * n->ncls;
    cur = nxt;
  }
  return EMP_OK;
}

}  // namespace

extern "C" {

int emp_pdl_create(const emp_pdl_config* cfg, emp_pdl_t** out) {
  EMP_REQUIRE(cfg && out, "pdl_create: null argument");
  EMP_REQUIRE(cfg->num_classes >= 1 && cfg->num_classes <= 8, "num_classes=%d unsupported", cfg->num_classes);
  EMP_REQUIRE(cfg->stage4_stride == 16 || cfg->stage4_stride == 32, "stage4_stride must be 16 or 32");
  EMP_REQUIRE(cfg->arch == 0 || cfg->arch == 1, "arch must be 0 (PanopticDeepLabPR) or 1 (PanopticBiFPNPR)");
  if (cfg->arch == 1) {
    EMP_REQUIRE(cfg->fpn_dim > 0 && cfg->fpn_dim % 64 == 0 && cfg->fpn_layers >= 1 && cfg->stage4_stride == 32,
                "BiFPN: fpn_dim must be a multiple of 64, fpn_layers >= 1, stage4_stride 32");
  } else {
  EMP_REQUIRE(cfg->n_stages >= 1 && cfg->n_stages <= 3, "n_stages=%d unsupported", cfg->n_stages);
  EMP_REQUIRE(cfg->decoder_channels % 64 == 0 && cfg->decoder_channels > 0, "decoder_channels must be a multiple of 64");
  }
  EMP_REQUIRE(cfg->num_fc >= 1 && cfg->subdivision_num_points > 0, "bad PointRend configuration");
  for (int i = 0; i < (cfg->arch == 0 ? cfg->n_stages : 0); ++i) {
    EMP_REQUIRE(cfg->low_level_stages[i] >= 1 && cfg->low_level_stages[i] <= 3, "low_level_stages[%d] out of range", i);
    EMP_REQUIRE(cfg->low_level_proj_sem[i] % 8 == 0 && (!cfg->ins_decoder || cfg->low_level_proj_ins[i] % 8 == 0),
                "projected low-level channels must be multiples of 8");
  }
  EMP_REQUIRE(cfg->encoder == 0 || cfg->encoder == 1, "encoder must be 0 (ResNet50) or 1 (RegNet)");
  if (cfg->encoder == 1) {
    EMP_REQUIRE(cfg->rn_stem > 0 && cfg->rn_stem % 4 == 0, "RegNet: stem width %d must be a positive multiple of 4", cfg->rn_stem);
    for (int i = 0; i < 4; ++i) {
      const int w = cfg->rn_widths[i], g = cfg->rn_groups[i];
      EMP_REQUIRE(w > 0 && g > 0 && w % g == 0 && (w / g) % 4 == 0 && w % 8 == 0 && cfg->rn_depths[i] >= 1 &&
                      (cfg->rn_strides[i] == 1 || cfg->rn_strides[i] == 2),
                  "RegNet stage %d: width %d (multiple of 8), groups %d (group width a multiple of 4), depth %d, stride %d", i + 1, w, g,
                  cfg->rn_depths[i], cfg->rn_strides[i]);
    }
    int os_ = 2;
    for (int i = 0; i < 4; ++i) os_ *= cfg->rn_strides[i];
    EMP_REQUIRE(cfg->rn_strides[0] == 2 && (os_ == 32 || os_ == 16) && (cfg->arch == 0 || os_ == 32),
                "RegNet: stage 1 must be at stride 4 and the encoder's output stride 16 or 32 (BiFPN: 32)");
  }
  emp_pdl* n = new (std::nothrow) emp_pdl();
  if (!n) return EMP_ERR_NOMEM;
  n->cfg = *cfg;
  n->dec_ch = cfg->arch == 1 ? cfg->fpn_dim : cfg->decoder_channels;
  n->aspp_ch = cfg->aspp_channels > 0 ? cfg->aspp_channels : cfg->decoder_channels;
  n->ncls = cfg->num_classes;
  build_param_list(n);
  *out = n;
  return EMP_OK;
}

void emp_pdl_destroy(emp_pdl_t* net) { delete net; }

int emp_pdl_num_params(const emp_pdl_t* net) { return net ? (int)net->param_names.size() : 0; }
const char* emp_pdl_param_name(const emp_pdl_t* net, int i) {
  if (!net || i < 0 || i >= (int)net->param_names.size()) return nullptr;
  return net->param_names[i].c_str();
}

int emp_pdl_set_param(emp_pdl_t* net, const char* name, const float* h_w, const int64_t* shape, int ndim,
                      const float* h_b) {
  EMP_REQUIRE(net && name && h_w && shape, "set_param: null argument");
  auto it = net->params.find(name);
  EMP_REQUIRE(it != net->params.end(), "set_param: unknown parameter '%s'", name);
  EMP_REQUIRE(ndim == 1 || ndim == 3 || ndim == 4, "set_param(%s): ndim must be 1, 3 or 4", name);
  HostParam& hp = it->second;
  hp.shape.assign(shape, shape + ndim);
  size_t cnt = 1;
  for (int i = 0; i < ndim; ++i) {
    EMP_REQUIRE(shape[i] > 0, "set_param(%s): non-positive dimension", name);
    cnt *= (size_t)shape[i];
  }
  hp.w.assign(h_w, h_w + cnt);
  hp.b.assign((size_t)shape[0], 0.f);
  if (h_b) hp.b.assign(h_b, h_b + shape[0]);
  hp.set = true;
  net->finalized = false;
  return EMP_OK;
}

int emp_pdl_finalize(emp_pdl_t* n) {
  EMP_REQUIRE(n, "finalize: null network");
  for (const auto& nm : n->param_names)
    if (!n->params[nm].set) {
      set_error("finalize: parameter '%s' was never set", nm.c_str());
      return EMP_ERR_STATE;
    }
  const emp_pdl_config& c = n->cfg;
  if (c.encoder == 1 && n->fp32_graph()) {
    // RegNet in the fp32 mode: none of the fp16 packs below; what run32 reads besides finalize32's conv weights:
    if (c.arch == 1) {
      for (const auto& nm : n->param_names)
        if (nm.size() > 8 && nm.compare(nm.size() - 8, 8, ".weights") == 0) {
          const HostParam& hp = n->params[nm];
          EMP_REQUIRE(hp.w.size() == 5, "%s: expected 5 fusion weights", nm.c_str());
          std::vector<float> w(5);
          float sum = 0.f;
          for (int i = 0; i < 5; ++i) { w[i] = hp.w[i] > 0.f ? hp.w[i] : 0.f; sum += w[i]; }
          for (int i = 0; i < 5; ++i) w[i] = w[i] / (sum + 1e-4f);   // bifpn.py:52-55
          n->fusew[nm] = w;
        }
    } else {
      const char* decs[2] = {"semantic_decoder", "instance_decoder"};
      for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
        const std::string p = decs[d];
        const int A = n->aspp_ch;
        RC(upload_f32(n, p + ".pool.w", n->params[p + ".aspp.convs.4.aspp_pooling.1"].w));
        const HostParam& hp = n->params[p + ".aspp.project.0"];
        EMP_REQUIRE(hp.shape[0] == A && hp.shape[1] == 5 * A, "%s.aspp.project.0 must be (%d,%d,1,1)", p.c_str(), A, 5 * A);
        std::vector<float> tail((size_t)A * A);
        for (int o = 0; o < A; ++o)
          for (int i = 0; i < A; ++i) tail[(size_t)o * A + i] = hp.w[(size_t)o * 5 * A + 4 * A + i];
        RC(upload_f32(n, p + ".projpool.w", tail));
      }
    }
    const char* heads32[3] = {"semantic_head", "ins_center", "ins_xy"};
    for (int k = 0; k < 3; ++k) {
      const std::string p = heads32[k];
      RC(upload_f32(n, p + ".head.1.w", n->params[p + ".head.1"].w));
      RC(upload_f32(n, p + ".head.1.b", n->params[p + ".head.1"].b));
    }
    RC(upload_f32(n, "pr.predictor.b", n->params["semantic_pr.point_head.predictor"].b));
    RC(finalize32(n));
    n->finalized = true;
    return EMP_OK;
  }
  if (c.encoder == 1) {
    // RegNet on the fp16 engine (regnet.py:38-160): every layer on the generic implicit-GEMM conv (conv_igemm.hip); the
    // grouped 3x3 as one launch per group on a channel slice (group width 56 / 72 -> K padded to 64 / 128 with zero
    // weights, Cout = the group width); the squeeze convolution's width (w / 4) padded to a multiple of 8 with zero rows
    RC(upload_regnet_stem(n));
    for (int si = 1; si <= 4; ++si)
      for (int b = 1; b <= c.rn_depths[si - 1]; ++b) {
        const std::string p = "encoder.stage" + std::to_string(si) + ".block" + std::to_string(b);
        const int w = c.rn_widths[si - 1], g = c.rn_groups[si - 1], gw = w / g;
        RC(pack_conv(n, p + ".bottleneck.a.0"));
        {
          const HostParam& hb = n->params.at(p + ".bottleneck.b.0");
          EMP_REQUIRE(hb.shape.size() == 4 && hb.shape[0] == w && hb.shape[1] == gw && hb.shape[2] == 3 && hb.shape[3] == 3 && gw % 8 == 0,
                      "%s.bottleneck.b.0 must be (%d,%d,3,3) with a group width that is a multiple of 8", p.c_str(), w, gw);
          // a group's couts padded to its K padding (56 -> 64, 72 -> 128) with zero rows and zero bias: a 64 -> 64 / 128 -> 128
          // 3x3 is what the register-weight kernels take (conv3x3c64.hip, ~1 PFLOP/s on ResNet layer1's conv2).  The zeros
          // a group writes past its width land on the first channels of the NEXT group -- whose launch follows on the
          // same stream and overwrites them -- or, for the last group, on the row's zero tail
          const int gwp = round_up(gw, 64);
          for (int gi = 0; gi < g; ++gi) {
            HostParam t;
            t.shape = {gwp, gw, 3, 3};
            t.w.assign((size_t)gwp * gw * 9, 0.f);
            std::copy(hb.w.begin() + (size_t)gi * gw * gw * 9, hb.w.begin() + (size_t)(gi + 1) * gw * gw * 9, t.w.begin());
            t.b.assign((size_t)gwp, 0.f);
            std::copy(hb.b.begin() + (size_t)gi * gw, hb.b.begin() + (size_t)(gi + 1) * gw, t.b.begin());
            const std::string tn = p + ".bottleneck.b.0#" + std::to_string(gi);
            n->params[tn] = t;
            const int rc = pack_conv(n, tn);
            n->params.erase(tn);
            if (rc) return rc;
          }
          // ... and all groups in one blob [g][gw][9][gw padded to 64] for the grouped launch (no cout padding: a group
          // stores its own couts only, the groups run side by side)
          {
            const std::string tn = p + ".bottleneck.b.0#grouped";
            n->params[tn] = hb;
            const int rc = pack_conv(n, tn);
            n->params.erase(tn);
            if (rc) return rc;
          }
        }
        if (c.rn_se) {
          const HostParam& h0 = n->params.at(p + ".bottleneck.se.se.0");
          const int ns = (int)h0.shape[0], ns8 = round_up(ns, 8);
          HostParam t;      // zero rows + zero bias for the padded squeeze channels: relu(0) = 0 meets zero input weights
          t.shape = {ns8, h0.shape[1], 1, 1};
          t.w.assign((size_t)ns8 * h0.shape[1], 0.f);
          std::copy(h0.w.begin(), h0.w.end(), t.w.begin());
          t.b.assign((size_t)ns8, 0.f);
          std::copy(h0.b.begin(), h0.b.end(), t.b.begin());
          const std::string tn = p + ".bottleneck.se.se.0#pad";
          n->params[tn] = t;
          int rc = pack_conv(n, tn);
          n->params.erase(tn);
          if (rc) return rc;
          RC(pack_conv(n, p + ".bottleneck.se.se.2"));
        }
        RC(pack_conv(n, p + ".bottleneck.c.0"));
        if (regnet_has_shortcut(c, si, b)) RC(pack_conv(n, p + ".downsample.conv.0"));
      }
  } else {
  // stem: (64,1,7,7) -> [49][64] fp32
  {
    const HostParam& hp = n->params["encoder.conv1"];
    EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[0] == 64 && hp.shape[1] == 1 && hp.shape[2] == 7 && hp.shape[3] == 7,
                "encoder.conv1 must be (64,1,7,7)");
    std::vector<float> w(49 * 64);
    for (int o = 0; o < 64; ++o)
      for (int t = 0; t < 49; ++t) w[t * 64 + o] = hp.w[o * 49 + t];
    RC(upload_f32(n, "stem.w", w));
    RC(upload_f32(n, "stem.b", hp.b));
  }
  for (int li = 1; li <= 4; ++li)
    for (int b = 0; b < kLayers[li - 1]; ++b) {
      std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
      RC(pack_conv(n, p + ".conv1"));
      RC(pack_conv(n, p + ".conv2"));
      RC(pack_conv(n, p + ".conv3"));
      if (b == 0) {
        RC(pack_conv(n, p + ".downsample.0"));
        RC(pack_conv3_ds(n, p));
      }
    }
  }
  if (c.arch == 1) {
    const int F = c.fpn_dim;
    RC(pack_conv(n, "p2_resample.conv.0"));
    for (const auto& nm : n->param_names) {
      const bool fpn = nm.find("_fpn.") != std::string::npos, dec = nm.find("_decoder.") != std::string::npos;
      if (!fpn && !dec) continue;
      if (nm.size() > 8 && nm.compare(nm.size() - 8, 8, ".weights") == 0) {
        const HostParam& hp = n->params[nm];
        EMP_REQUIRE(hp.w.size() == 5, "%s: expected 5 fusion weights", nm.c_str());
        std::vector<float> w(5);
        float sum = 0.f;
        for (int i = 0; i < 5; ++i) { w[i] = hp.w[i] > 0.f ? hp.w[i] : 0.f; sum += w[i]; }
        for (int i = 0; i < 5; ++i) w[i] = w[i] / (sum + 1e-4f);   // bifpn.py:52-55
        n->fusew[nm] = w;
      } else if (nm.find(".sepconv.0") != std::string::npos) {
        const HostParam& hp = n->params[nm];
        RC(pack_dw(n, nm, round_up((int)hp.shape[0], 64), nm.find(".after_combines.") != std::string::npos && fsplit_on(n, nm)));
      } else if (nm.find(".upsamplings.") != std::string::npos) {
        const char* cdec = c.ins_decoder ? "instance_decoder." : "semantic_decoder.";
        RC(pack_convT(n, nm, wsplit_on(n) && nm.compare(0, strlen(cdec), cdec) == 0));
        EMP_REQUIRE(n->convs[nm].cout == 4 * F, "%s: transposed conv must produce fpn_dim channels", nm.c_str());
      } else {
        RC(pack_conv(n, nm));
        if (nm.size() > 19 && nm.compare(nm.size() - 19, 19, ".fusion.0.sepconv.1") == 0) {
          RC(pack_sepconv_pw(n, nm));
          if (precise_layer(n, nm.substr(0, nm.size() - 10))) RC(pack_sepconvp_pw(n, nm));
        }
        if (nm.find(".after_combines.0.0.sepconv.1") != std::string::npos) {      // BiFPN nodes
          RC(pack_sepconv_pw(n, nm));
          if (fsplit_on(n, nm)) RC(pack_sepconvp_pw(n, nm, true));
          else if (precise_node(n, nm)) RC(pack_sepconvp_pw(n, nm));
        }
      }
    }
  } else {
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
    std::string p = decs[d];
    for (int i = 0; i <= 3; ++i) RC(pack_conv(n, p + ".aspp.convs." + std::to_string(i) + ".0"));
    if (d == 1 && n->fuse_aspp) {
      // both decoders' ASPP branch i read the same stride-16 map: one conv of 2 x aspp_ch couts, the second 256-cout
      // tile lands in the instance decoder's concat buffer (ConvParams::out2); the pixel tile is fetched once
      for (int i = 0; i <= 3; ++i) {
        const std::string b = ".aspp.convs." + std::to_string(i) + ".0";
        const HostParam& hs = n->params.at(std::string(decs[0]) + b);
        const HostParam& hi = n->params.at(std::string(decs[1]) + b);
        if (hs.shape != hi.shape || hs.shape[0] % 256 != 0) continue;
        HostParam m;
        m.shape = hs.shape;
        m.shape[0] = hs.shape[0] + hi.shape[0];
        m.w = hs.w;
        m.w.insert(m.w.end(), hi.w.begin(), hi.w.end());
        m.b = hs.b;
        m.b.insert(m.b.end(), hi.b.begin(), hi.b.end());
        const std::string pm = "decoders" + b;
        n->params[pm] = m;
        int rc = pack_conv(n, pm);
        n->params.erase(pm);
        if (rc) return rc;
      }
    }
    RC(upload_f32(n, p + ".pool.w", n->params[p + ".aspp.convs.4.aspp_pooling.1"].w));  // (aspp, 2048) row-major
    // projection (aspp, 5*aspp): first 4*aspp input channels -> conv, last aspp -> per-image bias GEMV
    {
      const HostParam& hp = n->params[p + ".aspp.project.0"];
      const int A = n->aspp_ch;
      EMP_REQUIRE(hp.shape[0] == A && hp.shape[1] == 5 * A, "%s.aspp.project.0 must be (%d,%d,1,1)", p.c_str(), A, 5 * A);
      HostParam head;
      head.shape = {A, 4 * A, 1, 1};
      head.w.resize((size_t)A * 4 * A);
      head.b = hp.b;
      std::vector<float> tail((size_t)A * A);
      for (int o = 0; o < A; ++o) {
        for (int i = 0; i < 4 * A; ++i) head.w[(size_t)o * 4 * A + i] = hp.w[(size_t)o * 5 * A + i];
        for (int i = 0; i < A; ++i) tail[(size_t)o * A + i] = hp.w[(size_t)o * 5 * A + 4 * A + i];
      }
      head.set = true;
      HostParam keep = hp;
      n->params[p + ".aspp.project.0"] = head;
      int rc = pack_conv(n, p + ".aspp.project.0");
      n->params[p + ".aspp.project.0"] = keep;
      if (rc) return rc;
      RC(upload_f32(n, p + ".projpool.w", tail));
    }
    int xch = n->aspp_ch;
    for (int i = 0; i < c.n_stages; ++i) {
      const int lp = d == 0 ? c.low_level_proj_sem[i] : c.low_level_proj_ins[i];
      const int cpad = round_up(xch + lp, 64);
      RC(pack_conv(n, p + ".project." + std::to_string(i) + ".0"));
      EMP_REQUIRE(n->convs[p + ".project." + std::to_string(i) + ".0"].cout == lp, "%s.project.%d: Cout != %d", p.c_str(), i, lp);
      RC(pack_dw(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.0", cpad));
      RC(pack_conv(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1", cpad));
      RC(pack_sepconv_pw(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1"));
      if (precise_layer(n, p + ".fuse." + std::to_string(i) + ".0")) RC(pack_sepconvp_pw(n, p + ".fuse." + std::to_string(i) + ".0.sepconv.1"));
      xch = n->dec_ch;
    }
  }
  // the two decoders' low-level projections of one pyramid level as ONE conv with two destinations
  if (n->fuse_ds && c.ins_decoder)
    for (int i = 0; i < c.n_stages; ++i) {
      const HostParam& hs = n->params.at("semantic_decoder.project." + std::to_string(i) + ".0");
      const HostParam& hi = n->params.at("instance_decoder.project." + std::to_string(i) + ".0");
      if (hs.shape[1] != hi.shape[1] || hs.shape[0] % 8 != 0) continue;
      HostParam m;
      m.shape = {hs.shape[0] + hi.shape[0], hs.shape[1], 1, 1};
      m.w = hs.w;
      m.w.insert(m.w.end(), hi.w.begin(), hi.w.end());
      m.b = hs.b;
      m.b.insert(m.b.end(), hi.b.begin(), hi.b.end());
      const std::string pm = "decoders.project." + std::to_string(i);
      n->params[pm] = m;
      int rc = pack_conv(n, pm);
      n->params.erase(pm);
      if (rc) return rc;
      // ... and, when that level is a ResNet stage output whose next consumer is the following stage's first conv1 (a
      // 1x1 conv of the same map), all THREE as one conv with three destinations: the map (the widest the projections
      // read: 1 GiB per 32 tiles at stride 4) is read once instead of twice
      const int st = c.low_level_stages[i];
      const std::string c1n = "encoder.layer" + std::to_string(st + 1) + ".0.conv1";
      if (n->fuse_proj && st >= 1 && st < 4 && n->params.count(c1n)) {
        const HostParam& h1 = n->params.at(c1n);
        if (h1.shape.size() == 4 && h1.shape[2] == 1 && h1.shape[3] == 1 && h1.shape[1] == hs.shape[1] && h1.shape[0] % 8 == 0) {
          HostParam t;
          t.shape = {h1.shape[0] + m.shape[0], h1.shape[1], 1, 1};
          t.w = h1.w;
          t.w.insert(t.w.end(), m.w.begin(), m.w.end());
          t.b = h1.b;
          t.b.insert(t.b.end(), m.b.begin(), m.b.end());
          const std::string tm = c1n + "+project." + std::to_string(i);
          n->params[tm] = t;
          rc = pack_conv(n, tm);
          n->params.erase(tm);
          if (rc) return rc;
        }
      }
    }
  }
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  for (int k = 0; k < 3; ++k) {
    std::string p = heads[k];
    RC(pack_dw(n, p + ".head.0.0.sepconv.0", n->dec_ch));
    RC(pack_conv(n, p + ".head.0.0.sepconv.1"));
    RC(pack_sepconv_pw(n, p + ".head.0.0.sepconv.1"));
    if (precise_layer(n, p + ".head.0.0")) RC(pack_sepconvp_pw(n, p + ".head.0.0.sepconv.1"));
    RC(upload_f32(n, p + ".head.1.w", n->params[p + ".head.1"].w));
    RC(upload_f32(n, p + ".head.1.b", n->params[p + ".head.1"].b));
  }
  const int ldp = round_up(n->dec_ch + n->ncls, 64);
  for (int k = 0; k < c.num_fc; ++k) RC(pack_conv(n, "semantic_pr.point_head.fc_layers." + std::to_string(k) + ".0", ldp));
  {
    const HostParam& hp = n->params["semantic_pr.point_head.predictor"];
    const int K = (int)hp.shape[1];
    std::vector<float> w((size_t)n->ncls * ldp, 0.f);
    for (int o = 0; o < n->ncls; ++o)
      for (int i = 0; i < K; ++i) w[(size_t)o * ldp + i] = hp.w[(size_t)o * K + i];
    RC(upload_f32(n, "pr.predictor.w", w));
    RC(upload_f32(n, "pr.predictor.b", hp.b));
  }
  if (n->fp32_graph()) RC(finalize32(n));      // fp32 reference mode: fp32 copies of every weight
  if (n->pack256 && !n->fp32_graph()) {
    // packed weight images of the 256 x 256 tile (whole 128-byte lines per LDS-DMA instruction; conv_igemm256.hip pack256_kernel) for
    // every layer whose shape the tile takes at SOME batch: Cout % 256 == 0, 64-channel granularity along K, K >= 512 (conv_uses_256's
    // floor; K >= 128 with a second source).  A second copy of the layer3 / layer4 / ASPP weights (~90 MB; emp_pdl_arena_bytes
    // counts activations only, as before).
    for (auto& kv : n->convs) {
      DevConv& dc = kv.second;
      const int K = dc.kh * dc.kw * dc.cin_pad + dc.cin2_pad;
      if (dc.w256 || !dc.w || dc.cout % 256 != 0 || dc.cin_pad % 64 != 0 || dc.cin2_pad % 64 != 0 || K < (dc.cin2_pad ? 128 : 512)) continue;
      const size_t halfs = (size_t)dc.cout * K;
      EMP_CHECK_HIP(hipMalloc((void**)&dc.w256, halfs * sizeof(half_t)));
      n->owned.push_back(dc.w256);
      n->image_bytes += halfs * sizeof(half_t);
      const int prc = conv256_pack_weights(dc.w, dc.w256, dc.cout, dc.kh * dc.kw, dc.cin_pad, dc.cin2_pad, nullptr);
      if (prc) return prc;
    }
    EMP_CHECK_HIP(hipStreamSynchronize(nullptr));
  }
  n->finalized = true;
  return EMP_OK;
}

int emp_pdl_set_precision(emp_pdl_t* net, int precision) {
  EMP_REQUIRE(net, "set_precision: null network");
  EMP_REQUIRE(precision >= 0 && precision <= 2, "set_precision: 0 = fp16 engine, 1 = fp32 reference mode, 2 = fp16x3 mode (got %d)", precision);
  EMP_REQUIRE(!net->finalized, "set_precision: call before emp_pdl_finalize");
  net->precision = precision;
  return EMP_OK;
}
int emp_pdl_precision(const emp_pdl_t* net) { return net ? net->precision : -1; }

int emp_pdl_reserve(emp_pdl_t* net, int N, int H, int W) {
  EMP_REQUIRE(net, "reserve: null network");
  if (net->fp32_graph()) return EMP_OK;      // the fp32 / fp16x3 modes size its buffers on first use
  return plan(net, N, H, W, net->pRS > 2 ? net->pRS : 2, nullptr);
}
size_t emp_pdl_arena_bytes(const emp_pdl_t* net) { return net ? net->arena_used : 0; }

int emp_pdl_forward_padded(emp_pdl_t* net, const void* d_image, int image_dtype, float sub, float mul, int N, int vh, int vw,
                           int H, int W, int render_steps, int interpolate_ins, float* d_sem_logits, float* d_ctr_hmp,
                           float* d_offsets, void* stream) {
  EMP_REQUIRE(net && d_image && d_sem_logits && d_ctr_hmp && d_offsets, "forward: null argument");
  EMP_REQUIRE(vh > 0 && vw > 0 && vh <= H && vw <= W, "forward: valid size %dx%d must fit the padded size %dx%d", vh, vw, H, W);
  if (!net->finalized) {
    set_error("forward: call emp_pdl_finalize first");
    return EMP_ERR_STATE;
  }
  if (net->fp32_graph())
    return run32(net, d_image, image_dtype, sub, mul, N, H, W, vh, vw, render_steps, interpolate_ins, d_sem_logits, d_ctr_hmp,
                 d_offsets, (hipStream_t)stream);
  if (H != net->pH || W != net->pW || render_steps != net->pRS || N > net->capN) {
    RC(plan(net, N, H, W, render_steps, (hipStream_t)stream));
  } else if (N != net->pN) {
    rebatch(net, N);
  }
  const int rc = run(net, d_image, image_dtype, sub, mul, N, H, W, vh, vw, render_steps, interpolate_ins, d_sem_logits,
                     d_ctr_hmp, d_offsets, (hipStream_t)stream);
  if (net->layer_log) { fprintf(net->layer_log, "end\n"); fflush(net->layer_log); }
  return rc;
}

int emp_pdl_forward(emp_pdl_t* net, const void* d_image, int image_dtype, float sub, float mul, int N, int H, int W,
                    int render_steps, int interpolate_ins, float* d_sem_logits, float* d_ctr_hmp, float* d_offsets,
                    void* stream) {
  return emp_pdl_forward_padded(net, d_image, image_dtype, sub, mul, N, H, W, H, W, render_steps, interpolate_ins,
                                d_sem_logits, d_ctr_hmp, d_offsets, stream);
}

double emp_pdl_flops(const emp_pdl_t* net, int, int, int, int) { return net ? net->flops : 0.0; }

int emp_pdl_profile(emp_pdl_t* net, int enable) {
  EMP_REQUIRE(net != nullptr, "profile: null network");
  net->profile = enable != 0;
  net->prof_used = 0;
  net->prof_flops = 0.0;
  return EMP_OK;
}

int emp_pdl_profile_read(emp_pdl_t* net, double* ms_total, double* flops_total, int* launches) {
  EMP_REQUIRE(net && ms_total && flops_total && launches, "profile_read: null pointer");
  double ms = 0.0;
  for (size_t i = 0; i < net->prof_used; ++i) {
    float t = 0.f;
    EMP_CHECK_HIP(hipEventSynchronize(net->prof_events[i].second));
    EMP_CHECK_HIP(hipEventElapsedTime(&t, net->prof_events[i].first, net->prof_events[i].second));
    ms += t;
  }
  *ms_total = ms;
  *flops_total = net->prof_flops;
  *launches = (int)net->prof_used;
  net->prof_used = 0;
  net->prof_flops = 0.0;
  return EMP_OK;
}

int emp_pdl_num_taps(const emp_pdl_t* net) { return net ? (int)net->act_order.size() : 0; }
const char* emp_pdl_tap_name(const emp_pdl_t* net, int i) {
  if (!net || i < 0 || i >= (int)net->act_order.size()) return nullptr;
  return net->act_order[i].c_str();
}
int emp_pdl_tap(emp_pdl_t* net, const char* name, void** d_ptr, int64_t shape5[5]) {
  EMP_REQUIRE(net && name && d_ptr && shape5, "tap: null argument");
  auto it = net->acts.find(name);
  EMP_REQUIRE(it != net->acts.end(), "tap: unknown activation '%s'", name);
  const Act& a = it->second;
  *d_ptr = a.p;
  shape5[0] = a.N; shape5[1] = a.H; shape5[2] = a.W; shape5[3] = a.C; shape5[4] = a.ld;
  return EMP_OK;
}

int emp_pdl_tap_raw(emp_pdl_t* net, const char* name, void** d_ptr, int64_t* bytes) {
  EMP_REQUIRE(net && name && d_ptr && bytes, "tap_raw: null argument");
  if (net->fp32_graph()) {      // fp32 / fp16x3 mode: every buffer of the last forward by name (maps: NHWC fp32)
    auto it32 = net->pool32.find(name);
    EMP_REQUIRE(it32 != net->pool32.end(), "tap_raw: unknown fp32 buffer '%s'", name);
    *d_ptr = it32->second.first;
    *bytes = (int64_t)(it32->second.second * sizeof(float));
    return EMP_OK;
  }
  auto it = net->raw.find(name);
  EMP_REQUIRE(it != net->raw.end(), "tap_raw: unknown buffer '%s'", name);
  *d_ptr = net->arena + it->second.first;
  *bytes = (int64_t)it->second.second;
  return EMP_OK;
}

}  // extern "C"
