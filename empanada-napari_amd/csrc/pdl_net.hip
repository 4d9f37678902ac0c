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
  float* b = nullptr;
  int cout = 0, cin = 0, cin_pad = 0, kh = 1, kw = 1;
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

  // device parameters
  std::map<std::string, DevConv> convs;
  std::map<std::string, float*> f32w;  // fp32 device blobs (stem, gemv, heads)
  std::map<std::string, half_t*> f16w;  // fp16 device blobs (depthwise taps)
  std::vector<void*> owned;

  // arena
  char* arena = nullptr;
  size_t arena_cap = 0, arena_used = 0;
  int pN = 0, pH = 0, pW = 0, pRS = 0;  // planned shape
  std::map<std::string, Act> acts;
  std::vector<std::string> act_order;
  std::map<std::string, std::pair<size_t, size_t>> raw;  // name -> (offset, bytes)
  double flops = 0.0;

  ~emp_pdl() {
    for (void* p : owned) (void)hipFree(p);
    if (arena) (void)hipFree(arena);
  }
};

namespace {

const int kLayers[4] = {3, 4, 6, 3};
const int kPlanes[4] = {64, 128, 256, 512};

void expect(emp_pdl* n, const std::string& name) {
  n->param_names.push_back(name);
  n->params[name] = HostParam();
}

void build_param_list(emp_pdl* n) {
  const emp_pdl_config& c = n->cfg;
  expect(n, "encoder.conv1");
  for (int li = 1; li <= 4; ++li)
    for (int b = 0; b < kLayers[li - 1]; ++b) {
      std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
      expect(n, p + ".conv1");
      expect(n, p + ".conv2");
      expect(n, p + ".conv3");
      if (b == 0) expect(n, p + ".downsample.0");
    }
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
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

int upload_f32(emp_pdl* n, const std::string& key, const std::vector<float>& v) {
  void* d;
  int rc = dev_upload(n, v.data(), v.size() * sizeof(float), &d);
  if (rc) return rc;
  n->f32w[key] = (float*)d;
  return EMP_OK;
}

// depthwise (C,1,5,5) -> [25][Cpad] fp16
int pack_dw(emp_pdl* n, const std::string& name, int cpad) {
  const HostParam& hp = n->params.at(name);
  const int C = (int)hp.shape[0];
  EMP_REQUIRE(hp.shape.size() == 4 && hp.shape[1] == 1 && hp.shape[2] == 5 && hp.shape[3] == 5 && cpad >= C,
              "%s: expected a (C,1,5,5) depthwise weight", name.c_str());
  EMP_REQUIRE(cpad % 64 == 0, "%s: padded channel count must be a multiple of 64", name.c_str());
  std::vector<half_t> pk((size_t)25 * cpad, (half_t)0.f);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < 25; ++t) pk[(size_t)t * cpad + c] = (half_t)hp.w[(size_t)c * 25 + t];
  void* d;
  int rc = dev_upload(n, pk.data(), pk.size() * sizeof(half_t), &d);
  if (rc) return rc;
  n->f16w[name] = (half_t*)d;
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



int plan(emp_pdl* n, int N, int H, int W, int RS) {
  const emp_pdl_config& c = n->cfg;
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0,
              "forward: H=%d W=%d must be positive multiples of 16 (factor_pad first)", H, W);
  EMP_REQUIRE(RS >= 1 && RS <= 6, "render_steps=%d out of range", RS);
  n->acts.clear();
  n->act_order.clear();
  n->raw.clear();
  Planner pl;
  add_raw(n, pl, "zero", 256);
  add_act(n, pl, "stem", N, H / 2, W / 2, 64);
  add_act(n, pl, "p1", N, H / 4, W / 4, 64);
  int h = H / 4, w = W / 4, inpl = 64;
  int ph[5], pw[5];
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
  const int hq = ph[st_last], wq = pw[st_last];  // resolution of semantic_x (1/4 for MitoNet)
  EMP_REQUIRE(hq * 4 == H && wq * 4 == W, "the last decoder stage must be at 1/4 resolution (got %dx%d)", hq, wq);
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
  // channel-padding regions must read as exact zeros
  EMP_CHECK_HIP(hipMemset(n->arena, 0, need));
  n->arena_used = need;
  for (auto& kv : n->acts) kv.second.p = (half_t*)(n->arena + kv.second.off);
  n->pN = N; n->pH = H; n->pW = W; n->pRS = RS;
  return EMP_OK;
}

template <typename T>
T* rawp(emp_pdl* n, const std::string& name) {
  return (T*)(n->arena + n->raw.at(name).first);
}

// conv helper: in (channels [0,Cin_pad) of `in`), out channels [coff, coff+Cout) of `out`
int conv(emp_pdl* n, const std::string& wname, const Act& in, int in_coff, const Act& out, int out_coff, int stride,
         int pad, int dil, bool relu, const Act* res, const float* bias_n, hipStream_t s) {
  const DevConv& dc = n->convs.at(wname);
  ConvParams p{};
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
  EMP_REQUIRE(p.Ho == out.H && p.Wo == out.W && in.N == out.N, "%s: output shape mismatch (%dx%d vs %dx%d)",
              wname.c_str(), p.Ho, p.Wo, out.H, out.W);
  EMP_REQUIRE(in_coff + dc.cin_pad <= in.ld && out_coff + dc.cout <= out.ld, "%s: channel slice out of range",
              wname.c_str());
  p.out_ld = out.ld;
  p.relu = relu ? 1 : 0;
  p.M = p.N * p.Ho * p.Wo;
  n->flops += 2.0 * (double)p.M * dc.cout * (double)(dc.cin * dc.kh * dc.kw);
  return launch_conv_igemm(p, 0, s);
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

  // ---- encoder ----
  RC(launch_stem7x7(img, dtype, sub, mul, N, H, W, vh, vw, n->f32w.at("stem.w"), n->f32w.at("stem.b"), A("stem").p, s));
  n->flops += 2.0 * N * (H / 2) * (W / 2) * 64.0 * 49.0;
  RC(launch_maxpool3x3s2(A("stem").p, N, H / 2, W / 2, 64, A("p1").p, s));
  std::string xname = "p1";
  std::string pyr[5];
  pyr[0] = "p1";
  for (int li = 1; li <= 4; ++li) {
    int stride = li == 1 ? 1 : 2, dil = 1;
    if (li == 4 && c.stage4_stride == 16) { stride = 1; dil = 2; }
    for (int b = 0; b < kLayers[li - 1]; ++b) {
      const int sb = b == 0 ? stride : 1;
      std::string p = "encoder.layer" + std::to_string(li) + "." + std::to_string(b);
      RC(conv(n, p + ".conv1", A(xname), 0, A(p + ".c1"), 0, 1, 0, 1, true, nullptr, nullptr, s));
      RC(conv(n, p + ".conv2", A(p + ".c1"), 0, A(p + ".c2"), 0, sb, dil, dil, true, nullptr, nullptr, s));
      const Act* idn = &A(xname);
      if (b == 0) {
        RC(conv(n, p + ".downsample.0", A(xname), 0, A(p + ".ds"), 0, sb, 0, 1, false, nullptr, nullptr, s));
        idn = &A(p + ".ds");
      }
      RC(conv(n, p + ".conv3", A(p + ".c2"), 0, A(p), 0, 1, 0, 1, true, idn, nullptr, s));
      xname = p;
    }
    pyr[li] = xname;
  }
  const Act& p5 = A(pyr[4]);

  // ---- decoders ----
  RC(launch_avgpool(p5.p, N, p5.H * p5.W, p5.C, p5.ld, rawp<float>(n, "pooled"), rawp<float>(n, "pool_part"), s));
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  std::string dec_out[2];
  for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
    std::string p = decs[d];
    float* poolfeat = rawp<float>(n, p + ".poolfeat");
    float* bias_n = rawp<float>(n, p + ".bias_n");
    RC(launch_gemv(rawp<float>(n, "pooled"), N, p5.C, n->f32w.at(p + ".pool.w"), nullptr, n->aspp_ch, 1, poolfeat, s));
    RC(launch_gemv(poolfeat, N, n->aspp_ch, n->f32w.at(p + ".projpool.w"), nullptr, n->aspp_ch, 0, bias_n, s));
    const Act& cat = A(p + ".aspp.cat");
    RC(conv(n, p + ".aspp.convs.0.0", p5, 0, cat, 0, 1, 0, 1, true, nullptr, nullptr, s));
    for (int i = 1; i <= 3; ++i) {
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
      RC(conv(n, p + ".project." + std::to_string(i) + ".0", A(pyr[st]), 0, cb, xch, 1, 0, 1, true, nullptr, nullptr, s));
      const std::string fz = p + ".fuse." + std::to_string(i) + ".0.sepconv.";
      RC(launch_dwconv5x5(cb.p, N, cb.H, cb.W, cb.ld, cb.ld, n->f16w.at(fz + "0"), A(q + ".dw").p, cb.ld,
                          rawp<half_t>(n, "zero"), s));
      n->flops += 2.0 * 25.0 * (double)N * cb.H * cb.W * (xch + n->convs.at(p + ".project." + std::to_string(i) + ".0").cout);
      RC(conv(n, fz + "1", A(q + ".dw"), 0, A(q + ".out"), 0, 1, 0, 1, true, nullptr, nullptr, s));
      x = q + ".out";
      xch = n->dec_ch;
    }
    dec_out[d] = x;
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
    std::string p = heads[k];
    const Act& xin = k == 0 ? semx : insx;
    RC(launch_dwconv5x5(xin.p, N, hq, wq, n->dec_ch, xin.ld, n->f16w.at(p + ".head.0.0.sepconv.0"), A(p + ".dw").p,
                        n->dec_ch, rawp<half_t>(n, "zero"), s));
    n->flops += 2.0 * 25.0 * (double)N * hq * wq * n->dec_ch;
    RC(conv(n, p + ".head.0.0.sepconv.1", A(p + ".dw"), 0, A(p + ".pw"), 0, 1, 0, 1, true, nullptr, nullptr, s));
    float* dst = rawp<float>(n, p + ".out");
    if (k == 1 && !interp) dst = o_ctr;
    if (k == 2 && !interp) dst = o_off;
    head_out[k] = dst;
    RC(launch_head1x1(A(p + ".pw").p, N, hq * wq, n->dec_ch, n->dec_ch, n->f32w.at(p + ".head.1.w"),
                      n->f32w.at(p + ".head.1.b"), hc[k], dst, (int64_t)hq * wq, nullptr, s));
    n->flops += 2.0 * (double)N * hq * wq * n->dec_ch * hc[k];
  }
  if (interp) {
    RC(launch_bilinear_ac_f32_nchw(head_out[1], N * 1, hq, wq, o_ctr, 4, s));
    RC(launch_bilinear_ac_f32_nchw(head_out[2], N * 2, hq, wq, o_off, 4, s));
  }

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
  return EMP_OK;
}

}  // namespace

extern "C" {

int emp_pdl_create(const emp_pdl_config* cfg, emp_pdl_t** out) {
  EMP_REQUIRE(cfg && out, "pdl_create: null argument");
  EMP_REQUIRE(cfg->num_classes >= 1 && cfg->num_classes <= 8, "num_classes=%d unsupported", cfg->num_classes);
  EMP_REQUIRE(cfg->stage4_stride == 16 || cfg->stage4_stride == 32, "stage4_stride must be 16 or 32");
  EMP_REQUIRE(cfg->n_stages >= 1 && cfg->n_stages <= 3, "n_stages=%d unsupported", cfg->n_stages);
  EMP_REQUIRE(cfg->decoder_channels % 64 == 0 && cfg->decoder_channels > 0, "decoder_channels must be a multiple of 64");
  EMP_REQUIRE(cfg->num_fc >= 1 && cfg->subdivision_num_points > 0, "bad PointRend configuration");
  for (int i = 0; i < cfg->n_stages; ++i) {
    EMP_REQUIRE(cfg->low_level_stages[i] >= 1 && cfg->low_level_stages[i] <= 3, "low_level_stages[%d] out of range", i);
    EMP_REQUIRE(cfg->low_level_proj_sem[i] % 8 == 0 && (!cfg->ins_decoder || cfg->low_level_proj_ins[i] % 8 == 0),
                "projected low-level channels must be multiples of 8");
  }
  emp_pdl* n = new (std::nothrow) emp_pdl();
  if (!n) return EMP_ERR_NOMEM;
  n->cfg = *cfg;
  n->dec_ch = cfg->decoder_channels;
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
  EMP_REQUIRE(ndim == 3 || ndim == 4, "set_param(%s): ndim must be 3 or 4", name);
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
      if (b == 0) RC(pack_conv(n, p + ".downsample.0"));
    }
  const char* decs[2] = {"semantic_decoder", "instance_decoder"};
  for (int d = 0; d < (c.ins_decoder ? 2 : 1); ++d) {
    std::string p = decs[d];
    for (int i = 0; i <= 3; ++i) RC(pack_conv(n, p + ".aspp.convs." + std::to_string(i) + ".0"));
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
      xch = n->dec_ch;
    }
  }
  const char* heads[3] = {"semantic_head", "ins_center", "ins_xy"};
  for (int k = 0; k < 3; ++k) {
    std::string p = heads[k];
    RC(pack_dw(n, p + ".head.0.0.sepconv.0", n->dec_ch));
    RC(pack_conv(n, p + ".head.0.0.sepconv.1"));
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
  n->finalized = true;
  return EMP_OK;
}

int emp_pdl_reserve(emp_pdl_t* net, int N, int H, int W) {
  EMP_REQUIRE(net, "reserve: null network");
  return plan(net, N, H, W, net->pRS > 2 ? net->pRS : 2);
}
size_t emp_pdl_arena_bytes(const emp_pdl_t* net) { return net ? net->arena_used : 0; }

int emp_pdl_forward(emp_pdl_t* net, const void* d_image, int image_dtype, float sub, float mul, int N, int H, int W,
                    int render_steps, int interpolate_ins, float* d_sem_logits, float* d_ctr_hmp, float* d_offsets,
                    void* stream) {
  EMP_REQUIRE(net && d_image && d_sem_logits && d_ctr_hmp && d_offsets, "forward: null argument");
  if (!net->finalized) {
    set_error("forward: call emp_pdl_finalize first");
    return EMP_ERR_STATE;
  }
  if (N != net->pN || H != net->pH || W != net->pW || render_steps != net->pRS) RC(plan(net, N, H, W, render_steps));
  return run(net, d_image, image_dtype, sub, mul, N, H, W, H, W, render_steps, interpolate_ins, d_sem_logits, d_ctr_hmp,
             d_offsets, (hipStream_t)stream);
}

double emp_pdl_flops(const emp_pdl_t* net, int, int, int, int) { return net ? net->flops : 0.0; }

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

}  // extern "C"
