// C-ABI glue of libempanada_hip: error reporting, device query and the
// operator-level entry points that are not tied to a network object.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace emp {

static thread_local char g_err[1024] = "no error";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const void* zero_page() {
  static void* page = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    if (hipMalloc(&page, 2048) != hipSuccess) { page = nullptr; return; }
    if (hipMemset(page, 0, 2048) != hipSuccess) { (void)hipFree(page); page = nullptr; }
  });
  return page;
}

int ensure_dyn_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;      // (device, kernel) -> bytes granted
  int dev = 0;
  EMP_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({dev, kernel});
  if (it != done.end() && it->second >= bytes) return EMP_OK;
  EMP_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[{dev, kernel}] = bytes;
  return EMP_OK;
}

}  // namespace emp

using namespace emp;

extern "C" {

const char* emp_last_error(void) { return g_err; }
int emp_abi_version(void) { return 5; }      // 2: emp_pdl_config carries the encoder (RegNet) fields; 3: precision 2 (fp16x3), emp_conv2d_nhwc_f16x3; 4: emp_conv256_pack_weights; 5: the hl32 plane format (emp_hl32_*, emp_x3p_pack_weights, emp_conv2d_hl32_f16x3), emp_conv2d_nhwc_f16x3_ex, default precision 2

int emp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return EMP_ERR_HIP;
  }
  return n;
}

int emp_copy_d2d(void* d_dst, const void* d_src, size_t bytes, void* stream) {
  EMP_REQUIRE(d_dst && d_src, "copy_d2d: null pointer");
  EMP_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return EMP_OK;
}

int emp_conv2d_nhwc_f16(const void* d_in, int N, int H, int W, int Cin, int in_ld, const void* d_w,
                        const float* d_bias, const float* d_bias_n, const void* d_res, int res_ld, void* d_out,
                        int out_ld, int Cout, int KH, int KW, int stride, int pad, int dil, int relu, int variant,
                        void* stream) {
  EMP_REQUIRE(d_in && d_w && d_out, "conv2d: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d: bad geometry");
  EMP_REQUIRE(variant >= 0 && (variant & 15) <= 3 && ((variant >> 4) & 15) <= 7 && ((variant >> 8) & 255) <= 64 && (variant >> 16) <= 31, "conv2d: bad variant %d", variant);
  ConvParams p{};
  p.in = (const half_t*)d_in;
  p.wgt = (const half_t*)d_w;
  p.bias = d_bias;
  p.bias_n = d_bias_n;
  p.res = (const half_t*)d_res;
  p.res_ld = res_ld;
  p.out = (half_t*)d_out;
  p.zero = (const half_t*)zero_page();
  EMP_REQUIRE(p.zero != nullptr, "conv2d: could not allocate the zero page");
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.in_ld = in_ld;
  p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d: empty output");
  p.out_ld = out_ld;
  p.act = relu;
  p.ps_cout = 0;
  p.M = N * p.Ho * p.Wo;
  return launch_conv_igemm(p, variant, (hipStream_t)stream);
}

int emp_conv256_pack_weights(const void* d_w, void* d_packed, int Cout, int KT, int Cin, int Cin2, void* stream) {
  return conv256_pack_weights((const half_t*)d_w, (half_t*)d_packed, Cout, KT, Cin, Cin2, (hipStream_t)stream);
}

int emp_conv2d_nhwc_f32(const float* d_in, int N, int H, int W, int Cin, int in_ld, const float* d_w, const float* d_bias,
                        const float* d_bias_n, const float* d_res, int res_ld, float* d_out, int out_ld, int Cout, int KH,
                        int KW, int stride, int pad, int dil, int act, void* stream) {
  EMP_REQUIRE(d_in && d_w && d_out, "conv2d_f32: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d_f32: bad geometry");
  Conv32 p{};
  p.in = d_in; p.in_ld = in_ld; p.w = d_w; p.bias = d_bias; p.bias_n = d_bias_n; p.res = d_res; p.res_ld = res_ld;
  p.out = d_out; p.out_ld = out_ld;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && out_ld >= Cout && (d_res == nullptr || res_ld >= Cout), "conv2d_f32: bad output geometry");
  p.act = act;
  return launch_conv32(p, (hipStream_t)stream);
}

int emp_conv2d_nhwc_f16x3(const float* d_in, int N, int H, int W, int Cin, int in_ld, const float* d_w, const float* d_bias,
                          const float* d_bias_n, const float* d_res, int res_ld, float* d_out, int out_ld, int Cout, int KH,
                          int KW, int stride, int pad, int dil, int act, int groups, int cin_g, void* stream) {
  EMP_REQUIRE(d_in && d_w && d_out, "conv2d_f16x3: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0 && groups >= 1, "conv2d_f16x3: bad geometry");
  Conv32 p{};
  p.in = d_in; p.in_ld = in_ld; p.w = d_w; p.bias = d_bias; p.bias_n = d_bias_n; p.res = d_res; p.res_ld = res_ld;
  p.out = d_out; p.out_ld = out_ld;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && out_ld >= groups * Cout && (d_res == nullptr || res_ld >= Cout), "conv2d_f16x3: bad output geometry");
  p.act = act;
  p.x3 = 1;
  if (groups > 1) { p.groups = groups; p.cin_g = cin_g; }
  return launch_conv32(p, (hipStream_t)stream);
}

// Test and tuning entry of every variant of the fp16x3 convolution that the network reaches and emp_conv2d_nhwc_f16x3 cannot
// express: pre-split weight pairs, the split-role kernel's LDS-DMA weight image, a K-concatenated second source,
// the fused 1x1 head, an hl32 output.  Allocates its temporaries (synchronising: not a hot-path call).
int emp_conv2d_nhwc_f16x3_ex(const float* d_in, int N, int H, int W, int Cin, int in_ld, const float* d_w, const float* d_bias,
                             const float* d_bias_n, const float* d_res, int res_ld, void* d_out, int out_ld, int out_fmt, int Cout, int KH,
                             int KW, int stride, int pad, int dil, int act, int wmode, const float* d_in2, int H2, int W2, int Cin2,
                             int in2_ld, int stride2, const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                             void* stream) {
  EMP_REQUIRE(d_in && d_w && (d_out || d_head_w), "conv2d_f16x3_ex: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0 && wmode >= 0 && (wmode & 3) <= 2 && wmode < 8,
              "conv2d_f16x3_ex: bad geometry (wmode 0 fp32 weights, 1 split pairs, 2 pairs + LDS-DMA image; + 4: split-K allowed)");
  const bool ksplit = (wmode & 4) != 0;
  wmode &= 3;
  hipStream_t s = (hipStream_t)stream;
  Conv32 p{};
  p.in = d_in; p.in_ld = in_ld; p.w = d_w; p.bias = d_bias; p.bias_n = d_bias_n; p.res = d_res; p.res_ld = res_ld;
  p.out = (float*)d_out; p.out_ld = out_ld; p.out_fmt = out_fmt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && (d_head_w || out_ld >= Cout) && (d_res == nullptr || res_ld >= Cout), "conv2d_f16x3_ex: bad output geometry");
  p.act = act;
  p.x3 = 1;
  if (d_in2) { p.in2 = d_in2; p.in2_ld = in2_ld; p.Cin2 = Cin2; p.H2 = H2; p.W2 = W2; p.stride2 = stride2; }
  const int64_t Krow = (int64_t)KH * KW * Cin + (d_in2 ? Cin2 : 0);
  std::vector<void*> tmp;
  auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); };
  int rc = EMP_OK;
  if (wmode >= 1) {
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)Cout * Krow * sizeof(uint32_t)) != hipSuccess) { cleanup(); set_error("conv2d_f16x3_ex: out of memory"); return EMP_ERR_NOMEM; }
    tmp.push_back(d);
    p.wpair = (const uint32_t*)d;
    rc = launch_split_pairs(d_w, (uint32_t*)d, (int64_t)Cout * Krow, s);
  }
  if (!rc && wmode == 2 && !d_in2) {
    const int64_t ih = x3_weight_image_halfs(Cout, (int)Krow, Cin);
    if (ih > 0) {
      void* d = nullptr;
      if (hipMalloc(&d, (size_t)ih * sizeof(half_t)) != hipSuccess) { cleanup(); set_error("conv2d_f16x3_ex: out of memory"); return EMP_ERR_NOMEM; }
      tmp.push_back(d);
      p.wimg = (const half_t*)d;
      rc = launch_x3_weight_image(d_w, (half_t*)d, Cout, (int)Krow, s);
    }
  }
  if (!rc && ksplit && !d_head_w && !d_in2) {      // scratch for the launcher's split-K rule (Conv32::kpart)
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)X3_KPART_BYTES) != hipSuccess) { cleanup(); set_error("conv2d_f16x3_ex: out of memory"); return EMP_ERR_NOMEM; }
    tmp.push_back(d);
    p.kpart = (float*)d; p.kpart_bytes = X3_KPART_BYTES;
  }
  const int64_t MP = (int64_t)N * p.Ho * p.Wo;
  const int tiles = conv16x3_cout_tiles(Cout);
  if (!rc && d_head_w) {
    EMP_REQUIRE(d_head_out && head_c >= 1 && head_c <= 4, "conv2d_f16x3_ex: the fused head writes (N, head_c, Ho * Wo) floats, head_c <= 4");
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)tiles * MP * head_c * sizeof(float)) != hipSuccess) { cleanup(); set_error("conv2d_f16x3_ex: out of memory"); return EMP_ERR_NOMEM; }
    tmp.push_back(d);
    p.head_w = d_head_w; p.head_part = (float*)d; p.head_c = head_c;
    if (!p.out) { p.out = d_head_out; p.out_ld = Cout; }      // never written: the fused head stores no activation map
  }
  if (!rc) rc = launch_conv32(p, s);
  if (!rc && d_head_w) rc = launch_head_finish_f32(p.head_part, tiles, N, p.Ho * p.Wo, head_c, d_head_b, d_head_out, s);
  hipError_t e = hipStreamSynchronize(s);
  cleanup();
  if (!rc && e != hipSuccess) { set_error("conv2d_f16x3_ex: %s", hipGetErrorString(e)); return EMP_ERR_HIP; }
  return rc;
}

int emp_sepconv_x3_pack(const float* d_dw, const float* d_pw, int ks, int C, int Cout, float* d_dw_packed, void* d_pw_packed, void* stream) {
  int rc = launch_sepx3_pack_dw(d_dw, ks, C, C, d_dw_packed, (hipStream_t)stream);
  if (rc) return rc;
  return launch_sepx3_pack_pw(d_pw, C, C, Cout, (half_t*)d_pw_packed, (hipStream_t)stream);
}
int emp_sepconv_x3_nhwc_f32(const float* d_in, int N, int H, int W, int C, int in_ld, const float* d_dw_packed, const void* d_pw_packed,
                            const float* d_bias, int Cout, int act, float* d_out, int out_ld, const float* d_head_w, const float* d_head_b,
                            int head_c, float* d_head_out, int ks, void* stream) {
  EMP_REQUIRE(N > 0 && H > 0 && W > 0, "sepconv_x3: bad geometry");
  return launch_sepconv_x3(d_in, N, H, W, C, in_ld, d_dw_packed, (const half_t*)d_pw_packed, d_bias, Cout, act, d_out, out_ld, d_head_w,
                           d_head_b, head_c, d_head_out, (int64_t)H * W, (hipStream_t)stream, ks);
}

int emp_hl32_from_f32(const float* d_in, void* d_out, int64_t rows, int C, int in_ld, int out_ld, void* stream) {
  return launch_hl32_from_f32(d_in, (half_t*)d_out, rows, C, in_ld, out_ld, (hipStream_t)stream);
}
int emp_hl32_to_f32(const void* d_in, float* d_out, int64_t rows, int C, int in_ld, int out_ld, void* stream) {
  return launch_hl32_to_f32((const half_t*)d_in, d_out, rows, C, in_ld, out_ld, (hipStream_t)stream);
}
int emp_x3p_pack_weights(const float* d_w, void* d_img, int Cout, int K, void* stream) {
  return launch_x3p_pack(d_w, (half_t*)d_img, Cout, K, (hipStream_t)stream);
}
int emp_conv2d_hl32_f16x3(const void* d_in, int N, int H, int W, int Cin, int in_ld, const void* d_wimg, const float* d_bias,
                          const float* d_bias_n, const void* d_res, int res_ld, int res_fmt, void* d_out, int out_ld, int out_fmt,
                          int Cout, int KH, int KW, int stride, int pad, int dil, int act, void* stream) {
  EMP_REQUIRE(d_in && d_wimg && d_out, "conv2d_hl32_f16x3: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d_hl32_f16x3: bad geometry");
  Conv32 p{};
  p.in = (const float*)d_in; p.in_ld = in_ld; p.in_fmt = 1; p.wimgp = (const half_t*)d_wimg;
  p.bias = d_bias; p.bias_n = d_bias_n; p.res = (const float*)d_res; p.res_ld = res_ld; p.res_fmt = res_fmt;
  p.out = (float*)d_out; p.out_ld = out_ld; p.out_fmt = out_fmt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && out_ld >= Cout && (d_res == nullptr || res_ld >= Cout), "conv2d_hl32_f16x3: bad output geometry");
  p.act = act;
  p.x3 = 1;
  return launch_conv16x3p(p, (hipStream_t)stream);
}
int emp_conv2d_hl32_f16x3_ksplit(const void* d_in, int N, int H, int W, int Cin, int in_ld, const void* d_wimg, const float* d_bias,
                                 const float* d_bias_n, const void* d_res, int res_ld, int res_fmt, void* d_out, int out_ld, int out_fmt,
                                 void* d_out2, int out2_ld, int split2, int Cout, int KH, int KW, int stride, int pad, int dil, int act,
                                 void* d_scratch, int64_t scratch_bytes, void* stream) {
  EMP_REQUIRE(d_in && d_wimg && d_out && d_scratch, "conv2d_hl32_f16x3_ksplit: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d_hl32_f16x3_ksplit: bad geometry");
  Conv32 p{};
  p.in = (const float*)d_in; p.in_ld = in_ld; p.in_fmt = 1; p.wimgp = (const half_t*)d_wimg;
  p.bias = d_bias; p.bias_n = d_bias_n; p.res = (const float*)d_res; p.res_ld = res_ld; p.res_fmt = res_fmt;
  p.out = (float*)d_out; p.out_ld = out_ld; p.out_fmt = out_fmt;
  if (d_out2) { p.out2 = (float*)d_out2; p.out2_ld = out2_ld; p.split2 = split2; }
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && out_ld >= (d_out2 ? split2 : Cout) && (d_res == nullptr || res_ld >= Cout), "conv2d_hl32_f16x3_ksplit: bad output geometry");
  p.act = act;
  p.x3 = 1;
  p.kpart = (float*)d_scratch; p.kpart_bytes = scratch_bytes;
  return launch_conv16x3p(p, (hipStream_t)stream);
}

int emp_conv2d_grouped_nhwc_f32(const float* d_in, int N, int H, int W, int groups, int cin_g, int Cin16, int in_ld, const float* d_w,
                                const float* d_bias, float* d_out, int out_ld, int cout_g, int KH, int KW, int stride, int pad,
                                int dil, int act, void* stream) {
  EMP_REQUIRE(d_in && d_w && d_out, "conv2d_grouped_f32: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0 && groups >= 1 && cout_g > 0,
              "conv2d_grouped_f32: bad geometry");
  Conv32 p{};
  p.in = d_in; p.in_ld = in_ld; p.w = d_w; p.bias = d_bias; p.out = d_out; p.out_ld = out_ld;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin16; p.Cout = cout_g; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  EMP_REQUIRE(p.Ho > 0 && p.Wo > 0 && out_ld >= groups * cout_g, "conv2d_grouped_f32: bad output geometry");
  p.act = act;
  if (groups > 1) { p.groups = groups; p.cin_g = cin_g; }
  else EMP_REQUIRE(cin_g == Cin16, "conv2d_grouped_f32: one group reads Cin16 channels");
  return launch_conv32(p, (hipStream_t)stream);
}

int emp_conv1x1_dual_nhwc_f16(const void* d_in, int N, int H, int W, int Cin, int in_ld, const void* d_in2, int H2, int W2,
                              int Cin2, int in2_ld, int stride2, const void* d_w, const float* d_bias, void* d_out,
                              int out_ld, int Cout, int relu, int variant, void* stream) {
  EMP_REQUIRE(d_in && d_in2 && d_w && d_out, "conv1x1_dual: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0, "conv1x1_dual: bad geometry");
  EMP_REQUIRE(variant >= 0 && (variant & 15) <= 3 && ((variant >> 4) & 15) <= 7, "conv1x1_dual: bad variant %d", variant);
  ConvParams p{};
  p.in = (const half_t*)d_in;
  p.in2 = (const half_t*)d_in2;
  p.wgt = (const half_t*)d_w;
  p.bias = d_bias;
  p.out = (half_t*)d_out;
  p.zero = (const half_t*)zero_page();
  EMP_REQUIRE(p.zero != nullptr, "conv1x1_dual: could not allocate the zero page");
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.in_ld = in_ld;
  p.H2 = H2; p.W2 = W2; p.Cin2 = Cin2; p.in2_ld = in2_ld; p.stride2 = stride2;
  p.Cout = Cout; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.dil = 1;
  p.Ho = H; p.Wo = W;
  p.out_ld = out_ld;
  p.act = relu;
  p.M = N * H * W;
  return launch_conv_igemm(p, variant, (hipStream_t)stream);
}

int emp_dwconv_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, const void* d_w, int K, void* d_out,
                        int out_ld, void* stream) {
  EMP_REQUIRE(d_in && d_w && d_out, "dwconv: null pointer");
  const half_t* zero = (const half_t*)zero_page();
  EMP_REQUIRE(zero != nullptr, "dwconv: could not allocate the zero page");
  return launch_dwconv((const half_t*)d_in, N, H, W, C, in_ld, (const half_t*)d_w, K, (half_t*)d_out, out_ld, zero,
                       (hipStream_t)stream);
}

int emp_sepconv5x5_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream) {
  EMP_REQUIRE(d_pw_w && d_packed, "sepconv5x5 pack: null pointer");
  return launch_sepconv5_pack_pw((const half_t*)d_pw_w, pw_ld, C, Cout, (half_t*)d_packed, (hipStream_t)stream);
}

int emp_sepconv5x5_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, const void* d_dw_w,
                            const void* d_pw_w, const float* d_bias, int Cout, int act, void* d_out,
                            int out_ld, const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out,
                            void* stream) {
  EMP_REQUIRE(d_in && d_dw_w && d_pw_w, "sepconv5x5: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0, "sepconv5x5: bad geometry");
  EMP_REQUIRE(head_c == 0 || (d_head_w && d_head_out), "sepconv5x5: head pointers missing");
  const half_t* zero = (const half_t*)zero_page();
  EMP_REQUIRE(zero != nullptr, "sepconv5x5: could not allocate the zero page");
  return launch_sepconv5((const half_t*)d_in, N, H, W, C, in_ld, (const half_t*)d_dw_w, (const half_t*)d_pw_w,
                         d_bias, Cout, act, head_c ? nullptr : (half_t*)d_out, out_ld, d_head_w, d_head_b, head_c,
                         d_head_out, (int64_t)H * W, zero, (hipStream_t)stream);
}

int emp_sepconvp_pack_dw(const void* d_dw_w, int K, int C, void* d_packed, void* stream) {
  EMP_REQUIRE(d_dw_w && d_packed, "sepconvp pack_dw: null pointer");
  return launch_sepconvp_pack_dw((const float*)d_dw_w, K, C, (float*)d_packed, (hipStream_t)stream);
}

int emp_sepconvp_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream) {
  EMP_REQUIRE(d_pw_w && d_packed, "sepconvp pack_pw: null pointer");
  return launch_sepconvp_pack_pw((const float*)d_pw_w, pw_ld, C, Cout, (half_t*)d_packed, (hipStream_t)stream);
}

int emp_sepconvp_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, int K, const void* d_dw_w,
                          const void* d_pw_w, const float* d_bias, int Cout, int act, void* d_out, int out_ld,
                          const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out, void* stream) {
  EMP_REQUIRE(d_in && d_dw_w && d_pw_w, "sepconvp: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0, "sepconvp: bad geometry");
  EMP_REQUIRE(head_c == 0 || (d_head_w && d_head_out), "sepconvp: head pointers missing");
  EMP_REQUIRE(head_c != 0 || d_out, "sepconvp: output pointer missing");
  const half_t* zero = (const half_t*)zero_page();
  EMP_REQUIRE(zero != nullptr, "sepconvp: could not allocate the zero page");
  return launch_sepconvp((const half_t*)d_in, N, H, W, C, in_ld, (const float*)d_dw_w, (const half_t*)d_pw_w, d_bias, Cout,
                         act, head_c ? nullptr : (half_t*)d_out, out_ld, d_head_w, d_head_b, head_c, d_head_out,
                         (int64_t)H * W, zero, (hipStream_t)stream, K);
}

int emp_sepconvp_ws_pack_pw(const void* d_pw_w, int pw_ld, int C, int Cout, void* d_packed, void* stream) {
  EMP_REQUIRE(d_pw_w && d_packed, "sepconvp pack_pw: null pointer");
  return launch_sepconvp_pack_pw((const float*)d_pw_w, pw_ld, C, Cout, (half_t*)d_packed, (hipStream_t)stream, 1);
}

int emp_sepconvp_ws_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, int K, const void* d_dw_w,
                             const void* d_pw_w, const float* d_bias, int Cout, int act, void* d_out, int out_ld,
                             const float* d_head_w, const float* d_head_b, int head_c, float* d_head_out, void* stream) {
  EMP_REQUIRE(d_in && d_dw_w && d_pw_w, "sepconvp: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0, "sepconvp: bad geometry");
  EMP_REQUIRE(head_c == 0 || (d_head_w && d_head_out), "sepconvp: head pointers missing");
  EMP_REQUIRE(head_c != 0 || d_out, "sepconvp: output pointer missing");
  const half_t* zero = (const half_t*)zero_page();
  EMP_REQUIRE(zero != nullptr, "sepconvp: could not allocate the zero page");
  return launch_sepconvp((const half_t*)d_in, N, H, W, C, in_ld, (const float*)d_dw_w, (const half_t*)d_pw_w, d_bias, Cout,
                         act, head_c ? nullptr : (half_t*)d_out, out_ld, d_head_w, d_head_b, head_c, d_head_out,
                         (int64_t)H * W, zero, (hipStream_t)stream, K, 1);
}

int emp_sepconv3x3_nhwc_f16(const void* d_in, int N, int H, int W, int C, int in_ld, const void* d_dw_w, const void* d_pw_w,
                            const float* d_bias, int Cout, int act, void* d_out, int out_ld, void* stream) {
  EMP_REQUIRE(d_in && d_dw_w && d_pw_w && d_out, "sepconv3x3: null pointer");
  EMP_REQUIRE(N > 0 && H > 0 && W > 0, "sepconv3x3: bad geometry");
  const half_t* zero = (const half_t*)zero_page();
  EMP_REQUIRE(zero != nullptr, "sepconv3x3: could not allocate the zero page");
  return launch_sepconv5((const half_t*)d_in, N, H, W, C, in_ld, (const half_t*)d_dw_w, (const half_t*)d_pw_w, d_bias, Cout,
                         act, (half_t*)d_out, out_ld, nullptr, nullptr, 0, nullptr, (int64_t)H * W, zero,
                         (hipStream_t)stream, 3);
}

}  // extern "C"
