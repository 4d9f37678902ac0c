// Shared declarations of libempanada_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/empanada_hip.h"

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace emp {

void set_error(const char* fmt, ...);

#define EMP_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      emp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return EMP_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

#define EMP_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      emp::set_error(__VA_ARGS__);      \
      return EMP_ERR_INVALID;           \
    }                                   \
  } while (0)

#define EMP_LAUNCH_CHECK() EMP_CHECK_HIP(hipGetLastError())

// A 256-byte device buffer of zeros (source for out-of-bounds LDS-DMA lanes).
const void* zero_page();

// Opt a kernel into more than 64 KiB of dynamic LDS, once per (device, kernel symbol): hipFuncSetAttribute's effect is per
// device, and a `static bool` at the call site is shared by every kernel pointer of the same TYPE that passes through a generic
// lambda (the three ACT instantiations of conv16x3s_kernel shared one flag; a second GPU in the process got none)
int ensure_dyn_lds(const void* kernel, int bytes);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------
// implicit-GEMM convolution (conv_igemm.hip)
// ---------------------------------------------------------------------------
struct ConvParams {
  const half_t* in;   // (N,H,W,in_ld)
  const half_t* wgt;  // (Cout, KH*KW, Cin)
  const float* bias;  // (Cout) or null
  const float* bias_n;  // (N,Cout) or null
  const half_t* res;  // (N,Ho,Wo,res_ld) or null
  half_t* out;        // (N,Ho,Wo,out_ld)
  const half_t* zero; // zero page
  int N, H, W, Cin, in_ld;
  int Cout, KH, KW, stride, pad, dil;
  int Ho, Wo, out_ld, res_ld;
  int act;         // 0 none, 1 ReLU, 2 SiLU
  int ps_cout;     // > 0: k2s2 transposed-conv pixel-shuffle store, Cout == 4*ps_cout
  int M;           // N*Ho*Wo
  int mt, nt;      // tiles along pixels / couts
  int mt_per_xcd;  // ceil(mt/8)
  int kgroup;      // channel slabs (64 ch) per K-walk group, set by launch_conv_igemm
  int ngroup;      // 256 x 256 tiles only (round 6): cout tiles an XCD keeps resident while it walks its pixel tiles; 0 = all of them (cout tile fastest)
  // optional second source, K-concatenated behind the first one (1x1 main conv only): a strided 1x1 conv of `in2`
  // summed into the same accumulators -- the bottleneck's downsample branch folded into conv3.
  // Weight rows are then [KH*KW*Cin | Cin2] long.
  const half_t* in2;  // (N,H2,W2,in2_ld) or null
  int Cin2, in2_ld, H2, W2, stride2;
  // optional second destination: couts [split, Cout) go to out2 (channel 0 of out2 = cout `split`); split % 8 == 0.
  // Two convolutions of the same input (the two decoders' low-level projections) share one pass over it.
  half_t* out2;
  int out2_ld, split;
  // optional third destination: couts [split3, Cout) go to out3 (then out2 gets [split, split3)); split3 % 8 == 0.
  // Three convolutions of one map -- the next bottleneck's conv1 and the two decoders' low-level projections of a
  // ResNet stage output -- share one pass over it.
  half_t* out3;
  int out3_ld, split3;
  // optional back-to-back 1x1 convolution of the finished tile (256x256 tile only, Cout == 256 so that a workgroup owns
  // all channels of its pixels): next_out[m][0..next_cout) = relu(next_w . out[m][0..Cout) + next_b) -- the conv1 of the
  // NEXT bottleneck block computed from the tile while it is still in LDS, so that launch and its read of the widest
  // map go away.  next_w (next_cout, Cout) fp16 row-major, next_cout == 64 (ResNet layer1).
  const half_t* next_w;
  const float* next_b;
  half_t* next_out;
  int next_cout, next_ld;
};

// variant: 0 = auto, 1 = register-staged, 2 = LDS-DMA staged
int launch_conv_igemm(const ConvParams& p, int variant, hipStream_t stream);
// the generic tile as a grouped convolution, blockIdx.y = group (conv_igemm_grouped.hip): p describes one group
int launch_conv_igemm_grouped(const ConvParams& p, int groups, int gs_in, long long gs_w, int gs_out, hipStream_t stream);
// 256x256 tile for the MFMA-bound layers (conv_igemm256.hip)
bool conv_igemm256_supported(const ConvParams& p);
bool conv_uses_256(const ConvParams& p);      // launch_conv_igemm's auto choice
int launch_conv_igemm256(ConvParams p, hipStream_t stream, int kg = 0, int mode = 0, bool wpacked = false);
// packed weight image of the 256 x 256 tile (conv_igemm256.hip): ConvParams::wgt = the image, launch with wpacked (variant bit 20
// of launch_conv_igemm); made once per layer for the default K walk
int conv256_pack_kgroup(int KT, int Cin);
int conv256_pack_weights(const half_t* w, half_t* out, int Cout, int KT, int Cin, int Cin2, hipStream_t stream);
bool conv_b2b_supported(const ConvParams& p);   // can the 256x256 tile run p with its next_* convolution fused in
// half tile, two workgroups per CU, for the short-K layers (conv_igemm256.hip)
bool conv_igemm_h256_supported(const ConvParams& p);
int launch_conv_igemm_h256(ConvParams p, hipStream_t stream, int kg = 0);
// 64 x 64 tile with a deep LDS-DMA ring for launches with few tiles (batch-1 latency; conv_igemm_s64.hip)
bool conv_igemm_s64_supported(const ConvParams& p);
int launch_conv_igemm_s64(ConvParams p, hipStream_t stream, int kg = 0);
// 64 -> 64 channel 3x3 with register-resident weights and an LDS halo tile (conv3x3c64.hip)
bool conv3x3_c64_supported(const ConvParams& p);
int launch_conv3x3_c64(const ConvParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------
// element-wise / stencil kernels (layers.hip)
// ---------------------------------------------------------------------------
// img: (N,vh,vw) raw or normalised; pixels outside (vh,vw) read as 0.0 (factor_pad + conv padding)
int launch_stem7x7(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                   const float* w /*49x64*/, const float* b /*64*/, half_t* out, hipStream_t s);
int launch_maxpool3x3s2(const half_t* in, int N, int H, int W, int C, half_t* out, hipStream_t s);
int launch_stem3x3s2_f16(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                         const float* w /*[9][C]*/, const float* b, int C, half_t* out, int out_ld, hipStream_t s);
int launch_gate_mul_f16(const half_t* x, int x_ld, half_t* g /* in: gate logits, out: x * sigmoid(g) */, int g_ld, int64_t rows, int C,
                        hipStream_t s);
// conv1 + bn1 + relu + maxpool on MFMA in one launch (stem.hip): img -> (N,H/4,W/4,64)
int launch_stem_pool(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                     const float* w, const float* b, half_t* out, hipStream_t s);
// the same launch with an fp32 conv tile and an fp32 output (the fp16x3 mode's stem): img -> (N,H/4,W/4,64) fp32
int launch_stem_pool_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                         const float* w, const float* b, float* out, hipStream_t s);
// depthwise KxK (K = 3 or 5), stride 1, pad K/2; weights fp16 [K*K][C]
int launch_dwconv(const half_t* in, int N, int H, int W, int C, int in_ld, const half_t* w, int K,
                  half_t* out, int out_ld, const half_t* zero, hipStream_t s);
// BiFPN fast-normalised fusion (bifpn.py:52-69,106-134): out = ca*resize(a) + cb*b (+ cc*c)
//   mode 0: a is at half resolution, nearest x2 up-sampling; mode 1: a is at double resolution, 3x3/2 max-pool
//   out_lo != nullptr: the result as an fp16 hi + lo pair (hi -> out, lo -> out_lo), rows out_ld apart (0: C)
int launch_fuse_combine(const half_t* a, const half_t* b, const half_t* c, float ca, float cb, float cc, int mode,
                        int N, int H, int W, int C, half_t* out, hipStream_t s, half_t* out_lo = nullptr, int out_ld = 0);
int launch_bilinear_ac(const half_t* in, int N, int h, int w, int C, int in_ld, half_t* out, int H, int W,
                       int out_ld, hipStream_t s);
int launch_avgpool(const half_t* in, int N, int HW, int C, int in_ld, float* out /*N x C*/, float* part,
                   hipStream_t s);
int avgpool_scratch_floats(int N, int C);
// out[n][co] = act(sum_k in[n][k] * w[co][k] + b[co]); fp32, tiny shapes
int launch_gemv(const float* in, int N, int K, const float* w, const float* b, int Cout, int relu,
                float* out, hipStream_t s);
// out[n][c][pix] = sum_k in[n][pix][k]*w[c][k] + b[c]; fp16 NHWC in -> fp32 NCHW planes out.
// if scatter_idx != null: pix index of row r is scatter_idx[n*P + r] into a plane of plane_size.
// fused depthwise 5x5 + pointwise (+ optional 1x1 head), sepconv.hip
bool sepconv5_supported(int C, int Cout, int head_c);
int launch_sepconv5_pack_pw(const half_t* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s);
int launch_sepconv5(const half_t* in, int N, int H, int W, int C, int in_ld, const half_t* dww, const half_t* pww_packed,
                    const float* bias, int Cout, int act, half_t* out, int out_ld, const float* head_w,
                    const float* head_b, int head_c, float* hout, int64_t plane, const half_t* zero, hipStream_t s,
                    int ks = 5);
// the same block with an exact depthwise half (sepconv_precise.hip): fp32 taps, depthwise result as an fp16 hi + lo pair
// (two MFMAs per product), fp16 pointwise weights
bool sepconvp_supported(int C, int Cout, int head_c);
// hi + lo pointwise weights (a third MFMA per product: w_lo * x_hi) for the 128-cout blocks
bool sepconvp_wsplit_supported(int Cout);
// pointwise weights (Cout, pw_ld) fp32 -> C * Cout fp16 in MFMA fragment order (wsplit: 2 * C * Cout, hi then lo)
int launch_sepconvp_pack_pw(const float* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s, int wsplit = 0);
// depthwise taps (ks*ks, C) fp32 -> chunk-major [C/64][ks*ks][64] fp32
int launch_sepconvp_pack_dw(const float* w, int ks, int C, float* packed, hipStream_t s);
int launch_sepconvp(const half_t* in, int N, int H, int W, int C, int in_ld, const float* dww_packed, const half_t* pww_packed,
                    const float* bias, int Cout, int act, half_t* out, int out_ld, const float* head_w,
                    const float* head_b, int head_c, float* hout, int64_t plane, const half_t* zero, hipStream_t s,
                    int ks = 5, int wsplit = 0);
int launch_head1x1(const half_t* in, int N, int P, int K, int in_ld, const float* w, const float* b, int C,
                   float* out, int64_t plane_size, const int32_t* scatter_idx, hipStream_t s);
int launch_bilinear_ac_f32_nchw(const float* in, int NC, int h, int w, float* out, int scale, hipStream_t s);
int launch_zero(void* p, size_t bytes, hipStream_t s);

// ---------------------------------------------------------------------------
// fp32 reference mode (ref32.hip): NHWC fp32 maps, fp32 weights [Cout][KH*KW][Cin16], exact fp32 MFMA
// ---------------------------------------------------------------------------
struct Conv32 {
  const float* in; int in_ld;
  const float* w; const float* bias; const float* bias_n;      // bias_n: (N, Cout) per-image bias or null
  const float* res; int res_ld;
  float* out; int out_ld;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, dil, Ho, Wo;
  int act;          // 0 none, 1 ReLU, 2 SiLU
  int ps_cout;      // > 0: k2s2 transposed-conv pixel-shuffle store, Cout == 4 * ps_cout
  int groups;       // > 1: grouped convolution -- Cout and Cin are PER GROUP (Cin padded to 16), weights [G * Cout][KH*KW][Cin]
  int cin_g;        // grouped: real input channels per group (the channel step from one group to the next); else 0
  int x3;           // 1: the fp16x3 mode -- the same convolution on the fp16 matrix pipe with split operands (conv16x3.hip)
  // x3 only, optional: a second source K-concatenated behind the first (the bottleneck's projection shortcut folded into
  // conv3, as ConvParams::in2): a 1x1 convolution of in2 (N, H2, W2, in2_ld) at stride2 summed into the same accumulators;
  // weight rows are then [KH*KW*Cin | Cin2] long (Cin2 % 16 == 0)
  const float* in2; int in2_ld, Cin2, H2, W2, stride2;
  // x3 only, optional: a fused 1x1 head (heads.py:14) -- the kernel does not store the activation map; instead every cout
  // tile writes, per pixel, the partial sums of head_c dot products of relu(out) with head_w[h][Cout] over its couts to
  // head_part[(cout tile * M + pixel) * head_c + h]; launch_head_finish_f32 adds the tiles in ascending order (+ bias)
  const float* head_w; float* head_part; int head_c;
  int x3_mt, x3_nt, x3_mtx;   // x3 only: pixel tiles, cout tiles, pixel tiles per XCD (set by launch_conv16x3)
  const float* zero;          // x3 only: 64 B of zeros (the source of out-of-range operand chunks), set by launch_conv16x3
  const uint32_t* wpair;      // x3 only, optional: the weights already split, one uint32 = fp16 hi | fp16 lo << 16, layout of `w`
  const half_t* wimg;         // x3 only, optional: the weights as the split-role kernel's LDS image (launch_x3_weight_image), fetched by LDS-DMA
  // x3 only (round 6): "hl32" maps -- rows of 2 * ld halfs, per block of 32 channels 32 hi halfs then 32 lo halfs (x = hi + lo,
  // hi = fp16(x), lo = fp16(x - hi): the operand split done once by the producer) -- the plane region's format (conv16x3p.hip)
  int in_fmt, out_fmt, res_fmt;      // 0: fp32 rows, 1: hl32 rows (in / out / res then point at halfs; ld stays the channel count)
  const half_t* wimgp;               // the weights as conv16x3p_kernel's packed image (launch_x3p_pack); needs in_fmt == 1
  float* out2; int out2_ld, split2;  // conv16x3p only, optional: couts [split2, Cout) go to out2 (format out_fmt), split2 % 256 == 0
  int x3p_kg;                        // conv16x3p only: the K-walk group wimgp was packed along (0: tap-major)
  // x3 only, optional (round 6, late): split-K for launches that leave most of the chip idle (a 3x3 ASPP branch of ONE 1024^2 tile is 64
  // workgroups walking K = 18 432).  kpart: scratch of kpart_bytes (>= X3_KPART_BYTES covers every launch the rule below splits);
  // the launcher picks ksplit = S in {2, 4, 8} workgroups per tile (split s on the XCDs = s mod S), each over x3_ksteps steps of 32 K, writing raw fp32 partial sums to
  // kpart[(s * M + m) * Cout + co]; ksplit_finish32 adds them in ascending s, then bias / bias_n / residual / activation.
  float* kpart; int64_t kpart_bytes; int ksplit, x3_ksteps;
};
constexpr int64_t X3_KPART_BYTES = 64ll << 20;      // S * workgroups <= 512 tiles of 128 x 128 fp32 (conv16x3s) / 256 tiles of 256 x 256 (conv16x3p)
int launch_conv32(const Conv32& p, hipStream_t s);
int launch_conv16x3(const Conv32& p, hipStream_t s);      // called by launch_conv32 when p.x3 (its checks have run)
int launch_split_pairs(const float* w, uint32_t* out, int64_t n, hipStream_t s);
int64_t x3_weight_image_halfs(int Cout, int K, int Cin);      // 0: the shape never reaches the kernel that reads images
int launch_x3_weight_image(const float* w, half_t* out, int Cout, int K, hipStream_t s);
// 256 x 256 tile over hl32 maps, both operands by LDS-DMA (conv16x3p.hip)
bool conv16x3p_supported(const Conv32& p);                    // shape only (Cout % 256, Cin % 32, >= 4 K steps, plain convolution)
int launch_conv16x3p(const Conv32& p, hipStream_t s);
int64_t x3p_image_halfs(int Cout, int K);                     // 2 * Cout * K, or 0 where the kernel cannot take the shape
int x3p_kgroup(int KT, int Cin);                              // the K-walk group the network packs and launches a layer with
int launch_x3p_pack(const float* w, half_t* out, int Cout, int K, hipStream_t s, int KT = 0, int Cin = 0, int kg = 0);      // KT == 0: linear (tap-major)
int launch_hl32_from_f32(const float* in, half_t* out, int64_t rows, int C, int in_ld, int out_ld, hipStream_t s);
int launch_hl32_to_f32(const half_t* in, float* out, int64_t rows, int C, int in_ld, int out_ld, hipStream_t s);
int launch_avgpool_hl32(const half_t* in, int N, int HW, int C, int in_ld, float* out, hipStream_t s);
// fused depthwise + pointwise block of the fp16x3 graph (sepconv_x3.hip): fp32 NHWC in / out, 3 fp16 MFMAs per product
bool sepconv_x3_supported(int C, int Cout, int head_c, int ks);
int64_t sepx3_pw_halfs(int C, int Cout);
int launch_sepx3_pack_pw(const float* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s);
int launch_sepx3_pack_dw(const float* w /*(ks*ks, ld)*/, int ks, int C, int ld, float* packed, hipStream_t s);
int launch_sepconv_x3(const float* in, int N, int H, int W, int C, int in_ld, const float* dww, const half_t* pww, const float* bias,
                      int Cout, int act, float* out, int out_ld, const float* head_w, const float* head_b, int head_c, float* hout,
                      int64_t plane, hipStream_t s, int ks = 5);
int conv16x3_cout_tiles(int Cout);      // cout tiles of a launch (the size of head_part's first dimension)
int launch_head_finish_f32(const float* part, int tiles, int N, int P, int C, const float* b, float* out, hipStream_t s);
int launch_stem3x3s2_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                         const float* w /*[9][C]*/, const float* b, int C, float* out, int out_ld, hipStream_t s);
int launch_gate_mul_f32(float* x, int x_ld, const float* g, int g_ld, int64_t rows, int C, hipStream_t s);
int launch_stem7x7_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                       const float* w /*49x64*/, const float* b /*64*/, float* out, hipStream_t s);
int launch_maxpool3x3s2_f32(const float* in, int N, int H, int W, int C, float* out, hipStream_t s);
int launch_dwconv_f32(const float* in, int N, int H, int W, int C, int in_ld, const float* w /*[K*K][C]*/, int K, float* out,
                      int out_ld, hipStream_t s);
int launch_bilinear_ac_f32_nhwc(const float* in, int N, int h, int w, int C, int in_ld, float* out, int H, int W, int out_ld,
                                hipStream_t s);
int launch_avgpool_f32(const float* in, int N, int HW, int C, int in_ld, float* out, hipStream_t s);
int launch_fuse_combine_f32(const float* a, const float* b, const float* c, float ca, float cb, float cc, int mode, int N,
                            int H, int W, int C, float* out, hipStream_t s);
int launch_head1x1_f32(const float* in, int N, int P, int K, int in_ld, const float* w, const float* b, int C, float* out,
                       int64_t plane_size, const int32_t* scatter_idx, hipStream_t s);
int launch_point_features_f32(const float* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse, int ncls,
                              const int32_t* idx, int P, int H2, int W2, float* x0, float* x1, int ld, hipStream_t s);

// ---------------------------------------------------------------------------
// PointRend (pointrend.hip)
// ---------------------------------------------------------------------------
size_t topk_work_bytes(int N, int64_t plane);
int launch_upsample2x_keys(const float* in, int N, int C, int h, int w, float* out, uint32_t* keys, hipStream_t s);
int launch_topk_smallest(const uint32_t* keys, int N, int64_t plane, int k, void* work, size_t work_bytes,
                         int32_t* idx_out, hipStream_t s);
int launch_point_features(const half_t* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse,
                          int ncls, const int32_t* idx, int P, int H2, int W2, half_t* x0, half_t* x1, int ld,
                          hipStream_t s);

// fused point head: sampling + fc layers + predictor + scatter in one launch (pointrend.hip)
bool pr_mlp_supported(int C, int ld, int ncls, int num_fc);
int launch_pr_mlp(const half_t* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse, int ncls,
                  const int32_t* idx, int P, int H2, int W2, const half_t* const* fc_w, const float* const* fc_b,
                  int num_fc, int ld, const float* pred_w, const float* pred_b, float* out, int64_t plane, hipStream_t s);

// ---------------------------------------------------------------------------
// post-processing (postprocess.hip)
// ---------------------------------------------------------------------------

}  // namespace emp
