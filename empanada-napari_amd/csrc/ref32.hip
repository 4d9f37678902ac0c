// fp32 reference mode of the network forward on gfx950 (round 4; EMP_PRECISION=fp32 / emp_pdl_set_precision).
//
// The reference runs this path in fp32 (empanada/inference/engines.py:248-255: the TorchScript model in eval, no autocast);
// the fp16 engine meets the north star's "within 1e-3" on the float heat-maps in rms, not in the max norm (DESIGN.md,
// parity).  This file is the other mode: every activation map fp32 in HBM, every weight fp32, every product on the exact
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32: an fmaf chain per output, 155 TFLOP/s measured on MI355X = 1/16 of the fp16
// rate) or the fp32 vector pipe.  Slow by design -- one generic implicit-GEMM kernel, no layer fusion, no tuned tiles -- it
// is the device-side fp32 comparator: heads within ~1e-5 of the oracle's fp32 forward at any size, in seconds.
//
// Layout: NHWC fp32 activations (row stride `ld`, channel slices like the fp16 engine: torch.cat is free), weights
// [Cout][KH*KW][Cin16] fp32 (Cin padded to 16 with zeros), bias fp32.
#include "common.h"

namespace emp {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int R_BM = 64, R_BN = 64, R_BK = 16, R_LD = R_BK + 1;

template <int ACT>
__device__ __forceinline__ float r_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

// One workgroup = 64 pixels x 64 couts, four waves of 32 x 32 (MFMA 32x32x2 fp32: a = pixel row lane % 32, k = lane / 32;
// b = cout column lane % 32; D register i <-> pixel row 8 (i / 4) + 4 (lane / 32) + i % 4, cout column lane % 32).
// K = KH * KW * Cin16 walked tap-major in steps of 16 channels through LDS.
template <int ACT>
__global__ void __launch_bounds__(256) conv32_kernel(const Conv32 p) {
  __shared__ float As[R_BM][R_LD];
  __shared__ float Bs[R_BN][R_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * R_BM, n0 = blockIdx.y * R_BN;
  // grouped convolution (RegNet's 3x3): blockIdx.z = group; Cout / Cin are PER GROUP, the group's input channels start
  // at g * cin_g (Cin >= cin_g is its padding to 16: the extra channels meet zero weights), its outputs at g * Cout
  const int g = blockIdx.z, gco = g * p.Cout;
  const float* gin = p.in + (size_t)g * p.cin_g;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.N * HoWo;
  const int K = p.KH * p.KW * p.Cin;
  // staging roles: thread -> (row, 4 consecutive k)
  const int srow = tid >> 2, skq = (tid & 3) * 4;
  const int am = m0 + srow;
  int an = 0, aoy = 0, aox = 0;
  const bool arow_ok = am < M;
  if (arow_ok) {
    an = am / HoWo;
    const int r = am - an * HoWo;
    aoy = r / p.Wo;
    aox = r - aoy * p.Wo;
  }
  const int bco = n0 + srow;
  const float* brow = bco < p.Cout ? p.w + (size_t)(gco + bco) * K : nullptr;
  const int wrow = (wave & 1) * 32, wcol = (wave >> 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += R_BK) {
    const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
    if (arow_ok) {
      const int iy = aoy * p.stride - p.pad + ky * p.dil, ix = aox * p.stride - p.pad + kx * p.dil;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        av = *reinterpret_cast<const float4*>(gin + (((size_t)an * p.H + iy) * p.W + ix) * p.in_ld + c0 + skq);
    }
    if (brow) bv = *reinterpret_cast<const float4*>(brow + k0 + skq);
    __syncthreads();      // the previous step's fragment reads are done
    As[srow][skq + 0] = av.x; As[srow][skq + 1] = av.y; As[srow][skq + 2] = av.z; As[srow][skq + 3] = av.w;
    Bs[srow][skq + 0] = bv.x; Bs[srow][skq + 1] = bv.y; Bs[srow][skq + 2] = bv.z; Bs[srow][skq + 3] = bv.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R_BK; k += 2) {
      const float a = As[wrow + (lane & 31)][k + (lane >> 5)];
      const float b = Bs[wcol + (lane & 31)][k + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // epilogue: bias (+ per-image bias) (+ residual), activation, store (plain NHWC slice or k2s2 pixel shuffle)
  const int col = n0 + wcol + (lane & 31);
  if (col >= p.Cout) return;
  const int co = gco + col;
  const float bias = p.bias ? p.bias[co] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + wrow + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
    if (m >= M) continue;
    const int n = m / HoWo;
    float v = acc[i] + bias;
    if (p.bias_n) v += p.bias_n[(size_t)n * p.Cout + co];      // (never with groups: launch_conv32)
    if (p.res) v += p.res[(size_t)m * p.res_ld + co];
    v = r_act<ACT>(v);
    size_t o;
    if (p.ps_cout == 0) {
      o = (size_t)m * p.out_ld + co;
    } else {      // ConvTranspose2d(k=2, s=2) as four sub-pixel 1x1 convs: (n,y,x,q*C+c) -> (n, 2y+dy, 2x+dx, c)
      const int q = co / p.ps_cout, c = co - q * p.ps_cout;
      const int r = m - n * HoWo;
      const int y = r / p.Wo, x = r - y * p.Wo;
      o = (((size_t)n * (2 * p.Ho) + 2 * y + (q >> 1)) * (2 * p.Wo) + 2 * x + (q & 1)) * p.out_ld + c;
    }
    p.out[o] = v;
  }
}

// thread = one output pixel x 4 channels; everything below is a byte mover on the fp32 vector pipe
inline int grid_for(int64_t total, int cap = 256 * 64) {
  int64_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ void __launch_bounds__(256) maxpool32_kernel(const float* __restrict__ in, int N, int H, int W, int C,
                                                        float* __restrict__ out, int64_t total) {
  const int CG = C >> 2, Ho = H >> 1, Wo = W >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const int n = (int)(q / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = oy * 2 + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = ox * 2 + dx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)n * H + iy) * W + ix) * C + cg * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + cg * 4) = m;
  }
}

// depthwise KxK, stride 1, pad K/2; taps fp32 [K*K][C]; the tap order (ky-major, kx-minor, fmaf chain from 0) of the fp16 engine
__global__ void __launch_bounds__(256) dwconv32_kernel(const float* __restrict__ in, int N, int H, int W, int C, int in_ld,
                                                       const float* __restrict__ w, int K, float* __restrict__ out,
                                                       int out_ld, int64_t total) {
  const int CG = C >> 2, P = K / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox = (int)(q % W); q /= W;
    const int oy = (int)(q % H);
    const int n = (int)(q / H);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy + ky - P;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ox + kx - P;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + (((size_t)n * H + iy) * W + ix) * in_ld + cg * 4);
        const float4 t = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 4);
        a.x = fmaf(v.x, t.x, a.x); a.y = fmaf(v.y, t.y, a.y); a.z = fmaf(v.z, t.z, a.z); a.w = fmaf(v.w, t.w, a.w);
      }
    }
    *reinterpret_cast<float4*>(out + (((size_t)n * H + oy) * W + ox) * out_ld + cg * 4) = a;
  }
}

// the same depthwise conv, one thread = 4 channels x a strip of 8 consecutive output pixels of one row: a tap row's 8 + K - 1
// inputs are loaded once and feed all eight outputs (7.5 float4 loads per output instead of 25 at K = 5).  Per output the
// fmaf chain is dwconv32_kernel's (ky-major, kx-minor, from 0); a tap outside the map contributes fmaf(0, t, a) = a.
template <int K>
__global__ void __launch_bounds__(256) dwconv32_strip_kernel(const float* __restrict__ in, int N, int H, int W, int C, int in_ld,
                                                             const float* __restrict__ w, float* __restrict__ out, int out_ld,
                                                             int64_t total) {
  constexpr int P = K / 2, S = 8;
  const int CG = C >> 2, WS = W / S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int xs = (int)(q % WS); q /= WS;
    const int oy = (int)(q % H);
    const int n = (int)(q / H);
    const int x0 = xs * S;
    float4 a[S];
#pragma unroll
    for (int j = 0; j < S; ++j) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy + ky - P;
      if (iy < 0 || iy >= H) continue;
      const float* row = in + ((size_t)n * H + iy) * W * in_ld + cg * 4;
      float4 v[S + K - 1];
#pragma unroll
      for (int j = 0; j < S + K - 1; ++j) {
        const int ix = x0 + j - P;
        v[j] = (ix >= 0 && ix < W) ? *reinterpret_cast<const float4*>(row + (size_t)ix * in_ld) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 t = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + cg * 4);
#pragma unroll
        for (int j = 0; j < S; ++j) {
          a[j].x = fmaf(v[j + kx].x, t.x, a[j].x); a[j].y = fmaf(v[j + kx].y, t.y, a[j].y);
          a[j].z = fmaf(v[j + kx].z, t.z, a[j].z); a[j].w = fmaf(v[j + kx].w, t.w, a[j].w);
        }
      }
    }
    float* o = out + (((size_t)n * H + oy) * W + x0) * out_ld + cg * 4;
#pragma unroll
    for (int j = 0; j < S; ++j) *reinterpret_cast<float4*>(o + (size_t)j * out_ld) = a[j];
  }
}

// bilinear, align_corners=True, NHWC fp32 into a channel slice of `out` (torch's upsample_bilinear2d arithmetic:
// hy * (hx v00 + lx v01) + ly * (hx v10 + lx v11))
__global__ void __launch_bounds__(256) bilinear32_kernel(const float* __restrict__ in, int N, int h, int w, int C, int in_ld,
                                                         float* __restrict__ out, int H, int W, int out_ld, float sy,
                                                         float sx, int64_t total) {
#pragma clang fp contract(off)
  const int CG = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox = (int)(q % W); q /= W;
    const int oy = (int)(q % H);
    const int n = (int)(q / H);
    const float fy = sy * (float)oy, fx = sx * (float)ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + (size_t)n * h * w * in_ld + cg * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * w + x0) * in_ld);
    const float4 v01 = *reinterpret_cast<const float4*>(b + ((size_t)y0 * w + x1) * in_ld);
    const float4 v10 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * w + x0) * in_ld);
    const float4 v11 = *reinterpret_cast<const float4*>(b + ((size_t)y1 * w + x1) * in_ld);
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<float4*>(out + (((size_t)n * H + oy) * W + ox) * out_ld + cg * 4) = o;
  }
}

// The same up-sampling for FOUR consecutive output columns per thread (round 6): the four outputs of a row share at most three input
// columns when W >= 4 w (align_corners: the input step per output column is (w - 1) / (W - 1) < 1/4, three steps < 0.75), so a thread
// loads each of them once -- 6
// 16-byte loads per 4 outputs instead of 16 (the one-output kernel sat at 0.95 texture-data busy in the fp16x3 step's PMC census).
// Same loads, same arithmetic per output: bit-identical to bilinear32_kernel.
__global__ void __launch_bounds__(256) bilinear32x4_kernel(const float* __restrict__ in, int N, int h, int w, int C, int in_ld,
                                                           float* __restrict__ out, int H, int W, int out_ld, float sy,
                                                           float sx, int64_t total) {
#pragma clang fp contract(off)
  const int CG = C >> 2, W4 = W >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox0 = (int)(q % W4) * 4; q /= W4;
    const int oy = (int)(q % H);
    const int n = (int)(q / H);
    const float fy = sy * (float)oy;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float* b = in + (size_t)n * h * w * in_ld + cg * 4;
    const int xa = (int)(sx * (float)ox0);      // the first output's left column; the others' lie in [xa, xa + 1], their right ones in [xa, xa + 2]
    float4 r0[3], r1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int xc = min(xa + c, w - 1);
      r0[c] = *reinterpret_cast<const float4*>(b + ((size_t)y0 * w + xc) * in_ld);
      r1[c] = *reinterpret_cast<const float4*>(b + ((size_t)y1 * w + xc) * in_ld);
    }
    float* op = out + (((size_t)n * H + oy) * W + ox0) * out_ld + cg * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float fx = sx * (float)(ox0 + j);
      const int x0 = (int)fx;
      const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx = fx - (float)x0, hx = 1.f - lx;
      const int a = x0 - xa, bb = x1 - xa;      // 0 .. 2
      const float4 v00 = a == 0 ? r0[0] : (a == 1 ? r0[1] : r0[2]), v01 = bb == 0 ? r0[0] : (bb == 1 ? r0[1] : r0[2]);
      const float4 v10 = a == 0 ? r1[0] : (a == 1 ? r1[1] : r1[2]), v11 = bb == 0 ? r1[0] : (bb == 1 ? r1[1] : r1[2]);
      float4 o;
      o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
      o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
      o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
      o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
      *reinterpret_cast<float4*>(op + (size_t)j * out_ld) = o;
    }
  }
}

// global average pool: one workgroup of 1024 threads per (image, 64 channels) -- lane = channel (a wave reads 256
// contiguous bytes per pixel), the 16 waves take the pixels round-robin with four independent partial sums each, the 64
// partials of a channel are added in a fixed order (deterministic).  Round 5: the one-thread-per-channel form of round 4
// walked 4096 pixels serially -- 0.49 ms per call at 8 x 64^2 x 2048, 2 % of an fp16x3 step.
__global__ void __launch_bounds__(1024) avgpool32_kernel(const float* __restrict__ in, int N, int HW, int C, int in_ld,
                                                         float* __restrict__ out) {
  __shared__ float part[16][64];
  const int cgs = (C + 63) / 64;
  const int n = blockIdx.x / cgs, c = (blockIdx.x % cgs) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* b = in + (size_t)n * HW * in_ld + c;
    int p = q;
    for (; p + 48 < HW; p += 64) {
      s0 += b[(size_t)p * in_ld]; s1 += b[(size_t)(p + 16) * in_ld]; s2 += b[(size_t)(p + 32) * in_ld]; s3 += b[(size_t)(p + 48) * in_ld];
    }
    for (; p < HW; p += 16) s0 += b[(size_t)p * in_ld];
  }
  part[q][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
    out[(size_t)n * C + c] = t / (float)HW;
  }
}

// BiFPN fast-normalised fusion (layers.hip fuse_combine_kernel) on fp32 maps
__global__ void __launch_bounds__(256) fuse_combine32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ c, float ca, float cb, float cc,
                                                             int mode, int N, int H, int W, int C, float* __restrict__ out,
                                                             int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % C);
    int64_t q = i / C;
    const int x = (int)(q % W); q /= W;
    const int y = (int)(q % H);
    const int n = (int)(q / H);
    float ra;
    if (mode == 0) {
      const int ah = H >> 1, aw = W >> 1;
      ra = a[(((size_t)n * ah + (y >> 1)) * aw + (x >> 1)) * C + ch];
    } else {
      const int ah = H * 2, aw = W * 2;
      ra = -INFINITY;
      for (int dy = -1; dy <= 1; ++dy) {
        const int iy = 2 * y + dy;
        if (iy < 0 || iy >= ah) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int ix = 2 * x + dx;
          if (ix < 0 || ix >= aw) continue;
          ra = fmaxf(ra, a[(((size_t)n * ah + iy) * aw + ix) * C + ch]);
        }
      }
    }
    float v = ca * ra + cb * b[i];
    if (c) v += cc * c[i];
    out[i] = v;
  }
}

// small-Cout 1x1 conv: fp32 NHWC rows -> fp32 NCHW planes (heads.py:14; PointRend predictor with scatter_idx)
__global__ void __launch_bounds__(256) head1x1_32_kernel(const float* __restrict__ in, int N, int P, int K, int in_ld,
                                                         const float* __restrict__ w, const float* __restrict__ b, int C,
                                                         float* __restrict__ out, int64_t plane,
                                                         const int32_t* __restrict__ scatter_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t rows = (int64_t)N * P;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const int n = (int)(r / P);
    const int64_t pix = scatter_idx ? (int64_t)scatter_idx[r] : r - (int64_t)n * P;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int k = lane; k < K; k += 64) s = fmaf(in[(size_t)r * in_ld + k], w[(size_t)c * K + k], s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) out[((size_t)n * C + c) * plane + pix] = s + b[c];
    }
  }
}

// PointRend point sampling (pointrend.hip point_features_kernel) on an fp32 feature map: rows [C features | ncls coarse | 0]
__global__ void __launch_bounds__(256) point_features32_kernel(const float* __restrict__ feat, int N, int fh, int fw, int C,
                                                               int feat_ld, const float* __restrict__ coarse, int ncls,
                                                               const int32_t* __restrict__ idx, int P, int H2, int W2,
                                                               float* __restrict__ x0, float* __restrict__ x1, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t pt = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (pt >= (int64_t)N * P) return;
  const int n = (int)(pt / P);
  const int id = idx[pt];
  const int iy = id / W2, ix = id - iy * W2;
  const float w_step = 1.0f / (float)W2, h_step = 1.0f / (float)H2;
  const float cx = 0.5f * w_step + w_step * (float)ix;
  const float cy = 0.5f * h_step + h_step * (float)iy;
  const float gx = 2.0f * cx - 1.0f, gy = 2.0f * cy - 1.0f;
  const float sx = ((gx + 1.f) * (float)fw - 1.f) * 0.5f;
  const float sy = ((gy + 1.f) * (float)fh - 1.f) * 0.5f;
  const float fx0 = floorf(sx), fy0 = floorf(sy);
  const int xa = (int)fx0, ya = (int)fy0, xb = xa + 1, yb = ya + 1;
  const float lx = sx - fx0, ly = sy - fy0;
  const float w00 = (1.f - lx) * (1.f - ly), w01 = lx * (1.f - ly), w10 = (1.f - lx) * ly, w11 = lx * ly;
  const bool ok00 = xa >= 0 && xa < fw && ya >= 0 && ya < fh;
  const bool ok01 = xb >= 0 && xb < fw && ya >= 0 && ya < fh;
  const bool ok10 = xa >= 0 && xa < fw && yb >= 0 && yb < fh;
  const bool ok11 = xb >= 0 && xb < fw && yb >= 0 && yb < fh;
  float* r0 = x0 + (size_t)pt * ld;
  float* r1 = x1 + (size_t)pt * ld;
  const float* fb = feat + (size_t)n * fh * fw * feat_ld;
  for (int c = lane; c < ld; c += 64) {
    float v = 0.f;
    if (c < C) {
      if (ok00) v = fmaf(fb[((size_t)ya * fw + xa) * feat_ld + c], w00, v);
      if (ok01) v = fmaf(fb[((size_t)ya * fw + xb) * feat_ld + c], w01, v);
      if (ok10) v = fmaf(fb[((size_t)yb * fw + xa) * feat_ld + c], w10, v);
      if (ok11) v = fmaf(fb[((size_t)yb * fw + xb) * feat_ld + c], w11, v);
      r0[c] = v;
    } else {
      if (c - C < ncls) {
        const float* cb = coarse + ((size_t)n * ncls + (c - C)) * fh * fw;
        if (ok00) v = fmaf(cb[ya * fw + xa], w00, v);
        if (ok01) v = fmaf(cb[ya * fw + xb], w01, v);
        if (ok10) v = fmaf(cb[yb * fw + xa], w10, v);
        if (ok11) v = fmaf(cb[yb * fw + xb], w11, v);
      }
      r0[c] = v;
      r1[c] = v;
    }
  }
}

// The same sampling with 16 lanes per point and four channels per lane (round 6, late): a wave gathers four points at once, every
// corner row as 16-byte loads (8 192 points per image whatever its size: at 512^2 slices this kernel was 4 % of the 3-D job).  Per
// channel the same fmaf chain over the corners 00, 01, 10, 11: bit-identical to point_features32_kernel.  Needs C % 4 == 0, ld % 4 == 0.
__global__ void __launch_bounds__(256) point_features32v_kernel(const float* __restrict__ feat, int N, int fh, int fw, int C,
                                                                int feat_ld, const float* __restrict__ coarse, int ncls,
                                                                const int32_t* __restrict__ idx, int P, int H2, int W2,
                                                                float* __restrict__ x0, float* __restrict__ x1, int ld) {
  const int l16 = threadIdx.x & 15;
  const int64_t pt = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  if (pt >= (int64_t)N * P) return;
  const int n = (int)(pt / P);
  const int id = idx[pt];
  const int iy = id / W2, ix = id - iy * W2;
  const float w_step = 1.0f / (float)W2, h_step = 1.0f / (float)H2;
  const float cx = 0.5f * w_step + w_step * (float)ix;
  const float cy = 0.5f * h_step + h_step * (float)iy;
  const float gx = 2.0f * cx - 1.0f, gy = 2.0f * cy - 1.0f;
  const float sx = ((gx + 1.f) * (float)fw - 1.f) * 0.5f;
  const float sy = ((gy + 1.f) * (float)fh - 1.f) * 0.5f;
  const float fx0 = floorf(sx), fy0 = floorf(sy);
  const int xa = (int)fx0, ya = (int)fy0, xb = xa + 1, yb = ya + 1;
  const float lx = sx - fx0, ly = sy - fy0;
  const float w00 = (1.f - lx) * (1.f - ly), w01 = lx * (1.f - ly), w10 = (1.f - lx) * ly, w11 = lx * ly;
  const bool ok00 = xa >= 0 && xa < fw && ya >= 0 && ya < fh;
  const bool ok01 = xb >= 0 && xb < fw && ya >= 0 && ya < fh;
  const bool ok10 = xa >= 0 && xa < fw && yb >= 0 && yb < fh;
  const bool ok11 = xb >= 0 && xb < fw && yb >= 0 && yb < fh;
  float* r0 = x0 + (size_t)pt * ld;
  float* r1 = x1 + (size_t)pt * ld;
  const float* fb = feat + (size_t)n * fh * fw * feat_ld;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = l16 * 4; c < ld; c += 64) {
    if (c < C) {
      const float4 a = ok00 ? *reinterpret_cast<const float4*>(fb + ((size_t)ya * fw + xa) * feat_ld + c) : z;
      const float4 b = ok01 ? *reinterpret_cast<const float4*>(fb + ((size_t)ya * fw + xb) * feat_ld + c) : z;
      const float4 d = ok10 ? *reinterpret_cast<const float4*>(fb + ((size_t)yb * fw + xa) * feat_ld + c) : z;
      const float4 e = ok11 ? *reinterpret_cast<const float4*>(fb + ((size_t)yb * fw + xb) * feat_ld + c) : z;
      float4 v = z;
      if (ok00) { v.x = fmaf(a.x, w00, v.x); v.y = fmaf(a.y, w00, v.y); v.z = fmaf(a.z, w00, v.z); v.w = fmaf(a.w, w00, v.w); }
      if (ok01) { v.x = fmaf(b.x, w01, v.x); v.y = fmaf(b.y, w01, v.y); v.z = fmaf(b.z, w01, v.z); v.w = fmaf(b.w, w01, v.w); }
      if (ok10) { v.x = fmaf(d.x, w10, v.x); v.y = fmaf(d.y, w10, v.y); v.z = fmaf(d.z, w10, v.z); v.w = fmaf(d.w, w10, v.w); }
      if (ok11) { v.x = fmaf(e.x, w11, v.x); v.y = fmaf(e.y, w11, v.y); v.z = fmaf(e.z, w11, v.z); v.w = fmaf(e.w, w11, v.w); }
      *reinterpret_cast<float4*>(r0 + c) = v;
    } else {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = c + q - C;
        if (k < ncls) {
          const float* cb = coarse + ((size_t)n * ncls + k) * fh * fw;
          float v = 0.f;
          if (ok00) v = fmaf(cb[ya * fw + xa], w00, v);
          if (ok01) v = fmaf(cb[ya * fw + xb], w01, v);
          if (ok10) v = fmaf(cb[yb * fw + xa], w10, v);
          if (ok11) v = fmaf(cb[yb * fw + xb], w11, v);
          o[q] = v;
        }
      }
      const float4 v = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(r0 + c) = v;
      *reinterpret_cast<float4*>(r1 + c) = v;
    }
  }
}

}  // namespace

// RegNet stem (regnet.py:38-49): 3x3 stride-2 conv of the single-channel image + folded BN + ReLU, normalisation and
// factor_pad fused as in the 7x7 stem; thread = one output pixel x 4 channels, weights [9][C]
template <typename T>
__global__ void __launch_bounds__(256) stem3x3s2_32_kernel(const T* __restrict__ img, float sub, float mul, int normalise, int N,
                                                           int H, int W, int vh, int vw, const float* __restrict__ w,
                                                           const float* __restrict__ b, int C, float* __restrict__ out,
                                                           int out_ld, int64_t total) {
  const int CG = C >> 2, Ho = H >> 1, Wo = W >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const int n = (int)(q / Ho);
    const T* src = img + (size_t)n * vh * vw;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        float v = 0.f;
        if (iy >= 0 && iy < vh && ix >= 0 && ix < vw) {      // outside the valid image: zero AFTER normalisation
          v = (float)src[(size_t)iy * vw + ix];
          if (normalise) { v -= sub; v *= mul; }
        }
        const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)(ky * 3 + kx) * C + cg * 4);
        acc.x = fmaf(v, wv.x, acc.x); acc.y = fmaf(v, wv.y, acc.y); acc.z = fmaf(v, wv.z, acc.z); acc.w = fmaf(v, wv.w, acc.w);
      }
    }
    const float4 bv = *reinterpret_cast<const float4*>(b + cg * 4);
    float4 o;
    o.x = fmaxf(acc.x + bv.x, 0.f); o.y = fmaxf(acc.y + bv.y, 0.f); o.z = fmaxf(acc.z + bv.z, 0.f); o.w = fmaxf(acc.w + bv.w, 0.f);
    *reinterpret_cast<float4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * out_ld + cg * 4) = o;
  }
}

// the reference's squeeze-excite gate (blocks.py:35-50; per pixel, its "pool" is 1 x 1): x *= sigmoid(g), in place
__global__ void __launch_bounds__(256) gate_mul32_kernel(float* __restrict__ x, int x_ld, const float* __restrict__ g, int g_ld, int C,
                                                         int64_t total) {
  const int CG = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    const int64_t m = i / CG;
    float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * x_ld + cg * 4);
    const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)m * g_ld + cg * 4);
    xv.x *= 1.f / (1.f + expf(-gv.x)); xv.y *= 1.f / (1.f + expf(-gv.y));
    xv.z *= 1.f / (1.f + expf(-gv.z)); xv.w *= 1.f / (1.f + expf(-gv.w));
    *reinterpret_cast<float4*>(x + (size_t)m * x_ld + cg * 4) = xv;
  }
}

int launch_stem3x3s2_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw, const float* w,
                         const float* b, int C, float* out, int out_ld, hipStream_t s) {
  EMP_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && out_ld % 4 == 0 && out_ld >= C, "stem3x3: bad shape");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  const dim3 grid(grid_for(total));
  switch (dtype) {
    case EMP_IMG_F32:
      hipLaunchKernelGGL(stem3x3s2_32_kernel<float>, grid, dim3(256), 0, s, (const float*)img, sub, mul, 0, N, H, W, vh, vw, w, b, C,
                         out, out_ld, total);
      break;
    case EMP_IMG_U8:
      hipLaunchKernelGGL(stem3x3s2_32_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)img, sub, mul, 1, N, H, W, vh, vw, w, b,
                         C, out, out_ld, total);
      break;
    case EMP_IMG_U16:
      hipLaunchKernelGGL(stem3x3s2_32_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)img, sub, mul, 1, N, H, W, vh, vw, w, b,
                         C, out, out_ld, total);
      break;
    default:
      EMP_REQUIRE(false, "stem3x3: unknown image dtype %d", dtype);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_gate_mul_f32(float* x, int x_ld, const float* g, int g_ld, int64_t rows, int C, hipStream_t s) {
  EMP_REQUIRE(C % 4 == 0 && x_ld % 4 == 0 && g_ld % 4 == 0 && x_ld >= C && g_ld >= C, "gate_mul32: bad shape");
  const int64_t total = rows * (C / 4);
  hipLaunchKernelGGL(gate_mul32_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, x_ld, g, g_ld, C, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_conv32(const Conv32& p, hipStream_t s) {
  EMP_REQUIRE(p.in && p.w && p.out, "conv32: null pointer");
  EMP_REQUIRE(p.Cin % R_BK == 0 && p.in_ld % 4 == 0 && p.in_ld >= p.Cin && ((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.w % 16) == 0,
              "conv32: Cin=%d must be a multiple of 16 (padded with zero weights), in_ld=%d a multiple of 4", p.Cin, p.in_ld);
  EMP_REQUIRE(p.ps_cout == 0 || (p.Cout == 4 * p.ps_cout && p.res == nullptr), "conv32: pixel-shuffle store needs Cout == 4 * ps_cout");
  EMP_REQUIRE(p.act >= 0 && p.act <= 2, "conv32: bad activation");
  const int G = p.groups > 1 ? p.groups : 1;
  EMP_REQUIRE(G == 1 || (p.bias_n == nullptr && p.ps_cout == 0 && p.cin_g > 0 && p.cin_g % 4 == 0 && p.cin_g <= p.Cin &&
                         (G - 1) * p.cin_g + p.Cin <= p.in_ld && G < 65536),
              "conv32: grouped form needs cin_g %% 4 == 0 and (G - 1) * cin_g + Cin <= in_ld (G=%d cin_g=%d Cin=%d in_ld=%d)", G, p.cin_g,
              p.Cin, p.in_ld);
  EMP_REQUIRE(G > 1 || p.cin_g == 0, "conv32: cin_g is the grouped form's channel step");
  const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
  EMP_REQUIRE(M > 0 && M < (1ll << 31), "conv32: bad problem size");
  if (p.x3 && p.in_fmt) return launch_conv16x3p(p, s);      // an hl32 input map: the plane region's 256 x 256 tile (conv16x3p.hip)
  if (p.x3) return launch_conv16x3(p, s);
  const dim3 grid((unsigned)((M + R_BM - 1) / R_BM), (unsigned)((p.Cout + R_BN - 1) / R_BN), (unsigned)G);
  if (p.act == 1) hipLaunchKernelGGL(conv32_kernel<1>, grid, dim3(256), 0, s, p);
  else if (p.act == 2) hipLaunchKernelGGL(conv32_kernel<2>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv32_kernel<0>, grid, dim3(256), 0, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_maxpool3x3s2_f32(const float* in, int N, int H, int W, int C, float* out, hipStream_t s) {
  EMP_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool32: bad shape");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, N, H, W, C, out, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_dwconv_f32(const float* in, int N, int H, int W, int C, int in_ld, const float* w, int K, float* out, int out_ld,
                      hipStream_t s) {
  EMP_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && (K == 3 || K == 5), "dwconv32: bad shape");
  const char* strip_env = getenv("EMP_DW32_STRIP");      // =0: the one-output-per-thread kernel everywhere (read per call: tests A/B both in one process)
  if (W % 8 == 0 && !(strip_env && strip_env[0] == '0')) {
    const int64_t strips = (int64_t)N * H * (W / 8) * (C / 4);
    if (K == 5) hipLaunchKernelGGL(dwconv32_strip_kernel<5>, dim3(grid_for(strips)), dim3(256), 0, s, in, N, H, W, C, in_ld, w, out, out_ld, strips);
    else hipLaunchKernelGGL(dwconv32_strip_kernel<3>, dim3(grid_for(strips)), dim3(256), 0, s, in, N, H, W, C, in_ld, w, out, out_ld, strips);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  const int64_t total = (int64_t)N * H * W * (C / 4);
  hipLaunchKernelGGL(dwconv32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, N, H, W, C, in_ld, w, K, out, out_ld, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_bilinear_ac_f32_nhwc(const float* in, int N, int h, int w, int C, int in_ld, float* out, int H, int W, int out_ld,
                                hipStream_t s) {
  EMP_REQUIRE(C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0, "bilinear32: channels must be multiples of 4");
  const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int64_t total = (int64_t)N * H * W * (C / 4);
  const char* x4_env = getenv("EMP_BILINEAR32_X4");      // =0: the one-output kernel (A/B; bit-identical)
  if (W % 4 == 0 && W >= 4 * w && !(x4_env && x4_env[0] == '0')) {      // up-sampling by >= 4 (3 * sx < 0.75): four output columns per thread
    hipLaunchKernelGGL(bilinear32x4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, s, in, N, h, w, C, in_ld, out, H, W, out_ld, sy, sx,
                       total / 4);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  hipLaunchKernelGGL(bilinear32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, N, h, w, C, in_ld, out, H, W, out_ld, sy, sx,
                     total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_avgpool_f32(const float* in, int N, int HW, int C, int in_ld, float* out, hipStream_t s) {
  hipLaunchKernelGGL(avgpool32_kernel, dim3(N * cdiv(C, 64)), dim3(1024), 0, s, in, N, HW, C, in_ld, out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_fuse_combine_f32(const float* a, const float* b, const float* c, float ca, float cb, float cc, int mode, int N,
                            int H, int W, int C, float* out, hipStream_t s) {
  EMP_REQUIRE(mode == 0 || mode == 1, "fuse_combine32: bad mode");
  const int64_t total = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(fuse_combine32_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, b, c, ca, cb, cc, mode, N, H, W, C, out,
                     total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_head1x1_f32(const float* in, int N, int P, int K, int in_ld, const float* w, const float* b, int C, float* out,
                       int64_t plane, const int32_t* scatter_idx, hipStream_t s) {
  const int64_t rows = (int64_t)N * P;
  hipLaunchKernelGGL(head1x1_32_kernel, dim3(grid_for(rows * 64)), dim3(256), 0, s, in, N, P, K, in_ld, w, b, C, out, plane,
                     scatter_idx);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_point_features_f32(const float* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse, int ncls,
                              const int32_t* idx, int P, int H2, int W2, float* x0, float* x1, int ld, hipStream_t s) {
  const int64_t pts = (int64_t)N * P;
  const char* v_env = getenv("EMP_PF32_VEC");      // =0: one wave per point, one channel per lane (A/B; bit-identical)
  if (C % 4 == 0 && ld % 4 == 0 && feat_ld % 4 == 0 && ((uintptr_t)feat % 16) == 0 && ((uintptr_t)x0 % 16) == 0 && ((uintptr_t)x1 % 16) == 0 &&
      !(v_env && v_env[0] == '0')) {
    hipLaunchKernelGGL(point_features32v_kernel, dim3((unsigned)((pts * 16 + 255) / 256)), dim3(256), 0, s, feat, N, fh, fw, C, feat_ld, coarse,
                       ncls, idx, P, H2, W2, x0, x1, ld);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  hipLaunchKernelGGL(point_features32_kernel, dim3((unsigned)((pts * 64 + 255) / 256)), dim3(256), 0, s, feat, N, fh, fw, C,
                     feat_ld, coarse, ncls, idx, P, H2, W2, x0, x1, ld);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
