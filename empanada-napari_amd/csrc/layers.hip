// Non-GEMM layers of the Panoptic-DeepLab forward for gfx950: stem conv
// (Cin = 1, fused normalisation + zero padding), 3x3/2 max-pool, 5x5 depthwise
// conv, bilinear (align_corners=True) resampling, global average pool, tiny
// fp32 GEMV (ASPP image-pooling branch) and the small-Cout 1x1 head conv.
// All of them are HBM/L2-bound byte movers: NHWC fp16, 16-byte accesses per
// lane, consecutive lanes on consecutive channel groups (coalesced).
#include "common.h"

namespace emp {
namespace {

// ---------------------------------------------------------------------------
// stem: 7x7 stride-2 pad-3 conv, 1 -> 64 channels, + bias + ReLU
// reference: encoders/resnet.py:164-167,219-221 (conv1+bn1+relu, BN folded)
// One block = 16x16 output pixels; thread = 16 channels x 4 pixels.
// ---------------------------------------------------------------------------
// O = half_t (the fp16 engine) or float (the fp32 reference mode: same fp32 sums, stored unrounded)
template <typename T, typename O = half_t>
__global__ void __launch_bounds__(256) stem7x7_kernel(const T* __restrict__ img, float sub, float mul, int N,
                                                      int H, int W, int vh, int vw, const float* __restrict__ wgt,
                                                      const float* __restrict__ bias, O* __restrict__ out,
                                                      int normalise) {
  constexpr int TO = 16, PI = TO * 2 + 5;  // 37
  __shared__ float patch[PI][PI + 1];
  __shared__ __attribute__((aligned(16))) float wl[49][64];
  const int Ho = H >> 1, Wo = W >> 1;
  const int tiles_x = (Wo + TO - 1) / TO;
  const int tiles_y = (Ho + TO - 1) / TO;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; b /= tiles_y;
  const int n = b;
  const int oy0 = ty * TO, ox0 = tx * TO;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  const int tid = threadIdx.x;
  const T* src = img + (size_t)n * vh * vw;
  for (int i = tid; i < PI * PI; i += 256) {
    int py = i / PI, px = i - py * PI;
    int iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (iy >= 0 && iy < vh && ix >= 0 && ix < vw) {
      v = (float)src[(size_t)iy * vw + ix];
      if (normalise) { v -= sub; v *= mul; }
    }
    patch[py][px] = v;
  }
  for (int i = tid; i < 49 * 64; i += 256) (&wl[0][0])[i] = wgt[i];
  __syncthreads();

  const int cg = tid & 3;    // 16-channel group
  const int pl = tid >> 2;   // 0..63
  // fp32 accumulate on channel pairs (v_pk_fma_f32: two FMAs per lane and instruction; the kernel is bound by
  // VALU issue: 49 taps x 64 channels per output pixel)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc2[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc2[k][c] = f32x2{0.f, 0.f};
  int py[4], px[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int idx = pl + 64 * k;
    py[k] = idx >> 4;
    px[k] = idx & 15;
  }
  for (int ky = 0; ky < 7; ++ky) {
    for (int kx = 0; kx < 7; ++kx) {
      f32x2 wv[8];
      const float4* wp = reinterpret_cast<const float4*>(&wl[ky * 7 + kx][cg * 16]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 t = wp[q];
        wv[2 * q] = f32x2{t.x, t.y};
        wv[2 * q + 1] = f32x2{t.z, t.w};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = patch[py[k] * 2 + ky][px[k] * 2 + kx];
        const f32x2 xx = f32x2{x, x};
#pragma unroll
        for (int c = 0; c < 8; ++c) acc2[k][c] = __builtin_elementwise_fma(xx, wv[c], acc2[k][c]);
      }
    }
  }
  float acc[4][16];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 8; ++c) { acc[k][2 * c] = acc2[k][c][0]; acc[k][2 * c + 1] = acc2[k][c][1]; }
  float bv[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) bv[c] = bias[cg * 16 + c];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int oy = oy0 + py[k], ox = ox0 + px[k];
    if (oy >= Ho || ox >= Wo) continue;
    if constexpr (sizeof(O) == 4) {
      float* dst = reinterpret_cast<float*>(out) + (((size_t)n * Ho + oy) * Wo + ox) * 64 + cg * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float a = acc[k][c] + bv[c];
        dst[c] = a > 0.f ? a : 0.f;
      }
    } else {
      f16x8 o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = acc[k][c] + bv[c];
        float d = acc[k][8 + c] + bv[8 + c];
        o0[c] = (half_t)(a > 0.f ? a : 0.f);
        o1[c] = (half_t)(d > 0.f ? d : 0.f);
      }
      half_t* dst = reinterpret_cast<half_t*>(out) + (((size_t)n * Ho + oy) * Wo + ox) * 64 + cg * 16;
      *reinterpret_cast<f16x8*>(dst) = o0;
      *reinterpret_cast<f16x8*>(dst + 8) = o1;
    }
  }
}

// ---------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max-pool (encoders/resnet.py:168,222)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const half_t* __restrict__ in, int N, int H, int W, int C,
                                                           half_t* __restrict__ out, int64_t total) {
  const int CG = C >> 3;
  const int Ho = H >> 1, Wo = W >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int cg = (int)(i % CG);
    int64_t p = i / CG;
    int ox = (int)(p % Wo); p /= Wo;
    int oy = (int)(p % Ho);
    int n = (int)(p / Ho);
    float m[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      int iy = oy * 2 + dy;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        int ix = ox * 2 + dx;
        if (ix < 0 || ix >= W) continue;
        f16x8 v = *reinterpret_cast<const f16x8*>(in + (((size_t)n * H + iy) * W + ix) * C + cg * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], (float)v[c]);
      }
    }
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (half_t)m[c];
    *reinterpret_cast<f16x8*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + cg * 8) = o;
  }
}

// ---------------------------------------------------------------------------
// KxK depthwise conv (K = 3 | 5), stride 1, pad K/2, no bias (blocks.py:24-29, first conv of
// SeparableConv2d).  HBM-bound byte mover: one block = 8x32 output pixels x 64 channels; the
// (8+K-1)x(32+K-1)x64 input halo tile is brought into LDS by LDS-DMA (pixel-major, 128 B per
// pixel; out-of-image pixels read the zero page), each thread then slides a K-wide window down
// two columns for its 4 channels with the KxKx4 taps held in registers (fp32 accumulate).
// weights fp16 [K*K][C].
// ---------------------------------------------------------------------------
constexpr int DW_TH = 8, DW_TW = 32;
template <int K>
__global__ void __launch_bounds__(256, 2) dwconv_kernel(const half_t* __restrict__ in, int N, int H, int W, int C,
                                                        int in_ld, const half_t* __restrict__ wgt,
                                                        half_t* __restrict__ out, int out_ld,
                                                        const half_t* __restrict__ zero, int tiles_x, int tiles_y) {
  constexpr int P = K / 2, DW_IH = DW_TH + K - 1, DW_IW = DW_TW + K - 1;
  constexpr int NPIX = DW_IH * DW_IW;
  constexpr int NINSTR = (NPIX + 7) / 8;     // wave-instructions of 8 pixels
  __shared__ __attribute__((aligned(1024))) char lds[NINSTR * 1024];
  int b = blockIdx.x;
  const int cb = b % (C >> 6); b /= (C >> 6);
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; b /= tiles_y;
  const int n = b;
  const int y0 = ty * DW_TH, x0 = tx * DW_TW;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const half_t* src = in + (size_t)n * H * W * in_ld + cb * 64 + (l & 7) * 8;
  for (int i = w; i < NINSTR; i += 4) {
    const int p = i * 8 + (l >> 3);
    const int py = p / DW_IW, px = p - py * DW_IW;
    const int iy = y0 + py - P, ix = x0 + px - P;
    const bool ok = p < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const half_t* g = ok ? src + ((size_t)iy * W + ix) * in_ld : zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
  }
  // thread = 4 channels (cq) x columns {colb, colb+16}; fp32 taps in registers
  const int cq = tid & 15, colb = tid >> 4;
  float wv[K * K][4];
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(wgt + (size_t)t * C + cb * 64 + cq * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[t][c] = (float)h[c];
  }
  __syncthreads();
  const bool odd = cq & 1;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int col = colb + 16 * pass;
    float acc[DW_TH][4];
#pragma unroll
    for (int y = 0; y < DW_TH; ++y)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[y][c] = 0.f;
    f16x4 nxt[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
      nxt[kx] = *reinterpret_cast<const f16x4*>(lds + (0 * DW_IW + col + kx) * 128 + cq * 8);
#pragma unroll
    for (int r = 0; r < DW_IH; ++r) {
      float xv[K][4];
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[kx][c] = (float)nxt[kx][c];
      if (r + 1 < DW_IH) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
          nxt[kx] = *reinterpret_cast<const f16x4*>(lds + ((r + 1) * DW_IW + col + kx) * 128 + cq * 8);
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int y = r - ky;
        if (y < 0 || y >= DW_TH) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[y][c] = fmaf(xv[kx][c], wv[ky * K + kx][c], acc[y][c]);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the live range of a row's taps inside its iteration
    }
    // pair lanes (cq, cq^1) so that each lane stores 16 B: even lane -> row y, odd lane -> row y+1
    const int ox = x0 + col;
#pragma unroll
    for (int y = 0; y < DW_TH; y += 2) {
      f16x4 ra, rb;  // my rows y and y+1
#pragma unroll
      for (int c = 0; c < 4; ++c) { ra[c] = (half_t)acc[y][c]; rb[c] = (half_t)acc[y + 1][c]; }
      const uint2 ua = *reinterpret_cast<uint2*>(&ra), ub = *reinterpret_cast<uint2*>(&rb);
      uint2 snd = odd ? ua : ub, rcv;
      rcv.x = __shfl_xor(snd.x, 1);
      rcv.y = __shfl_xor(snd.y, 1);
      uint4 o = odd ? uint4{rcv.x, rcv.y, ub.x, ub.y} : uint4{ua.x, ua.y, rcv.x, rcv.y};
      const int oy = y0 + y + (odd ? 1 : 0);
      if (ox < W && oy < H)
        *reinterpret_cast<uint4*>(out + (((size_t)n * H + oy) * W + ox) * out_ld + cb * 64 + (cq & ~1) * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------
// BiFPN fast-normalised fusion: out = ca*resize(a) + cb*b (+ cc*c), NHWC fp16, same C for all.
//   mode 0 (top-down, bifpn.py:60-67): a is (N,H/2,W/2,C), nearest x2 up-sampling (Resize2d 'up')
//   mode 1 (bottom-up, bifpn.py:119-131): a is (N,2H,2W,C), 3x3 stride-2 pad-1 max-pool (Resize2d 'down')
// ---------------------------------------------------------------------------
// out_lo != nullptr (round 4): the fused value leaves as an fp16 hi + lo PAIR -- hi to out, lo = fp16(v - hi) to out_lo, both
// with row stride out_ld -- so that the node's separable conv (sepconv_precise.hip, channels [hi | lo] with duplicated taps
// and pointwise weights: dw and pw are linear) sees the fp32 sum to ~2^-22 instead of its fp16 rounding.
__global__ void __launch_bounds__(256) fuse_combine_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                           const half_t* __restrict__ c, float ca, float cb, float cc,
                                                           int mode, int N, int H, int W, int C,
                                                           half_t* __restrict__ out, half_t* __restrict__ out_lo, int out_ld,
                                                           int64_t total) {
  const int CG = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t p = i / CG;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    float ra[8];
    if (mode == 0) {
      const int ah = H >> 1, aw = W >> 1;
      const f16x8 v = *reinterpret_cast<const f16x8*>(a + (((size_t)n * ah + (y >> 1)) * aw + (x >> 1)) * C + cg * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) ra[k] = (float)v[k];
    } else {
      const int ah = H * 2, aw = W * 2;
#pragma unroll
      for (int k = 0; k < 8; ++k) ra[k] = -INFINITY;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int iy = 2 * y + dy;
        if (iy < 0 || iy >= ah) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int ix = 2 * x + dx;
          if (ix < 0 || ix >= aw) continue;
          const f16x8 v = *reinterpret_cast<const f16x8*>(a + (((size_t)n * ah + iy) * aw + ix) * C + cg * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) ra[k] = fmaxf(ra[k], (float)v[k]);
        }
      }
    }
    const size_t o = (((size_t)n * H + y) * W + x) * C + cg * 8;
    const size_t oo = (((size_t)n * H + y) * W + x) * out_ld + cg * 8;
    const f16x8 vb = *reinterpret_cast<const f16x8*>(b + o);
    float v[8];
    if (c) {
      const f16x8 vc = *reinterpret_cast<const f16x8*>(c + o);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ca * ra[k] + cb * (float)vb[k] + cc * (float)vc[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ca * ra[k] + cb * (float)vb[k];
    }
    f16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (half_t)v[k];
    *reinterpret_cast<f16x8*>(out + oo) = r;
    if (out_lo) {
      f16x8 l;
#pragma unroll
      for (int k = 0; k < 8; ++k) l[k] = (half_t)(v[k] - (float)r[k]);
      *reinterpret_cast<f16x8*>(out_lo + oo) = l;
    }
  }
}

// ---------------------------------------------------------------------------
// bilinear resize, align_corners=True, NHWC fp16 (decoders/panoptic_deeplab.py:75)
// ---------------------------------------------------------------------------
// one bilinear sample, explicit fma order (shared by both kernels below so that they agree bit for bit whatever the
// compiler's contraction choices): hy * (hx*v00 + lx*v01) + ly * (hx*v10 + lx*v11)
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, float hx, float lx, float hy, float ly) {
  const float t0 = __builtin_fmaf(lx, v01, hx * v00);
  const float t1 = __builtin_fmaf(lx, v11, hx * v10);
  return __builtin_fmaf(ly, t1, hy * t0);
}

// One block iteration = one output row segment: 256 threads = (256 / CG) pixels x CG 8-channel groups, so the
// index arithmetic is 32-bit and mostly wave-uniform and every pixel's channels are one contiguous store.
__global__ void __launch_bounds__(256) bilinear_ac_kernel(const half_t* __restrict__ in, int N, int h, int w, int C,
                                                          int in_ld, half_t* __restrict__ out, int H, int W,
                                                          int out_ld, float sy, float sx, int segs_per_row, int ppb) {
#pragma clang fp contract(off)      // coordinates and weights round like the oracle's; the sample itself is bilerp's explicit fmas
  const int CG = C >> 3;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;      // channel group, pixel within the segment
  const int total = N * H * segs_per_row;
  constexpr int U = 4;                               // row segments in flight per thread (16 loads before any store)
  for (int b0 = blockIdx.x * U; b0 < total; b0 += gridDim.x * U) {
    f16x8 v00[U], v01[U], v10[U], v11[U];
    float ly[U], lx[U];
    size_t oidx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u;
      const int seg = b % segs_per_row;
      const int row = b / segs_per_row;
      const int oy = row % H, n = row / H;
      const int ox = seg * ppb + pl;
      ok[u] = b < total && pl < ppb && ox < W;
      const float fy = sy * (float)oy, fx = sx * (float)ox;
      int y0 = (int)fy, x0 = (int)fx;
      if (!ok[u]) { y0 = 0; x0 = 0; }
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      ly[u] = fy - (float)y0;
      lx[u] = fx - (float)x0;
      const half_t* base = in + (size_t)(ok[u] ? n : 0) * h * w * in_ld + cg * 8;
      v00[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)y0 * w + x0) * in_ld);
      v01[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)y0 * w + x1) * in_ld);
      v10[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)y1 * w + x0) * in_ld);
      v11[u] = *reinterpret_cast<const f16x8*>(base + ((size_t)y1 * w + x1) * in_ld);
      oidx[u] = (((size_t)n * H + oy) * W + ox) * out_ld + cg * 8;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float hy = 1.f - ly[u], hx = 1.f - lx[u];
      f16x8 o;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        o[c] = (half_t)bilerp((float)v00[u][c], (float)v01[u][c], (float)v10[u][c], (float)v11[u][c], hx, lx[u], hy, ly[u]);
      }
      if (ok[u]) *reinterpret_cast<f16x8*>(out + oidx[u]) = o;
    }
  }
}

// Up-scaling by >= 3 (the decoder's 64^2 -> 256^2): a thread owns FOUR consecutive output pixels of a row.  Their
// source columns are x0a .. x0a + 2 (3 * sx < 1), so 6 loads feed 4 stores instead of 16 -- the one-pixel kernel moved
// 5x the output bytes through the CU's 64 B/clk vector-memory path [0.74 ms for 2.1 GB written].  Per output the
// arithmetic is the expression of bilinear_ac_kernel on the same source values: bit-identical results.
__global__ void __launch_bounds__(256) bilinear_ac_up4_kernel(const half_t* __restrict__ in, int N, int h, int w, int C,
                                                              int in_ld, half_t* __restrict__ out, int H, int W,
                                                              int out_ld, float sy, float sx, int segs_per_row, int ppb) {
#pragma clang fp contract(off)      // coordinates and weights round like the oracle's; the sample itself is bilerp's explicit fmas
  const int CG = C >> 3;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;      // channel group, 4-pixel group within the segment
  const int total = N * H * segs_per_row;
  constexpr int U = 2;                               // output rows in flight per thread (12 loads before any store)
  for (int b0 = blockIdx.x * U; b0 < total; b0 += gridDim.x * U) {
    f16x8 s0[U][3], s1[U][3];
    float ly[U];
    int oxs[U], xa[U];
    size_t obase[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int b = b0 + u;
      const int seg = b % segs_per_row;
      const int row = b / segs_per_row;
      const int oy = row % H, n = row / H;
      const int ox = (seg * ppb + pl) * 4;
      ok[u] = b < total && pl < ppb && ox < W;
      const float fy = sy * (float)oy;
      int y0 = (int)fy, x0 = (int)(sx * (float)ox);
      if (!ok[u]) { y0 = 0; x0 = 0; }
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
      ly[u] = fy - (float)y0;
      oxs[u] = ox;
      xa[u] = x0;
      const half_t* base = in + (size_t)(ok[u] ? n : 0) * h * w * in_ld + cg * 8;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int xc = x0 + k < w - 1 ? x0 + k : w - 1;
        s0[u][k] = *reinterpret_cast<const f16x8*>(base + ((size_t)y0 * w + xc) * in_ld);
        s1[u][k] = *reinterpret_cast<const f16x8*>(base + ((size_t)y1 * w + xc) * in_ld);
      }
      obase[u] = (((size_t)n * H + oy) * W + ox) * out_ld + cg * 8;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float hy = 1.f - ly[u];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = oxs[u] + j;
        const float fx = sx * (float)ox;
        const int x0 = (int)fx;
        const float lx = fx - (float)x0, hx = 1.f - lx;
        const bool d = x0 != xa[u];                   // x0 - xa in {0, 1}
        const bool edge = !(x0 < w - 1);              // x1 == x0
        const f16x8 a0 = d ? s0[u][1] : s0[u][0], a1 = edge ? a0 : (d ? s0[u][2] : s0[u][1]);
        const f16x8 c0 = d ? s1[u][1] : s1[u][0], c1 = edge ? c0 : (d ? s1[u][2] : s1[u][1]);
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          o[c] = (half_t)bilerp((float)a0[c], (float)a1[c], (float)c0[c], (float)c1[c], hx, lx, hy, ly[u]);
        }
        if (ok[u] && ox < W) *reinterpret_cast<f16x8*>(out + obase[u] + (size_t)j * out_ld) = o;
      }
    }
  }
}

// fp32 NCHW planes, integer scale, align_corners=True (Interpolate2d(4), panoptic_deeplab.py:89)
__global__ void __launch_bounds__(256) bilinear_ac_f32_kernel(const float* __restrict__ in, int NC, int h, int w,
                                                              float* __restrict__ out, int H, int W, float sy,
                                                              float sx, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int ox = (int)(i % W);
    int64_t p = i / W;
    int oy = (int)(p % H);
    int nc = (int)(p / H);
    float fy = sy * (float)oy, fx = sx * (float)ox;
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    float ly = fy - (float)y0, lx = fx - (float)x0;
    float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + (size_t)nc * h * w;
    out[i] = hy * (hx * b[y0 * w + x0] + lx * b[y0 * w + x1]) + ly * (hx * b[y1 * w + x0] + lx * b[y1 * w + x1]);
  }
}

// ---------------------------------------------------------------------------
// global average pool (decoders/aspp.py:33 AdaptiveAvgPool2d(1)): two stages,
// fixed summation order (deterministic).
// stage 1: grid (N, C/256, SEG): partial sums over a pixel segment
// stage 2: sum SEG partials, scale by 1/HW
// ---------------------------------------------------------------------------
constexpr int AVG_SEG = 16;
__global__ void __launch_bounds__(256) avgpool_partial_kernel(const half_t* __restrict__ in, int HW, int C, int in_ld,
                                                              float* __restrict__ part) {
  const int n = blockIdx.x, cb = blockIdx.y, seg = blockIdx.z;
  const int tid = threadIdx.x;
  const int cg = tid & 31, pr = tid >> 5;  // 32 channel groups x 8 pixel lanes
  const int c0 = cb * 256 + cg * 8;
  const int per = (HW + AVG_SEG - 1) / AVG_SEG;
  const int p0 = seg * per, p1 = min(HW, p0 + per);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    const half_t* base = in + (size_t)n * HW * in_ld + c0;
    for (int p = p0 + pr; p < p1; p += 8) {
      f16x8 v = *reinterpret_cast<const f16x8*>(base + (size_t)p * in_ld);
#pragma unroll
      for (int c = 0; c < 8; ++c) s[c] += (float)v[c];
    }
  }
  __shared__ float red[8][256 + 8];
#pragma unroll
  for (int c = 0; c < 8; ++c) red[pr][cg * 8 + c] = s[c];
  __syncthreads();
  if (tid < 256) {
    int c = cb * 256 + tid;
    if (c < C) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[r][tid];
      part[((size_t)n * AVG_SEG + seg) * C + c] = t;
    }
  }
}
__global__ void avgpool_final_kernel(const float* __restrict__ part, int C, float inv, float* __restrict__ out,
                                     int total) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int n = i / C, c = i - n * C;
  float t = 0.f;
  for (int s = 0; s < AVG_SEG; ++s) t += part[((size_t)n * AVG_SEG + s) * C + c];
  out[i] = t * inv;
}

// one wave per output element
__global__ void __launch_bounds__(256) gemv_kernel(const float* __restrict__ in, int N, int K,
                                                   const float* __restrict__ w, const float* __restrict__ b, int Cout,
                                                   int relu, float* __restrict__ out) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= N * Cout) return;
  const int n = wave / Cout, co = wave - n * Cout;
  const float* x = in + (size_t)n * K;
  const float* wr = w + (size_t)co * K;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s = fmaf(x[k], wr[k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    if (b) s += b[co];
    if (relu) s = s > 0.f ? s : 0.f;
    out[wave] = s;
  }
}

// ---------------------------------------------------------------------------
// small-Cout 1x1 conv with bias, fp16 NHWC rows in -> fp32 NCHW planes out
// (heads.py:14 final nn.Conv2d(nin, n_classes, 1); PointRend predictor
// point_rend.py:162).  Half a wave (32 lanes x 8 channels) per row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) head1x1_kernel(const half_t* __restrict__ in, int N, int P, int K, int in_ld,
                                                      const float* __restrict__ w, const float* __restrict__ b, int C,
                                                      float* __restrict__ out, int64_t plane,
                                                      const int32_t* __restrict__ scatter_idx) {
  const int lane = threadIdx.x & 63;
  const int hl = lane & 31;
  const int half_id = ((blockIdx.x * 256 + threadIdx.x) >> 5);
  const int nhalf = (gridDim.x * 256) >> 5;
  const int64_t rows = (int64_t)N * P;
  const int KC = K >> 3;
  for (int64_t r = half_id; r < rows; r += nhalf) {
    float xv[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int kc = hl + 32 * j;
      if (kc < KC) {
        f16x8 v = *reinterpret_cast<const f16x8*>(in + (size_t)r * in_ld + kc * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[j][c] = (float)v[c];
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[j][c] = 0.f;
      }
    }
    const int n = (int)(r / P);
    const int pr = (int)(r - (int64_t)n * P);
    const int64_t pix = scatter_idx ? (int64_t)scatter_idx[r] : (int64_t)pr;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int kc = hl + 32 * j;
        if (kc < KC) {
          const float4* wp = reinterpret_cast<const float4*>(w + (size_t)c * K + kc * 8);
          float4 w0 = wp[0], w1 = wp[1];
          s = fmaf(xv[j][0], w0.x, s); s = fmaf(xv[j][1], w0.y, s);
          s = fmaf(xv[j][2], w0.z, s); s = fmaf(xv[j][3], w0.w, s);
          s = fmaf(xv[j][4], w1.x, s); s = fmaf(xv[j][5], w1.y, s);
          s = fmaf(xv[j][6], w1.z, s); s = fmaf(xv[j][7], w1.w, s);
        }
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (hl == 0) out[((size_t)n * C + c) * plane + pix] = s + b[c];
    }
  }
}

__global__ void zero_kernel(uint4* p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
    p[i] = uint4{0, 0, 0, 0};
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 256 * 16) {
  int64_t g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// RegNet stem for the fp16 engine (regnet.py:38-49): 3x3 stride-2 conv of the single-channel image + folded BN + ReLU in
// fp32 (the arithmetic of ref32.hip's stem3x3s2_32_kernel), stored as fp16; thread = one output pixel x 8 channels
template <typename T>
__global__ void __launch_bounds__(256) stem3x3s2_f16_kernel(const T* __restrict__ img, float sub, float mul, int normalise, int N,
                                                            int H, int W, int vh, int vw, const float* __restrict__ w,
                                                            const float* __restrict__ b, int C, half_t* __restrict__ out,
                                                            int out_ld, int64_t total) {
  const int CG = C >> 3, Ho = H >> 1, Wo = W >> 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    int64_t q = i / CG;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho);
    const int n = (int)(q / Ho);
    const T* src = img + (size_t)n * vh * vw;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        float v = 0.f;
        if (iy >= 0 && iy < vh && ix >= 0 && ix < vw) {
          v = (float)src[(size_t)iy * vw + ix];
          if (normalise) { v -= sub; v *= mul; }
        }
        const float* wp = w + (size_t)(ky * 3 + kx) * C + cg * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wp[c], acc[c]);
      }
    }
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (half_t)fmaxf(acc[c] + b[cg * 8 + c], 0.f);
    *reinterpret_cast<f16x8*>(out + (((size_t)n * Ho + oy) * Wo + ox) * out_ld + cg * 8) = o;
  }
}

// the reference's per-pixel squeeze-excite gate (blocks.py:35-50) on fp16 maps: g = fp16(x * sigmoid(g)), fp32 inside.
// The gated map replaces the GATE, not x: the block's pre-gate map stays intact (teacher-forced layer checks read it)
__global__ void __launch_bounds__(256) gate_mul_f16_kernel(const half_t* __restrict__ x, int x_ld, half_t* __restrict__ g,
                                                           int g_ld, int C, int64_t total) {
  const int CG = C >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    const int64_t m = i / CG;
    const f16x8 xv = *reinterpret_cast<const f16x8*>(x + (size_t)m * x_ld + cg * 8);
    f16x8 gv = *reinterpret_cast<const f16x8*>(g + (size_t)m * g_ld + cg * 8);
#pragma unroll
    for (int c = 0; c < 8; ++c) gv[c] = (half_t)((float)xv[c] * (1.f / (1.f + expf(-(float)gv[c]))));
    *reinterpret_cast<f16x8*>(g + (size_t)m * g_ld + cg * 8) = gv;
  }
}

}  // namespace

int launch_stem7x7(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                   const float* w, const float* b, half_t* out, hipStream_t s) {
  EMP_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H, W must be even");
  const int Ho = H / 2, Wo = W / 2;
  const int grid = N * cdiv(Ho, 16) * cdiv(Wo, 16);
  switch (dtype) {
    case EMP_IMG_F32:
      hipLaunchKernelGGL(stem7x7_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)img, sub, mul, N, H, W, vh,
                         vw, w, b, out, 0);
      break;
    case EMP_IMG_U8:
      hipLaunchKernelGGL(stem7x7_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t*)img, sub, mul, N, H, W,
                         vh, vw, w, b, out, 1);
      break;
    case EMP_IMG_U16:
      hipLaunchKernelGGL(stem7x7_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t*)img, sub, mul, N, H,
                         W, vh, vw, w, b, out, 1);
      break;
    default:
      EMP_REQUIRE(false, "stem: unknown image dtype %d", dtype);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_stem7x7_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                       const float* w, const float* b, float* out, hipStream_t s) {
  EMP_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H, W must be even");
  const int Ho = H / 2, Wo = W / 2;
  const int grid = N * cdiv(Ho, 16) * cdiv(Wo, 16);
  switch (dtype) {
    case EMP_IMG_F32:
      hipLaunchKernelGGL((stem7x7_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)img, sub, mul, N, H, W,
                         vh, vw, w, b, out, 0);
      break;
    case EMP_IMG_U8:
      hipLaunchKernelGGL((stem7x7_kernel<uint8_t, float>), dim3(grid), dim3(256), 0, s, (const uint8_t*)img, sub, mul, N, H,
                         W, vh, vw, w, b, out, 1);
      break;
    case EMP_IMG_U16:
      hipLaunchKernelGGL((stem7x7_kernel<uint16_t, float>), dim3(grid), dim3(256), 0, s, (const uint16_t*)img, sub, mul, N,
                         H, W, vh, vw, w, b, out, 1);
      break;
    default:
      EMP_REQUIRE(false, "stem: unknown image dtype %d", dtype);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_stem3x3s2_f16(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw, const float* w,
                         const float* b, int C, half_t* out, int out_ld, hipStream_t s) {
  EMP_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 8 == 0 && out_ld % 8 == 0 && out_ld >= C, "stem3x3 (fp16): bad shape");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  const dim3 grid(grid_for(total));
  switch (dtype) {
    case EMP_IMG_F32:
      hipLaunchKernelGGL(stem3x3s2_f16_kernel<float>, grid, dim3(256), 0, s, (const float*)img, sub, mul, 0, N, H, W, vh, vw, w, b, C,
                         out, out_ld, total);
      break;
    case EMP_IMG_U8:
      hipLaunchKernelGGL(stem3x3s2_f16_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)img, sub, mul, 1, N, H, W, vh, vw, w,
                         b, C, out, out_ld, total);
      break;
    case EMP_IMG_U16:
      hipLaunchKernelGGL(stem3x3s2_f16_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)img, sub, mul, 1, N, H, W, vh, vw, w,
                         b, C, out, out_ld, total);
      break;
    default:
      EMP_REQUIRE(false, "stem3x3 (fp16): unknown image dtype %d", dtype);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_gate_mul_f16(const half_t* x, int x_ld, half_t* g, int g_ld, int64_t rows, int C, hipStream_t s) {
  EMP_REQUIRE(C % 8 == 0 && x_ld % 8 == 0 && g_ld % 8 == 0 && x_ld >= C && g_ld >= C, "gate_mul (fp16): bad shape");
  const int64_t total = rows * (C / 8);
  hipLaunchKernelGGL(gate_mul_f16_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, x_ld, g, g_ld, C, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_maxpool3x3s2(const half_t* in, int N, int H, int W, int C, half_t* out, hipStream_t s) {
  EMP_REQUIRE(C % 8 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool: bad shape");
  int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, N, H, W, C, out, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_dwconv(const half_t* in, int N, int H, int W, int C, int in_ld, const half_t* w, int K, half_t* out,
                  int out_ld, const half_t* zero, hipStream_t s) {
  EMP_REQUIRE(C % 64 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0, "dwconv: C=%d must be a multiple of 64", C);
  EMP_REQUIRE(K == 3 || K == 5, "dwconv: kernel size %d unsupported (3 or 5)", K);
  const int tiles_x = cdiv(W, DW_TW), tiles_y = cdiv(H, DW_TH);
  const int64_t grid = (int64_t)N * tiles_x * tiles_y * (C / 64);
  EMP_REQUIRE(grid < (1ll << 31), "dwconv: grid too large");
  if (K == 5)
    hipLaunchKernelGGL(dwconv_kernel<5>, dim3((unsigned)grid), dim3(256), 0, s, in, N, H, W, C, in_ld, w, out, out_ld,
                       zero, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL(dwconv_kernel<3>, dim3((unsigned)grid), dim3(256), 0, s, in, N, H, W, C, in_ld, w, out, out_ld,
                       zero, tiles_x, tiles_y);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_fuse_combine(const half_t* a, const half_t* b, const half_t* c, float ca, float cb, float cc, int mode, int N,
                        int H, int W, int C, half_t* out, hipStream_t s, half_t* out_lo, int out_ld) {
  EMP_REQUIRE(C % 8 == 0 && (mode == 0 || mode == 1), "fuse_combine: bad arguments");
  if (out_ld == 0) out_ld = C;
  EMP_REQUIRE(out_ld % 8 == 0 && out_ld >= C, "fuse_combine: bad output stride %d", out_ld);
  EMP_REQUIRE(mode == 1 || (H % 2 == 0 && W % 2 == 0), "fuse_combine: up-sampled operand needs even H, W");
  const int64_t total = (int64_t)N * H * W * (C / 8);
  hipLaunchKernelGGL(fuse_combine_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, a, b, c, ca, cb, cc,
                     mode, N, H, W, C, out, out_lo, out_ld, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_bilinear_ac(const half_t* in, int N, int h, int w, int C, int in_ld, half_t* out, int H, int W, int out_ld,
                       hipStream_t s) {
  EMP_REQUIRE(C % 8 == 0 && in_ld % 8 == 0 && out_ld % 8 == 0, "bilinear: channels must be multiples of 8");
  // area_pixel_compute_scale(align_corners=True): (in-1)/(out-1), 0 when out == 1
  float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int CG = C / 8;
  EMP_REQUIRE(CG >= 1 && CG <= 256, "bilinear: at most 2048 channels");
  const int ppb = 256 / CG;                         // pixels (up4: 4-pixel groups) per block iteration
  static const bool no_up4 = [] { const char* e = getenv("EMP_BILINEAR_NO_UP4"); return e && e[0] == '1'; }();   // A/B runs
  if (!no_up4 && W % 4 == 0 && 3.f * sx < 1.f && W >= 4 * ppb) {
    const int segs = cdiv(W / 4, ppb);
    const int64_t total = (int64_t)N * H * segs;
    EMP_REQUIRE(total < (1ll << 31), "bilinear: too many row segments");
    const int grid = (int)(total < 256 * 16 ? total : 256 * 16);
    hipLaunchKernelGGL(bilinear_ac_up4_kernel, dim3(grid), dim3(256), 0, s, in, N, h, w, C, in_ld, out, H, W, out_ld, sy, sx,
                       segs, ppb);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  const int segs = cdiv(W, ppb);
  const int64_t total = (int64_t)N * H * segs;
  EMP_REQUIRE(total < (1ll << 31), "bilinear: too many row segments");
  const int grid = (int)(total < 256 * 16 ? total : 256 * 16);
  hipLaunchKernelGGL(bilinear_ac_kernel, dim3(grid), dim3(256), 0, s, in, N, h, w, C, in_ld, out, H, W, out_ld, sy, sx, segs,
                     ppb);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_bilinear_ac_f32_nchw(const float* in, int NC, int h, int w, float* out, int scale, hipStream_t s) {
  const int H = h * scale, W = w * scale;
  float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  int64_t total = (int64_t)NC * H * W;
  hipLaunchKernelGGL(bilinear_ac_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, NC, h, w, out, H, W, sy, sx,
                     total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// part: scratch of N*AVG_SEG*C floats
int launch_avgpool(const half_t* in, int N, int HW, int C, int in_ld, float* out, float* part, hipStream_t s) {
  EMP_REQUIRE(C % 8 == 0, "avgpool: C must be a multiple of 8");
  hipLaunchKernelGGL(avgpool_partial_kernel, dim3(N, cdiv(C, 256), AVG_SEG), dim3(256), 0, s, in, HW, C, in_ld, part);
  EMP_LAUNCH_CHECK();
  const int total = N * C;
  hipLaunchKernelGGL(avgpool_final_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, part, C, 1.0f / (float)HW, out,
                     total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}
int avgpool_scratch_floats(int N, int C) { return N * AVG_SEG * C; }

int launch_gemv(const float* in, int N, int K, const float* w, const float* b, int Cout, int relu, float* out,
                hipStream_t s) {
  const int waves = N * Cout;
  hipLaunchKernelGGL(gemv_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, s, in, N, K, w, b, Cout, relu, out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_head1x1(const half_t* in, int N, int P, int K, int in_ld, const float* w, const float* b, int C, float* out,
                   int64_t plane_size, const int32_t* scatter_idx, hipStream_t s) {
  EMP_REQUIRE(K % 8 == 0 && K <= 512, "head1x1: K=%d must be a multiple of 8 and <= 512", K);
  int64_t rows = (int64_t)N * P;
  hipLaunchKernelGGL(head1x1_kernel, dim3(grid_for(rows, 8, 256 * 8)), dim3(256), 0, s, in, N, P, K, in_ld, w, b, C,
                     out, plane_size, scatter_idx);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_zero(void* p, size_t bytes, hipStream_t s) {
  EMP_REQUIRE(bytes % 16 == 0 && ((uintptr_t)p % 16) == 0, "zero: 16-byte granularity");
  if (bytes == 0) return EMP_OK;
  size_t n16 = bytes / 16;
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for((int64_t)n16)), dim3(256), 0, s, (uint4*)p, n16);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
