// NHWC fp16 implicit-GEMM convolution for gfx950 (MI355X): MFMA 16x16x32 f16,
// fp32 accumulate, fused bias / per-image bias / residual / ReLU epilogue.
//
// GEMM view:  out[m][co] = sum_{tap,ci} in[pix(m)+tap][ci] * w[co][tap][ci]
//   m  = (n, oy, ox) flattened output pixel, M = N*Ho*Wo
//   K  = KH*KW*Cin in slabs of BK = 64 input channels; walked group-major: for each group of `kgroup`
//        channel slabs all KH*KW taps, so that the rows a tap re-reads are still in the XCD's L2
//        [measured on ASPP 3x3 d4, 2048 ch: tap-major fetched 13.5x the input (7.4 GB, 6 TB/s = HBM-bound)]
// One workgroup = 256 threads = 4 waves owns a BM x BN (pixels x couts) tile.
// Both operands are K-contiguous in memory (NHWC pixels, [Cout][tap][Cin]
// weights), so every LDS row is one 128-byte K-slab and a fragment is one
// ds_read_b128.  The weights are the MFMA "A" operand (rows = couts) and the
// pixels the "B" operand (cols = pixels): the accumulator then holds 4
// consecutive couts of one pixel per lane, and two cout tiles are interleaved
// at staging time (perm32) so that a lane owns 8 consecutive couts.
//
// LDS image: rows of 128 B, 16-byte chunk c of row r stored at chunk
// c ^ (r & 7)  (conflict-free ds_read_b128 for 16 consecutive rows).  The
// image is written either through registers (ds_write_b128) or by LDS-DMA
// (global_load_lds_dwordx4, lane-linear destination, swizzle applied to the
// per-lane SOURCE address).  Out-of-image taps and out-of-range rows read a
// zero page, so the staging loop is branch-free: per K-step every staged row
// costs one 64-bit pointer increment; the (iy,ix) bounds test runs once per
// tap (every Cin/64 steps).  [measured: the first version recomputed the
// address per step and spent 124 VALU instructions per K-step per wave next to
// 32 MFMAs -- VALU issue, not the matrix pipe, bounded the loop.]
//
// Epilogue: through LDS (each wave transposes its fp32 tile so that one store
// instruction writes whole 128-byte pixel rows; residual read the same way,
// prefetched before the last K-step) or directly from the accumulators.
//
// Work-group -> tile map is XCD-aware: the 8 XCDs get contiguous ranges of
// pixel tiles (neighbouring tiles share input rows through the XCD's L2) and
// the cout tiles of one pixel tile run back to back on the same XCD.
#include "common.h"

// The same kernel as a GROUPED convolution (nn.Conv2d(groups = g): the 3x3 of a RegNet bottleneck) is compiled from this
// source by conv_igemm_grouped.hip, which defines EMP_IGEMM_GROUPED before including it: blockIdx.y = group, the group's
// input-channel / weight / output-channel strides arrive as extra KERNEL ARGUMENTS and ConvParams is untouched -- the struct
// is part of the tuned kernels' code generation (FINDINGS 44).  In this translation unit both macros expand to the tokens
// that were here before.
#ifdef EMP_IGEMM_GROUPED
#define EMP_IGEMM_KARGS const ConvParams p_, const int gs_in, const long long gs_w, const int gs_out
#define EMP_IGEMM_KPROLOGUE                                                \
  ConvParams p = p_;                                                       \
  {                                                                        \
    const int g_ = blockIdx.y;                                             \
    p.in += (size_t)g_ * gs_in;                                            \
    p.wgt += (size_t)g_ * (size_t)gs_w;                                    \
    p.out += (size_t)g_ * gs_out;                                          \
    if (p.bias) p.bias += (size_t)g_ * gs_out;                             \
  }
#else
#define EMP_IGEMM_KARGS const ConvParams p
#define EMP_IGEMM_KPROLOGUE
#endif

#include <type_traits>

namespace emp {

namespace {

constexpr int BK = 64;          // input channels per K-step
constexpr int ROWB = BK * 2;    // bytes per LDS row

__device__ __forceinline__ int perm32(int x) {
  // LDS row (x within a 32-cout group) -> cout within the group.
  // x = t*16 + i (t = cout-tile parity, i = MFMA row) -> (i>>2)*8 + t*4 + (i&3)
  int t = x >> 4, i = x & 15;
  return ((i >> 2) << 3) + (t << 2) + (i & 3);
}

// epilogue activation: 0 none | 1 ReLU | 2 SiLU (x * sigmoid(x), bifpn.py:35,91)
// compile-time ACT: a run-time switch inside the unrolled epilogue costs ~6 scalar branches PER ELEMENT
// [measured: 470-850 s_cbranch per kernel, the epilogue of a 256 x 256 tile took ~10 us]
template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}
// calls f(integral_constant<int, act>) with act in {0, 1, 2}
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
  if (act == 1) f(std::integral_constant<int, 1>{});
  else if (act == 2) f(std::integral_constant<int, 2>{});
  else f(std::integral_constant<int, 0>{});
}
// output element offset of pixel m / cout co.  ps_cout > 0: the GEMM computes a k=2,s=2 transposed
// convolution as 4 sub-pixel 1x1 convs (cout blocks q = dy*2+dx of ps_cout channels each) and the
// store does the pixel shuffle: (n,y,x,q*C+c) -> (n, 2y+dy, 2x+dx, c)   (blocks.py:154-171)
__device__ __forceinline__ size_t out_offset(const ConvParams& p, int m, int co, int HoWo);
// destination of the 8 couts starting at `co` of output pixel m (second destination: ConvParams::out2)
__device__ __forceinline__ half_t* out_ptr(const ConvParams& p, int m, int co, int HoWo) {
  if (p.out2 && co >= p.split) {
    if (p.out3 && co >= p.split3) return p.out3 + (size_t)m * p.out3_ld + (co - p.split3);
    return p.out2 + (size_t)m * p.out2_ld + (co - p.split);
  }
  return p.out + out_offset(p, m, co, HoWo);
}
__device__ __forceinline__ size_t out_offset(const ConvParams& p, int m, int co, int HoWo) {
  if (p.ps_cout == 0) return (size_t)m * p.out_ld + co;
  const int q = co / p.ps_cout, c = co - q * p.ps_cout;
  const int n = m / HoWo, r = m - n * HoWo;
  const int y = r / p.Wo, x = r - y * p.Wo;
  const size_t opix = ((size_t)n * (2 * p.Ho) + 2 * y + (q >> 1)) * (2 * p.Wo) + 2 * x + (q & 1);
  return opix * p.out_ld + c;
}

template <int BM, int BN, int WPM, int WPN, bool GLDS, bool EPI_LDS, int OCC>
__global__ void __launch_bounds__(256, OCC) conv_igemm_kernel(EMP_IGEMM_KARGS) {
  EMP_IGEMM_KPROLOGUE
  constexpr int TM = BM / WPM, TN = BN / WPN;
  constexpr int PT = TM / 16, CT = TN / 16;
  constexpr int A_ITERS = BM / 32, B_ITERS = BN / 32;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static_assert(WPM * WPN == 4, "4 waves per workgroup");
  static_assert(CT % 2 == 0, "cout tiles come in pairs");

  __shared__ __attribute__((aligned(1024))) char lds[2 * (A_BYTES + B_BYTES)];

  // ---- tile assignment (XCD-aware) ----
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int j = bid >> 3;
  const int mtile = xcd * p.mt_per_xcd + j / p.nt;
  const int ntile = j % p.nt;
  if (mtile >= p.mt) return;
  const int m0 = mtile * BM;
  const int n0 = ntile * BN;

  const int tid = threadIdx.x;
  const int w = tid >> 6;
  const int l = tid & 63;
  const int wm = w / WPN, wn = w % WPN;

  // ---- staging bookkeeping: this thread moves chunk (l&7)^(l>>3) of row 8*(w+4i)+(l>>3) ----
  const int srow = l >> 3;
  const int schunk = (l & 7) ^ srow;
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / BK;
  const int S1 = KT * CB;                    // K-steps of the main source
  const int S = S1 + (p.in2 ? p.Cin2 / BK : 0);
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  const half_t* a_img[A_ITERS];   // image base (+ chunk) of the row's pixel, general path
  int a_iy0[A_ITERS], a_ix0[A_ITERS];
  const half_t* a_cur[A_ITERS];   // source of the next K-slab (or the zero page)
  int a_inc[A_ITERS];             // BK, or 0 while parked on the zero page
  const half_t* a_two[A_ITERS];   // second source: the row's pixel in `in2` (or the zero page)
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int m = m0 + 8 * (w + 4 * i) + srow;
    a_img[i] = p.in;
    a_iy0[i] = a_ix0[i] = -(1 << 28);
    a_cur[i] = p.zero;
    a_inc[i] = 0;
    a_two[i] = p.zero;
    if (m < p.M) {
      if (p.in2) {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_two[i] = p.in2 + (((size_t)n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_ld + schunk * 8;
      }
      if (pointwise) {
        a_cur[i] = p.in + (size_t)m * p.in_ld + schunk * 8;
        a_inc[i] = BK;
      } else {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = p.in + (size_t)n * p.H * p.W * p.in_ld + schunk * 8;
      }
    }
  }
  const half_t* b_cur[B_ITERS];
  const half_t* b_base[B_ITERS];
  int b_inc[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int row = 8 * (w + 4 * i) + srow;                  // LDS row in the B tile
    const int co = n0 + (row & ~31) + perm32(row & 31);      // cout staged into that row
    const bool ok = co < p.Cout;
    b_base[i] = b_cur[i] = ok ? p.wgt + (size_t)co * (KT * p.Cin + (p.in2 ? p.Cin2 : 0)) + schunk * 8 : p.zero;
    b_inc[i] = ok ? BK : 0;
  }
  const int KG = p.kgroup;          // channel slabs per group (divides CB); KG == CB: plain tap-major walk

  // ---- epilogue operands that do not depend on the accumulators: fetch them now ----
  constexpr int CPR = TN / 8;       // 8-cout output chunks per staged row
  constexpr int RPI = 64 / CPR;     // pixel rows per wave store instruction
  const int c8 = l % CPR;
  const int co_l = n0 + wn * TN + c8 * 8;   // LDS-epilogue: this lane's 8 couts
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI_LDS) {
    if (p.bias && co_l < p.Cout) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co_l);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co_l + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
      bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
  }

  f16x8 ra[A_ITERS], rb[B_ITERS];
  int st_ky = 0, st_kx = 0, st_cb = 0, st_grp = 0;  // tap / slab within the group / group of the step being staged
  int st_n = 0;                                     // index of the step being staged

  auto stage_issue = [&](int buf) {
    char* a_s = lds + buf * (A_BYTES + B_BYTES);
    char* b_s = a_s + A_BYTES;
    if (st_n == S1 && p.in2) {       // uniform: the main source is exhausted, continue along K in the second one
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        a_cur[i] = a_two[i];
        a_inc[i] = (a_two[i] != p.zero) ? BK : 0;
      }
    }
    ++st_n;
    if (!pointwise && st_cb == 0 && st_n <= S1) {  // uniform: new tap (or group) -> re-derive the row sources once
      const int dy = st_ky * p.dil, dx = st_kx * p.dil;
      const int c0 = st_grp * KG * BK;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        a_cur[i] = ok ? a_img[i] + ((size_t)iy * p.W + ix) * p.in_ld + c0 : p.zero;
        a_inc[i] = ok ? BK : 0;
      }
      if (KG != CB) {
        const int koff = (st_ky * p.KW + st_kx) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) b_cur[i] = b_base[i] + (b_inc[i] ? koff : 0);
      }
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[i],
                                         (__attribute__((address_space(3))) void*)(a_s + (w + 4 * i) * 1024),
                                         16, 0, 0);
      } else {
        ra[i] = *reinterpret_cast<const f16x8*>(a_cur[i]);
      }
      a_cur[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      if constexpr (GLDS) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[i],
                                         (__attribute__((address_space(3))) void*)(b_s + (w + 4 * i) * 1024),
                                         16, 0, 0);
      } else {
        rb[i] = *reinterpret_cast<const f16x8*>(b_cur[i]);
      }
      b_cur[i] += b_inc[i];
    }
    if (++st_cb == KG) {
      st_cb = 0;
      if (++st_kx == p.KW) {
        st_kx = 0;
        if (++st_ky == p.KH) { st_ky = 0; ++st_grp; }
      }
    }
  };
  auto stage_commit = [&](int buf) {
    if constexpr (!GLDS) {
      char* a_s = lds + buf * (A_BYTES + B_BYTES);
      char* b_s = a_s + A_BYTES;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i)
        *reinterpret_cast<f16x8*>(a_s + (w + 4 * i) * 1024 + l * 16) = ra[i];
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i)
        *reinterpret_cast<f16x8*>(b_s + (w + 4 * i) * 1024 + l * 16) = rb[i];
    }
  };

  f32x4 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = l & 15;   // fragment row (cout row / pixel col)
  const int fq = l >> 4;   // k-quarter
  // fragment read offsets are loop-invariant: row*128 + ((kc ^ (row&7))<<4), kc = fq (+4 for the 2nd half)
  int wf_off[CT], pf_off[PT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int row = wn * TN + c * 16 + fr;
    wf_off[c] = A_BYTES + row * ROWB + ((fq ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int row = wm * TM + t * 16 + fr;
    pf_off[t] = row * ROWB + ((fq ^ (row & 7)) << 4);
  }

  auto compute = [&](int buf) {
    const char* base = lds + buf * (A_BYTES + B_BYTES);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 wf[CT], pf[PT];
      // chunk kc+4 = kc ^ 4: second half-slab is the same address with bit 6 flipped
#pragma unroll
      for (int c = 0; c < CT; ++c) wf[c] = *reinterpret_cast<const f16x8*>(base + (wf_off[c] ^ (kk << 6)));
#pragma unroll
      for (int t = 0; t < PT; ++t) pf[t] = *reinterpret_cast<const f16x8*>(base + (pf_off[t] ^ (kk << 6)));
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[t], acc[c][t], 0, 0, 0);
    }
  };

  // ---- main loop: double-buffered LDS, one barrier per K-step; last step peeled ----
  stage_issue(0);
  stage_commit(0);
  __syncthreads();
  for (int s = 0; s + 1 < S; ++s) {
    const int cur = s & 1;
    stage_issue(cur ^ 1);
    compute(cur);
    stage_commit(cur ^ 1);
    __syncthreads();
  }

  if constexpr (EPI_LDS) {
    // residual rows in the layout of the transposed store, requested before the last MFMAs
    f16x8 rres[TM / RPI];
    const bool co_ok = co_l < p.Cout;
    if (p.res) {
#pragma unroll
      for (int i = 0; i < TM / RPI; ++i) {
        const int m = m0 + wm * TM + i * RPI + l / CPR;
        const half_t* src = (m < p.M && co_ok) ? p.res + (size_t)m * p.res_ld + co_l : p.zero;
        rres[i] = *reinterpret_cast<const f16x8*>(src);
      }
    }
    compute((S - 1) & 1);
    __syncthreads();  // every wave is done reading the staging buffers

    // ---- epilogue through LDS (fp32 staging keeps the single fp16 rounding) ----
    constexpr int RB = TN * 4;        // bytes per staged pixel row (fp32)
    constexpr int C4 = TN / 4;        // 16-byte (4 x fp32) chunks per row
    static_assert(4 * TM * RB <= 2 * (A_BYTES + B_BYTES), "staging tile must fit the main-loop LDS");
    char* stg = lds + w * (TM * RB);  // wave-private
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int row = t * 16 + fr;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int ch = (c >> 1) * 8 + fq * 2 + (c & 1);  // fp32 chunk of couts (c>>1)*32 + fq*8 + (c&1)*4 ..+3
        *reinterpret_cast<f32x4*>(stg + row * RB + ((ch ^ (row & (C4 - 1))) << 4)) = acc[c][t];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave reads back only its own writes
    dispatch_act(p.act, [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int i = 0; i < TM / RPI; ++i) {
      const int row = i * RPI + l / CPR;
      const int m = m0 + wm * TM + row;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + row * RB + (((2 * c8) ^ (row & (C4 - 1))) << 4));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + row * RB + (((2 * c8 + 1) ^ (row & (C4 - 1))) << 4));
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = a0[r] + bv[r]; v[4 + r] = a1[r] + bv[4 + r]; }
      if (m >= p.M || !co_ok) continue;
      if (p.bias_n) {
        const float* bn = p.bias_n + (size_t)(m / HoWo) * p.Cout + co_l;
        const float4 b0 = *reinterpret_cast<const float4*>(bn);
        const float4 b1 = *reinterpret_cast<const float4*>(bn + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rres[i][r];
      }
      f16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (half_t)apply_act<ACT>(v[r]);
      *reinterpret_cast<f16x8*>(out_ptr(p, m, co_l, HoWo)) = o;
    }
    });
    return;
  } else {
    compute((S - 1) & 1);
    // ---- direct epilogue: lane (fq, fr) owns couts P*32 + fq*8 + [0,8) of pixel fr of each pixel tile ----
    dispatch_act(p.act, [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int m = m0 + wm * TM + t * 16 + fr;
      if (m >= p.M) continue;
      const int n_img = (p.bias_n != nullptr) ? m / HoWo : 0;
#pragma unroll
      for (int P = 0; P < CT / 2; ++P) {
        const int co = n0 + wn * TN + P * 32 + fq * 8;
        if (co >= p.Cout) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[2 * P][t][r];
          v[4 + r] = acc[2 * P + 1][t][r];
        }
        if (p.bias) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
          const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.bias_n) {
          const float* bn = p.bias_n + (size_t)n_img * p.Cout + co;
          const float4 b0 = *reinterpret_cast<const float4*>(bn);
          const float4 b1 = *reinterpret_cast<const float4*>(bn + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.res) {
          const f16x8 rv = *reinterpret_cast<const f16x8*>(p.res + (size_t)m * p.res_ld + co);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
        }
        f16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = (half_t)apply_act<ACT>(v[r]);
        *reinterpret_cast<f16x8*>(out_ptr(p, m, co, HoWo)) = o;
      }
    }
    });
  }
}

#ifdef EMP_IGEMM_GROUPED
template <int BM, int BN, int WPM, int WPN, bool GLDS, bool EPI_LDS, int OCC = 2>
int launch_grouped_tpl(ConvParams p, int groups, int gs_in, long long gs_w, int gs_out, hipStream_t stream) {
  p.mt = cdiv(p.M, BM);
  p.nt = cdiv(p.Cout, BN);
  p.mt_per_xcd = cdiv(p.mt, 8);
  const int grid = 8 * p.mt_per_xcd * p.nt;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WPM, WPN, GLDS, EPI_LDS, OCC>), dim3(grid, groups), dim3(256), 0, stream, p, gs_in,
                     gs_w, gs_out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace

// p describes ONE group (Cin = the group's input channels padded to 64 with zero weights, Cout = its output channels);
// group g reads p.in + g * gs_in, weights p.wgt + g * gs_w, writes p.out + g * gs_out (bias likewise).  Plain conv only.
int launch_conv_igemm_grouped(const ConvParams& p, int groups, int gs_in, long long gs_w, int gs_out, hipStream_t stream) {
  EMP_REQUIRE(groups >= 1 && groups < 65536, "grouped conv: bad group count %d", groups);
  EMP_REQUIRE(p.Cin % BK == 0 && p.Cin > 0 && p.in_ld % 8 == 0 && (groups - 1) * gs_in + p.Cin <= p.in_ld && gs_in % 8 == 0,
              "grouped conv: Cin=%d (multiple of 64), channel step %d (multiple of 8), row %d", p.Cin, gs_in, p.in_ld);
  EMP_REQUIRE(p.Cout % 8 == 0 && p.Cout > 0 && gs_out % 8 == 0 && gs_out >= p.Cout && p.out_ld % 8 == 0 &&
                  (groups - 1) * gs_out + p.Cout <= p.out_ld, "grouped conv: Cout=%d, cout step %d, row %d", p.Cout, gs_out, p.out_ld);
  EMP_REQUIRE(gs_w % 8 == 0 && gs_w >= (long long)p.Cout * p.KH * p.KW * p.Cin, "grouped conv: weight stride %lld", gs_w);
  EMP_REQUIRE(!p.in2 && !p.out2 && !p.out3 && !p.next_w && !p.res && !p.bias_n && p.ps_cout == 0, "grouped conv: plain convolutions only");
  EMP_REQUIRE(p.act >= 0 && p.act <= 2, "grouped conv: act must be 0, 1 or 2");
  EMP_REQUIRE(((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.out % 16) == 0 && ((uintptr_t)p.wgt % 16) == 0 &&
                  ((uintptr_t)p.zero % 256) == 0 && p.zero != nullptr, "grouped conv: pointers must be 16-byte aligned (zero page 256)");
  EMP_REQUIRE((int64_t)p.N * p.Ho * p.Wo < (1ll << 31), "grouped conv: too many output pixels");
  ConvParams q = p;
  {
    const int CB = p.Cin / 64, KT = p.KH * p.KW;
    int kg = (KT > 1 && CB > 4 && CB % 4 == 0) ? 4 : CB;
    q.kgroup = kg;
  }
  // 64-wide cout tiles when they pad less (56 -> 64, 72 -> 128 either way: the 128 x 64 tile then has twice the workgroups)
  if (p.Cout <= 64 || cdiv(p.Cout, 64) * 64 <= cdiv(p.Cout, 128) * 128)
    return launch_grouped_tpl<128, 64, 2, 2, true, true, 3>(q, groups, gs_in, gs_w, gs_out, stream);
  return launch_grouped_tpl<128, 128, 2, 2, true, true>(q, groups, gs_in, gs_w, gs_out, stream);
}

#else   // !EMP_IGEMM_GROUPED
template <int BM, int BN, int WPM, int WPN, bool GLDS, bool EPI_LDS, int OCC = 2>
int launch_tpl(ConvParams p, hipStream_t stream) {
  p.mt = cdiv(p.M, BM);
  p.nt = cdiv(p.Cout, BN);
  p.mt_per_xcd = cdiv(p.mt, 8);
  const int grid = 8 * p.mt_per_xcd * p.nt;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WPM, WPN, GLDS, EPI_LDS, OCC>), dim3(grid), dim3(256), 0, stream, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace

// auto dispatch rule: 256x256 tile (conv_igemm256.hip) when it fills the chip -- one workgroup per CU, so it needs
// >= ~1 tile per CU -- and K is long enough to amortise its prologue / epilogue: K >= 512, or K >= 128 when the
// reduction continues over a second source (projection shortcuts: 128 x 128 tiles would re-read both sources per cout
// tile).  [profiles/r02_rule_sweep.txt: layer1.0 conv3+ds 508 -> 402 us, layer2.0 conv3+ds 419 -> 313, layer4 conv3
// 490 -> 450; K = 128 / 256 with a residual (layer2 / layer3 conv3) stay on 128 x 128: 263 vs 294, 165 vs 180.]
bool conv_uses_256(const ConvParams& p) {
  static const bool no256 = [] { const char* e = getenv("EMP_CONV_NO256"); return e && e[0] == '1'; }();   // A/B runs
  static const int min_k = [] { const char* e = getenv("EMP_CONV_256_MINK"); return e ? atoi(e) : 512; }();
  const int64_t tiles256 = (int64_t)cdiv(p.M, 256) * (p.Cout / 256);
  const int k256 = p.KH * p.KW * p.Cin + (p.in2 ? p.Cin2 : 0);
  const bool out2_ok = !p.out2 || (!p.out3 && p.split % 256 == 0);      // whole cout tiles to the second tensor
  return !no256 && out2_ok && conv_igemm256_supported(p) && tiles256 >= 192 && k256 >= (p.in2 ? min_k / 4 : min_k);
}
// half tile 256 pixels x 128 couts, two workgroups per CU (conv_igemm256.hip): the 128-cout layers with K >= 256
// (layer2 conv1 / stride-2 conv2: 350 -> 325 us, 244 -> 230) -- the 128 x 128 tile is LDS-read bound there
bool conv_uses_h256(const ConvParams& p) {
  static const int h256 = [] { const char* e = getenv("EMP_CONV_H256"); return e ? atoi(e) : 1; }();   // A/B runs: 0 off, 2 all
  if (!h256 || !conv_igemm_h256_supported(p)) return false;
  const int64_t tiles = (int64_t)cdiv(p.M, 128) * (p.Cout / 128) / 2;
  if (tiles < 1024) return false;
  return h256 == 2 || (p.Cout == 128 && p.KH * p.KW * p.Cin >= 256);
}

int launch_conv_igemm(const ConvParams& p, int variant, hipStream_t stream) {
  EMP_REQUIRE(p.Cin % BK == 0 && p.Cin > 0, "conv: Cin=%d must be a positive multiple of 64", p.Cin);
  EMP_REQUIRE(p.in_ld % 8 == 0 && p.in_ld >= p.Cin, "conv: in_ld=%d must be >= Cin and a multiple of 8", p.in_ld);
  EMP_REQUIRE(p.Cout % 8 == 0 && p.Cout > 0, "conv: Cout=%d must be a positive multiple of 8", p.Cout);
  EMP_REQUIRE(p.out_ld % 8 == 0 && p.out_ld >= (p.ps_cout ? p.ps_cout : (p.out2 ? p.split : p.Cout)), "conv: out_ld=%d invalid", p.out_ld);
  EMP_REQUIRE(p.ps_cout == 0 || (p.ps_cout % 8 == 0 && p.Cout == 4 * p.ps_cout && p.res == nullptr),
              "conv: pixel-shuffle store needs Cout == 4*ps_cout, ps_cout %% 8 == 0, no residual");
  EMP_REQUIRE(p.act >= 0 && p.act <= 2, "conv: act must be 0 (none), 1 (ReLU) or 2 (SiLU)");
  EMP_REQUIRE(p.res == nullptr || p.res_ld % 8 == 0, "conv: res_ld=%d must be a multiple of 8", p.res_ld);
  EMP_REQUIRE(((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.out % 16) == 0 && ((uintptr_t)p.wgt % 16) == 0 &&
                  ((uintptr_t)p.res % 16) == 0 && ((uintptr_t)p.zero % 256) == 0 && p.zero != nullptr,
              "conv: pointers must be 16-byte aligned (zero page 256)");
  EMP_REQUIRE((int64_t)p.N * p.Ho * p.Wo < (1ll << 31), "conv: too many output pixels");
  if (p.out2) {
    EMP_REQUIRE(p.split > 0 && p.split < p.Cout && p.split % 8 == 0 && p.out2_ld % 8 == 0 &&
                    p.out2_ld >= (p.out3 ? p.split3 : p.Cout) - p.split &&
                    p.out_ld >= p.split && ((uintptr_t)p.out2 % 16) == 0 && p.ps_cout == 0 && p.res == nullptr,
                "conv: second destination: split=%d must be a multiple of 8 inside (0, Cout=%d)", p.split, p.Cout);
  }
  if (p.out3) {
    EMP_REQUIRE(p.out2 != nullptr && p.split3 > p.split && p.split3 < p.Cout && p.split3 % 8 == 0 && p.out3_ld % 8 == 0 &&
                    p.out3_ld >= p.Cout - p.split3 && ((uintptr_t)p.out3 % 16) == 0,
                "conv: third destination: split3=%d must be a multiple of 8 inside (split=%d, Cout=%d)", p.split3, p.split, p.Cout);
  }
  if (p.in2) {
    // (a pixel-shuffle store is fine: the second source only extends the K walk, the store is the epilogue's business --
    // round 4: the transposed convs of the BiFPN decoder carry their weights as [hi | lo] pairs against the same input)
    EMP_REQUIRE(p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0,
                "conv: a second source needs a 1x1 / stride 1 main convolution");
    EMP_REQUIRE(p.Cin2 > 0 && p.Cin2 % BK == 0 && p.in2_ld % 8 == 0 && p.in2_ld >= p.Cin2 && ((uintptr_t)p.in2 % 16) == 0,
                "conv: second source: Cin2=%d must be a positive multiple of 64, in2_ld=%d >= Cin2", p.Cin2, p.in2_ld);
    EMP_REQUIRE(p.stride2 >= 1 && (p.Ho - 1) * p.stride2 < p.H2 && (p.Wo - 1) * p.stride2 < p.W2,
                "conv: second source %dx%d / stride %d does not cover the %dx%d output", p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  }
  // variant = staging + 16 * tile
  //   staging: 0 auto | 1 register staging | 2 LDS-DMA | 3 LDS-DMA + LDS-transposed epilogue
  //   tile   : 0 auto | 1 128x128 | 2 128x64 | 3 64x64 | 4 256x256 (conv_igemm256.hip; Cout % 256 == 0) |
  //            5 half tile 128x256 / 256x128, two workgroups per CU (conv_igemm256.hip; Cout % 128 == 0) | 6 conv3x3_c64 |
  //            7 64x64 with a deep LDS-DMA ring (conv_igemm_s64.hip): what 'auto' takes instead of 3 for few-tile launches
  // (tried and removed, slower on MI355X: a persistent cross-tile pipeline, and a 256x128 8-wave tile with
  //  three LDS stages + counted vmcnt / raw barriers: 865 vs 990 TF/s on the ASPP shape -- DESIGN.md section 4;
  //  a 128x128 4-wave tile with the 256x256 kernel's four-slot ring of 32-channel K-tiles, LDS-DMA three tiles ahead,
  //  register-double-buffered fragments and one LDS-only barrier per K-tile, two workgroups per CU: bit-identical
  //  results but 5-15 % SLOWER than the two-stage kernel on every layer shape (e.g. 512->2048 1x1: 0.547 vs 0.502 ms,
  //  128->512: 0.316 vs 0.278) -- with two workgroups per CU the global-load latency is already covered and the
  //  second barrier per 64 channels costs more than the deeper prefetch saves)
  int v = variant & 15, tile = (variant >> 4) & 15, kg = (variant >> 8) & 255, mode256 = (variant >> 16) & 15;
  if ((variant >> 20) & 1) {      // ConvParams::wgt is the 256 x 256 tile's packed image (conv256_pack_weights): that tile, default K walk
    EMP_REQUIRE(!p.next_w && conv_igemm256_supported(p), "conv: packed weights belong to the 256x256 tile");
    return launch_conv_igemm256(p, stream, 0, mode256, true);
  }
  if (p.next_w) {
    EMP_REQUIRE(conv_b2b_supported(p), "conv: a fused next convolution needs the 256x256 tile (Cout == 256, >= 192 tiles)");
    return launch_conv_igemm256(p, stream, kg, mode256);
  }
  EMP_REQUIRE(v <= 3 && tile <= 7, "conv: bad variant %d", variant);
  if (tile == 7) return launch_conv_igemm_s64(p, stream, kg);
  if (tile == 6) return launch_conv3x3_c64(p, stream);
  if (tile == 5) return launch_conv_igemm_h256(p, stream, kg);      // kg counts 32-channel slabs
  {
    // 64 -> 64 channel 3x3 (ResNet layer1 conv2): weights in registers, halo tile in LDS (conv3x3c64.hip), once there
    // are enough 8 x 16 tiles for its 256 persistent workgroups
    static const bool no_c64 = [] { const char* e = getenv("EMP_CONV_NO_C64"); return e && e[0] == '1'; }();   // A/B runs
    if (tile == 0 && !no_c64 && conv3x3_c64_supported(p) && (int64_t)p.N * cdiv(p.H, 8) * cdiv(p.W, 16) >= 1024)
      return launch_conv3x3_c64(p, stream);
  }
  if (tile == 4) {
    return launch_conv_igemm256(p, stream, kg, mode256);      // here kg counts 32-channel slabs
  }
  // K walk (variant bits 8+: 0 auto | g = channel slabs per group): with many input channels and several taps a
  // tap-major walk streams the whole input once per tap through an L2 that holds only ~4 MB per XCD; groups of
  // 4 slabs (256 channels x ~4096 pixels in flight per XCD = 2 MB) keep the 9 taps' re-reads in L2.
  ConvParams q = p;
  {
    const int CB = p.Cin / 64, KT = p.KH * p.KW;
    static const int env_kg = [] { const char* e = getenv("EMP_CONV_KGROUP"); return e ? atoi(e) : 0; }();   // A/B runs
    if (kg == 0) kg = env_kg;
    if (kg == 0) kg = (KT > 1 && CB > 4 && CB % 4 == 0) ? 4 : CB;
    if (kg > CB || CB % kg != 0 || KT == 1) kg = CB;
    q.kgroup = kg;
  }
  if (tile == 0) {
    if (conv_uses_256(p)) return launch_conv_igemm256(p, stream);
    if (conv_uses_h256(p)) return launch_conv_igemm_h256(p, stream);
    tile = (p.Cout <= 64) ? 2 : 1;
    // a cout count that pads less in 64-wide tiles than in 128-wide ones (176 = 128 + 48: layer2.0 conv1 + the decoders'
    // projections in one launch) runs on the 128 x 64 tile: 564 -> 520 us (tools/tile_ab.py), same K order
    if (p.Cout > 128 && cdiv(p.Cout, 64) * 64 < cdiv(p.Cout, 128) * 128) tile = 2;
    // a single image leaves the deep layers with a handful of 128x128 tiles for 256 CUs (batch-1 latency, the
    // reference's own calling convention): 64x64 tiles walk K in the same order (bit-identical results) on 4x the CUs
    static const int small_below = [] { const char* e = getenv("EMP_CONV_SMALL_TILES_BELOW"); return e ? atoi(e) : 256; }();
    if (tile == 1 && (int64_t)cdiv(p.M, 128) * cdiv(p.Cout, 128) < small_below && !p.out2) {
      // one workgroup per CU: only the workgroup's own prefetch hides the memory latency -> deep-ring variant
      static const bool no_s64 = [] { const char* e = getenv("EMP_CONV_NO_S64"); return e && e[0] == '1'; }();   // A/B runs
      if (!no_s64 && conv_igemm_s64_supported(p)) return launch_conv_igemm_s64(p, stream);
      tile = 3;
    }
  }
  if (v == 0) v = 3;
  if (tile == 1) {
    if (v == 1) return launch_tpl<128, 128, 2, 2, false, false>(q, stream);
    if (v == 2) return launch_tpl<128, 128, 2, 2, true, false>(q, stream);
    return launch_tpl<128, 128, 2, 2, true, true>(q, stream);
  }
  if (tile == 2) {
    if (v == 1) return launch_tpl<128, 64, 2, 2, false, false, 3>(q, stream);
    if (v == 2) return launch_tpl<128, 64, 2, 2, true, false, 3>(q, stream);
    return launch_tpl<128, 64, 2, 2, true, true, 3>(q, stream);
  }
  if (v == 1) return launch_tpl<64, 64, 2, 2, false, false, 5>(q, stream);
  if (v == 2) return launch_tpl<64, 64, 2, 2, true, false, 5>(q, stream);
  return launch_tpl<64, 64, 2, 2, true, true, 5>(q, stream);
}

#endif  // EMP_IGEMM_GROUPED

}  // namespace emp
