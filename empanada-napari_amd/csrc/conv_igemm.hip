// NHWC fp16 implicit-GEMM convolution for gfx950 (MI355X): MFMA 16x16x32 f16,
// fp32 accumulate, fused bias / per-image bias / residual / ReLU epilogue.
//
// GEMM view:  out[m][co] = sum_{tap,ci} in[pix(m)+tap][ci] * w[co][tap][ci]
//   m  = (n, oy, ox) flattened output pixel, M = N*Ho*Wo
//   K  = KH*KW*Cin, walked tap-major in slabs of BK = 64 input channels
// One workgroup = 256 threads = 4 waves owns a BM x BN (pixels x couts) tile.
// Both operands are K-contiguous in memory (NHWC pixels, [Cout][tap][Cin]
// weights), so every LDS row is one 128-byte K-slab and a fragment is one
// ds_read_b128.  The weights are the MFMA "A" operand (rows = couts) and the
// pixels the "B" operand (cols = pixels): the accumulator then holds 4
// consecutive couts of one pixel per lane, and two cout tiles are interleaved
// at staging time (perm32) so that a lane owns 8 consecutive couts = one
// 16-byte NHWC store.
//
// LDS image: rows of 128 B, 16-byte chunk c of row r stored at chunk
// c ^ (r & 7)  (conflict-free ds_read_b128 for 16 consecutive rows).  The
// image is written either through registers (ds_write_b128) or by LDS-DMA
// (global_load_lds_dwordx4, lane-linear destination, swizzle applied to the
// per-lane SOURCE address; out-of-image lanes read a zero page).
//
// Work-group -> tile map is XCD-aware: the 8 XCDs get contiguous ranges of
// pixel tiles (neighbouring tiles share input rows through the XCD's L2) and
// the cout tiles of one pixel tile run back to back on the same XCD.
#include "common.h"

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

template <int BM, int BN, int WPM, int WPN, bool GLDS, bool EPI_LDS>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const ConvParams p) {
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
  if (mtile >= p.mt || j / p.nt >= p.mt_per_xcd) return;
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

  const half_t* a_base[A_ITERS];
  int a_iy0[A_ITERS], a_ix0[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    int m = m0 + 8 * (w + 4 * i) + srow;
    if (m < p.M) {
      int n = m / HoWo;
      int r = m - n * HoWo;
      int oy = r / p.Wo;
      int ox = r - oy * p.Wo;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_base[i] = p.in + (size_t)n * p.H * p.W * p.in_ld + schunk * 8;
    } else {
      a_iy0[i] = -(1 << 28);
      a_ix0[i] = -(1 << 28);
      a_base[i] = p.in;
    }
  }
  const half_t* b_ptr[B_ITERS];
  bool b_ok[B_ITERS];
  const int KT = p.KH * p.KW;
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    int row = 8 * (w + 4 * i) + srow;                       // LDS row in the B tile
    int co = n0 + (row & ~31) + perm32(row & 31);           // cout staged into that row
    b_ok[i] = co < p.Cout;
    b_ptr[i] = p.wgt + (size_t)(b_ok[i] ? co : 0) * KT * p.Cin + schunk * 8;
  }

  const int CB = p.Cin / BK;
  const int S = KT * CB;

  f16x8 ra[A_ITERS], rb[B_ITERS];
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // tap / channel-block counters of the step being staged
  int st_ky = 0, st_kx = 0, st_cb = 0;

  auto stage_issue = [&](int buf) {
    char* a_s = lds + buf * (A_BYTES + B_BYTES);
    char* b_s = a_s + A_BYTES;
    const int dy = st_ky * p.dil, dx = st_kx * p.dil;
    const int coff = st_cb * BK;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const half_t* src = a_base[i] + ((size_t)iy * p.W + ix) * p.in_ld + coff;
      if constexpr (GLDS) {
        const half_t* s2 = ok ? src : p.zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s2,
                                         (__attribute__((address_space(3))) void*)(a_s + (w + 4 * i) * 1024),
                                         16, 0, 0);
      } else {
        ra[i] = ok ? *reinterpret_cast<const f16x8*>(src) : zero8;
      }
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const half_t* src = b_ptr[i];
      if constexpr (GLDS) {
        const half_t* s2 = b_ok[i] ? src : p.zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s2,
                                         (__attribute__((address_space(3))) void*)(b_s + (w + 4 * i) * 1024),
                                         16, 0, 0);
      } else {
        rb[i] = b_ok[i] ? *reinterpret_cast<const f16x8*>(src) : zero8;
      }
      b_ptr[i] += BK;
    }
    // advance to the next step
    if (++st_cb == CB) {
      st_cb = 0;
      if (++st_kx == p.KW) { st_kx = 0; ++st_ky; }
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

  auto compute = [&](int buf) {
    const char* a_s = lds + buf * (A_BYTES + B_BYTES);
    const char* b_s = a_s + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kc = fq + 4 * kk;
      f16x8 wf[CT], pf[PT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        int row = wn * TN + c * 16 + fr;
        wf[c] = *reinterpret_cast<const f16x8*>(b_s + row * ROWB + ((kc ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        int row = wm * TM + t * 16 + fr;
        pf[t] = *reinterpret_cast<const f16x8*>(a_s + row * ROWB + ((kc ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t)
          acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[t], acc[c][t], 0, 0, 0);
    }
  };

  // ---- main loop: double-buffered LDS, one barrier per K-step ----
  stage_issue(0);
  stage_commit(0);
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    const int cur = s & 1;
    if (s + 1 < S) stage_issue(cur ^ 1);
    compute(cur);
    if (s + 1 < S) stage_commit(cur ^ 1);
    __syncthreads();
  }

  if constexpr (EPI_LDS) {
    // ---- epilogue through LDS: each wave transposes its TM x TN fp32 tile so that one wave
    //      store instruction writes whole TN*2-byte pixel rows (full 128-byte lines for TN = 64)
    //      and the residual is read the same way.  fp32 staging keeps the single fp16 rounding. ----
    constexpr int RB = TN * 4;        // bytes per staged pixel row (fp32)
    constexpr int C4 = TN / 4;        // 16-byte (4 x fp32) chunks per row
    constexpr int CPR = TN / 8;       // 8-cout output chunks per row
    constexpr int RPI = 64 / CPR;     // pixel rows per wave store instruction
    static_assert(4 * TM * RB <= 2 * (A_BYTES + B_BYTES), "staging tile must fit the main-loop LDS");
    char* stg = lds + w * (TM * RB);  // wave-private (main-loop buffers are dead: last barrier passed)
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
    const int c8 = l % CPR;
    const int co = n0 + wn * TN + c8 * 8;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool co_ok = co < p.Cout;
    if (p.bias && co_ok) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
      bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < TM / RPI; ++i) {
      const int row = i * RPI + l / CPR;
      const int m = m0 + wm * TM + row;
      if (m >= p.M || !co_ok) continue;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + row * RB + (((2 * c8) ^ (row & (C4 - 1))) << 4));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + row * RB + (((2 * c8 + 1) ^ (row & (C4 - 1))) << 4));
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = a0[r] + bv[r]; v[4 + r] = a1[r] + bv[4 + r]; }
      if (p.bias_n) {
        const float* bn = p.bias_n + (size_t)(m / HoWo) * p.Cout + co;
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
      for (int r = 0; r < 8; ++r) {
        float x = v[r];
        if (p.relu) x = x > 0.f ? x : 0.f;
        o[r] = (half_t)x;
      }
      *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + co) = o;
    }
    return;
  }
  // ---- epilogue: lane (fq, fr) owns couts P*32 + fq*8 + [0,8) of pixel fr of each pixel tile ----
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
      for (int r = 0; r < 8; ++r) {
        float x = v[r];
        if (p.relu) x = x > 0.f ? x : 0.f;
        o[r] = (half_t)x;
      }
      *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + co) = o;
    }
  }
}

template <int BM, int BN, int WPM, int WPN, bool GLDS, bool EPI_LDS>
int launch_tpl(ConvParams p, hipStream_t stream) {
  p.mt = cdiv(p.M, BM);
  p.nt = cdiv(p.Cout, BN);
  p.mt_per_xcd = cdiv(p.mt, 8);
  const int grid = 8 * p.mt_per_xcd * p.nt;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WPM, WPN, GLDS, EPI_LDS>), dim3(grid), dim3(256), 0, stream, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace

int launch_conv_igemm(const ConvParams& p, int variant, hipStream_t stream) {
  EMP_REQUIRE(p.Cin % BK == 0 && p.Cin > 0, "conv: Cin=%d must be a positive multiple of 64", p.Cin);
  EMP_REQUIRE(p.in_ld % 8 == 0 && p.in_ld >= p.Cin, "conv: in_ld=%d must be >= Cin and a multiple of 8", p.in_ld);
  EMP_REQUIRE(p.Cout % 8 == 0 && p.Cout > 0, "conv: Cout=%d must be a positive multiple of 8", p.Cout);
  EMP_REQUIRE(p.out_ld % 8 == 0 && p.out_ld >= p.Cout, "conv: out_ld=%d invalid", p.out_ld);
  EMP_REQUIRE(p.res == nullptr || p.res_ld % 8 == 0, "conv: res_ld=%d must be a multiple of 8", p.res_ld);
  EMP_REQUIRE(((uintptr_t)p.in % 16) == 0 && ((uintptr_t)p.out % 16) == 0 && ((uintptr_t)p.wgt % 16) == 0 &&
                  ((uintptr_t)p.res % 16) == 0,
              "conv: pointers must be 16-byte aligned");
  EMP_REQUIRE((int64_t)p.N * p.Ho * p.Wo < (1ll << 31), "conv: too many output pixels");
  // variant: 0 auto | 1 register staging | 2 LDS-DMA staging | 3 LDS-DMA staging + LDS-transposed epilogue
  const int v = variant == 0 ? 2 : variant;
  if (p.Cout > 64) {
    if (v == 1) return launch_tpl<128, 128, 2, 2, false, false>(p, stream);
    if (v == 2) return launch_tpl<128, 128, 2, 2, true, false>(p, stream);
    return launch_tpl<128, 128, 2, 2, true, true>(p, stream);
  }
  if (v == 1) return launch_tpl<128, 64, 2, 2, false, false>(p, stream);
  if (v == 2) return launch_tpl<128, 64, 2, 2, true, false>(p, stream);
  return launch_tpl<128, 64, 2, 2, true, true>(p, stream);
}

}  // namespace emp
