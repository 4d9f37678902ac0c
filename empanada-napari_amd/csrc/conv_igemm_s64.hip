// 64 x 64 implicit-GEMM convolution tile with a DEEP LDS-DMA ring, for launches with fewer tiles than the chip has
// CUs x 2 -- the batch-1 call of the reference API (`Engine2d.infer`, empanada/inference/engines.py:300-325: one image
// per call): a single 1024^2 tile leaves the 64^2-pixel deep layers with 256 - 2048 tiles of this size, i.e. one
// workgroup per CU, and nothing but the workgroup's own prefetch hides the L2 / HBM latency.  The two-stage 64 x 64
// variant of conv_igemm.hip waits a full memory latency per 64-channel K-step (~1 us: an ASPP 3x3 with K = 18432 took
// 0.3 ms per launch, 73 % of the batch-1 forward).  Here K moves in 32-channel K-tiles (64 pixel rows + 64 cout rows of
// 64 B = 8 KiB, two 1-KiB LDS-DMA pieces per wave), taken in PAIRS through a ring of NS = 5 slots filled NS - 1 pairs
// ahead (64 KiB in flight per workgroup), one LDS-only barrier per pair (64 channels), counted vmcnt.  4 waves, wave tile 32 pixels x 32 couts
// (4 MFMAs and 4 fragment reads per K-tile).  K order, operands and epilogue arithmetic are those of the other conv
// kernels: bit-identical results (tests/test_gpu_conv.py), so a batch of N still equals N batch-1 calls.
#include "common.h"

#include <type_traits>

namespace emp {

namespace {

constexpr int SKS = 32;                   // channels per K-tile
constexpr int S_SLOT = 128 * 64;          // 64 pixel rows + 64 cout rows
constexpr int S_WOFF = 64 * 64;

__device__ __forceinline__ int s_perm32(int x) {
  const int t = x >> 4, i = x & 15;
  return ((i >> 2) << 3) + (t << 2) + (i & 3);
}
template <int ACT>
__device__ __forceinline__ float s_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

template <int NS>
__global__ void __launch_bounds__(256, 2) conv_igemm_s64_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int mtile = xcd * p.mt_per_xcd + j / p.nt;
  const int ntile = j % p.nt;
  if (mtile >= p.mt) return;
  const int m0 = mtile * 64, n0 = ntile * 64;

  const int tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wc = wave & 1;          // pixel half / cout half of the tile
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / SKS;
  const int KG = p.kgroup;
  const int KT1 = KT * CB;
  const int KTOT = KT1 + (p.in2 ? p.Cin2 / SKS : 0);
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  // staging: wave w moves pixel piece w and cout piece w (16 rows x 64 B each)
  const int srow = l >> 2;
  const int schunk = (l & 3) ^ ((-(l >> 4)) & 3);
  const half_t* a_img = p.in;
  int a_iy0 = -(1 << 28), a_ix0 = -(1 << 28);
  const half_t* a_cur = p.zero;
  int a_inc = 0;
  const half_t* a_two = p.zero;
  {
    const int m = m0 + wave * 16 + srow;
    if (m < p.M) {
      const int n = m / HoWo;
      const int r = m - n * HoWo;
      const int oy = r / p.Wo;
      const int ox = r - oy * p.Wo;
      if (p.in2) a_two = p.in2 + (((size_t)n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_ld + schunk * 8;
      if (pointwise) {
        a_cur = p.in + (size_t)m * p.in_ld + schunk * 8;
        a_inc = SKS;
      } else {
        a_iy0 = oy * p.stride - p.pad;
        a_ix0 = ox * p.stride - p.pad;
        a_img = p.in + (size_t)n * p.H * p.W * p.in_ld + schunk * 8;
      }
    }
  }
  const half_t* b_base;
  const half_t* b_cur;
  int b_inc;
  {
    const int row = wave * 16 + srow;
    const int co = n0 + (row & ~31) + s_perm32(row & 31);
    const bool ok = co < p.Cout;
    b_base = b_cur = ok ? p.wgt + (size_t)co * (KT * p.Cin + (p.in2 ? p.Cin2 : 0)) + schunk * 8 : p.zero;
    b_inc = ok ? SKS : 0;
  }
  int st_ky = 0, st_kx = 0, st_cb = 0, st_grp = 0, st_u = 0;
  // address work + the 2 LDS-DMA pieces of K-tile st_u into half `h` of ring slot `slot`; K-tiles past the end of K
  // (the second half of the last pair when the tile count is odd) copy the zero page: 0 x 0 leaves the sums untouched
  auto stage = [&](char* dst) {
    if (st_u >= KTOT) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p.zero,
                                       (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p.zero,
                                       (__attribute__((address_space(3))) void*)(dst + S_WOFF + wave * 1024), 16, 0, 0);
      ++st_u;
      return;
    }
    if (st_u == KT1 && p.in2) {
      a_cur = a_two;
      a_inc = (a_two != p.zero) ? SKS : 0;
    }
    if (st_cb == 0 && st_u < KT1) {
      const int c0 = st_grp * KG * SKS;
      if (!pointwise) {
        const int iy = a_iy0 + st_ky * p.dil, ix = a_ix0 + st_kx * p.dil;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        a_cur = ok ? a_img + ((size_t)iy * p.W + ix) * p.in_ld + c0 : p.zero;
        a_inc = ok ? SKS : 0;
      }
      if (KG != CB) b_cur = b_inc ? b_base + (st_ky * p.KW + st_kx) * p.Cin + c0 : b_base;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur,
                                     (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur,
                                     (__attribute__((address_space(3))) void*)(dst + S_WOFF + wave * 1024), 16, 0, 0);
    a_cur += a_inc;
    b_cur += b_inc;
    if (++st_cb == KG) {
      st_cb = 0;
      if (++st_kx == p.KW) {
        st_kx = 0;
        if (++st_ky == p.KH) { st_ky = 0; ++st_grp; }
      }
    }
    ++st_u;
  };
  int st_pair = 0;
  auto stage_pair = [&]() {      // K-tiles 2*st_pair, 2*st_pair + 1 into ring slot st_pair % NS (two 8 KiB halves)
    char* slot = lds + (st_pair % NS) * (2 * S_SLOT);
    stage(slot);
    stage(slot + S_SLOT);
    ++st_pair;
  };

  const int fr = l & 15, fq = l >> 4;
  const int foff = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  const int p_off = wp * 32 * 64 + foff;                 // + q*1024
  const int w_off = S_WOFF + wc * 32 * 64 + foff;        // + c*1024

  f32x4 acc[2][2];      // [cout tile][pixel tile]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K-tiles move in PAIRS (64 channels per barrier: the loop is bound by the LDS-read latency + barrier per iteration,
  // not by its 4 MFMAs per K-tile); prologue: up to NS - 1 pairs in flight, the first one landed
  const int NPAIR = (KTOT + 1) >> 1;
  const int npro = NPAIR < NS - 1 ? NPAIR : NS - 1;
#pragma unroll 1
  for (int u = 0; u < npro; ++u) stage_pair();
  if (npro == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (NS - 2)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  for (int t = 0; t < NPAIR; ++t) {
    const char* base = lds + (t % NS) * (2 * S_SLOT);
    const f16x8 w0 = *reinterpret_cast<const f16x8*>(base + w_off);
    const f16x8 w1 = *reinterpret_cast<const f16x8*>(base + w_off + 1024);
    const f16x8 x0 = *reinterpret_cast<const f16x8*>(base + p_off);
    const f16x8 x1 = *reinterpret_cast<const f16x8*>(base + p_off + 1024);
    const f16x8 w2 = *reinterpret_cast<const f16x8*>(base + S_SLOT + w_off);
    const f16x8 w3 = *reinterpret_cast<const f16x8*>(base + S_SLOT + w_off + 1024);
    const f16x8 x2 = *reinterpret_cast<const f16x8*>(base + S_SLOT + p_off);
    const f16x8 x3 = *reinterpret_cast<const f16x8*>(base + S_SLOT + p_off + 1024);
    // pair t + NS - 1 goes into the slot of pair t - 1, which every wave finished reading before the previous barrier
    if (t + NS - 1 < NPAIR) {
      stage_pair();
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(4 * (NS - 2)) : "memory");     // my pieces of pair t + 1 landed
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x0, acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x0, acc[1][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, x1, acc[0][1], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x1, acc[1][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, x2, acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3, x2, acc[1][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, x3, acc[0][1], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3, x3, acc[1][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane (fq, fr) owns couts n0 + wc*32 + fq*8 + [0,8) of pixels m0 + wp*32 + q*16 + fr ----
  const int co = n0 + wc * 32 + fq * 8;
  if (co >= p.Cout) return;
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
    const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
    bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  auto finish = [&](auto act_tag) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = m0 + wp * 32 + q * 16 + fr;
      if (m >= p.M) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[0][q][r] + bv[r];
        v[4 + r] = acc[1][q][r] + bv[4 + r];
      }
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
      for (int r = 0; r < 8; ++r) o[r] = (half_t)s_act<ACT>(v[r]);
      *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.out_ld + co) = o;
    }
  };
  if (p.act == 1) finish(std::integral_constant<int, 1>{});
  else if (p.act == 2) finish(std::integral_constant<int, 2>{});
  else finish(std::integral_constant<int, 0>{});
}

}  // namespace

bool conv_igemm_s64_supported(const ConvParams& p) {
  return p.Cout % 8 == 0 && p.ps_cout == 0 && p.out2 == nullptr && p.next_w == nullptr && p.Cin % SKS == 0 &&
         (!p.in2 || p.Cin2 % SKS == 0);
}

// kg: channel slabs (32 ch) per K-walk group, 0 = default (256-channel groups, as the other kernels)
int launch_conv_igemm_s64(ConvParams p, hipStream_t stream, int kg) {
  EMP_REQUIRE(conv_igemm_s64_supported(p), "conv s64: unsupported shape (Cin=%d Cout=%d)", p.Cin, p.Cout);
  constexpr int NS = 5;      // ring of 5 pairs of K-tiles = 80 KiB: 4 pairs (64 KiB) in flight, two workgroups per CU
  const int CB = p.Cin / SKS, KT = p.KH * p.KW;
  if (kg == 0) kg = 8;
  if (KT == 1 || kg > CB || CB % kg != 0) kg = CB;
  p.kgroup = kg;
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv_igemm_s64_kernel<NS>), NS * 2 * S_SLOT)) return rc;
  p.mt = cdiv(p.M, 64);
  p.nt = cdiv(p.Cout, 64);
  p.mt_per_xcd = cdiv(p.mt, 8);
  const int grid = 8 * p.mt_per_xcd * p.nt;
  hipLaunchKernelGGL(conv_igemm_s64_kernel<NS>, dim3(grid), dim3(256), NS * 2 * S_SLOT, stream, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
