// Split-fp16 convolution of the `fp16x3` precision mode (round 5; emp_pdl_set_precision(net, 2) / EMP_PRECISION=fp16x3).
//
// The reference computes this path in fp32 (empanada/inference/engines.py:248-255).  The fp16 engine is within 1e-3 of it
// in rms only, and not for every weight draw (tests/test_gpu_parity_stats.py); the fp32 reference mode (ref32.hip) is
// within 1e-4 in the max norm but runs on the exact fp32 matrix pipe at 1/16 of the fp16 rate.  This kernel is the fp32
// mode's convolution on the FP16 matrix pipe: same operands (NHWC fp32 maps, fp32 weights [Cout][KH*KW][Cin16]), same
// epilogue, same launch contract (`Conv32`); every operand is split into an fp16 pair,
//     x = hi + lo,   hi = fp16(x),   lo = fp16(x - hi)          (22 significant bits: 2^-22 relative)
// and every product is three MFMAs into one fp32 accumulator:  w_lo . x_hi  +  w_hi . x_lo  +  w_hi . x_hi  (the
// w_lo . x_lo term is 2^-22 of the product: dropped).  3 x 1/16 of the fp32 pipe's time for the same result to ~1e-6
// relative per term -- the heads stay within 1e-3 of the fp32 forward in the MAX norm (tests/test_gpu_fp16x3.py: 2e-5).
//
// Tile: 128 pixels x BN couts (128 or 64), K walked tap-major in steps of 32 channels (two
// 16-channel chunks, each inside one tap because Cin16 % 16 == 0), four waves; A = weights, B = pixels, so a lane's four
// accumulator values are four consecutive couts of one pixel (float4 stores).  Operands are staged global -> registers
// (prefetched two steps ahead) -> split -> LDS (two buffers of fp16 hi / lo tiles, one barrier per step; 64-byte rows with an
// XOR chunk swizzle: conflict-free ds_read_b128 fragments and 128 contiguous bytes per ds_write_b128 lane group).  Activations are split in the kernel (5 vector ops per element); weights arrive either as fp32 (the C ABI's
// emp_conv2d_nhwc_f16x3) or already split -- `Conv32::wpair`, one uint32 per weight = hi | lo << 16, made once at
// emp_pdl_finalize by split_pairs -- and then only change lanes (v_perm).
// Values must fit fp16's range (|x| <= 65504), as every map of the fp16 engine does.
#include <type_traits>

#include "common.h"

namespace emp {
namespace {

constexpr int X_BK = 32, X_LD = 32;      // LDS rows are the step's 32 halfs = 64 B, four 16-byte chunks

template <int ACT>
__device__ __forceinline__ float x_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

// 16-byte chunk c of tile row r lives at chunk position c ^ swz(r): with the fragment addressing below (lane -> row
// lane % 16, chunk lane / 16) every one of gfx950's four non-contiguous ds_read_b128 lane groups touches 64 distinct
// banks -- the image of conv_igemm256.hip.  (Round 5's first layout -- rows padded to 80 B -- measured 50 % of its LDS
// cycles as bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, tools/conv16x3_pmc.sh.)
__device__ __forceinline__ int swz(int r) { return (-((r & 15) >> 2)) & 3; }

// workgroup barrier that orders LDS traffic only: a __syncthreads() would also drain vmcnt, i.e. wait for the global loads
// that were issued precisely to stay in flight across it (the prefetch of the next steps)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 fp32 values -> 8 hi halfs + 8 lo halfs, one 16-byte LDS store each
__device__ __forceinline__ void split_store(const float4 (&r)[2], half_t* hi, half_t* lo) {
  f16x8 h, l;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half_t a = (half_t)v[e];
      h[q * 4 + e] = a;
      l[q * 4 + e] = (half_t)(v[e] - (float)a);
    }
  }
  *reinterpret_cast<f16x8*>(hi) = h;
  *reinterpret_cast<f16x8*>(lo) = l;
}

// 8 pre-split weights (uint32 = hi | lo << 16, carried in float4 registers) -> the same two stores
__device__ __forceinline__ void pair_store(const float4 (&r)[2], half_t* hi, half_t* lo) {
  uint4 h, l;
  const uint32_t p0 = __float_as_uint(r[0].x), p1 = __float_as_uint(r[0].y), p2 = __float_as_uint(r[0].z), p3 = __float_as_uint(r[0].w);
  const uint32_t p4 = __float_as_uint(r[1].x), p5 = __float_as_uint(r[1].y), p6 = __float_as_uint(r[1].z), p7 = __float_as_uint(r[1].w);
  h.x = __builtin_amdgcn_perm(p1, p0, 0x05040100u); h.y = __builtin_amdgcn_perm(p3, p2, 0x05040100u);
  h.z = __builtin_amdgcn_perm(p5, p4, 0x05040100u); h.w = __builtin_amdgcn_perm(p7, p6, 0x05040100u);
  l.x = __builtin_amdgcn_perm(p1, p0, 0x07060302u); l.y = __builtin_amdgcn_perm(p3, p2, 0x07060302u);
  l.z = __builtin_amdgcn_perm(p5, p4, 0x07060302u); l.w = __builtin_amdgcn_perm(p7, p6, 0x07060302u);
  *reinterpret_cast<uint4*>(hi) = h;
  *reinterpret_cast<uint4*>(lo) = l;
}

// the same for FOUR values: one 8-byte LDS store each (the split-role kernel's whole-line pieces)
__device__ __forceinline__ void split_store4(const float4& r, half_t* hi, half_t* lo) {
  const float v[4] = {r.x, r.y, r.z, r.w};
  f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const half_t a = (half_t)v[e];
    h[e] = a;
    l[e] = (half_t)(v[e] - (float)a);
  }
  *reinterpret_cast<f16x4*>(hi) = h;
  *reinterpret_cast<f16x4*>(lo) = l;
}
__device__ __forceinline__ void pair_store4(const float4& r, half_t* hi, half_t* lo) {
  const uint32_t p0 = __float_as_uint(r.x), p1 = __float_as_uint(r.y), p2 = __float_as_uint(r.z), p3 = __float_as_uint(r.w);
  uint2 h, l;
  h.x = __builtin_amdgcn_perm(p1, p0, 0x05040100u); h.y = __builtin_amdgcn_perm(p3, p2, 0x05040100u);
  l.x = __builtin_amdgcn_perm(p1, p0, 0x07060302u); l.y = __builtin_amdgcn_perm(p3, p2, 0x07060302u);
  *reinterpret_cast<uint2*>(hi) = h;
  *reinterpret_cast<uint2*>(lo) = l;
}

// four consecutive couts of pixel m: one 16-byte store into an fp32 row, or -- Conv32::out_fmt, round 6 -- 8 B of hi and 8 B of lo
// into an hl32 row (the plane region's format, conv16x3p.hip: the consumer's operand split done here, once)
template <int ACT>
__device__ __forceinline__ void store4(const Conv32& p, int m, int co, const float4& t) {
  const float v[4] = {x_act<ACT>(t.x), x_act<ACT>(t.y), x_act<ACT>(t.z), x_act<ACT>(t.w)};
  if (p.out2 && co >= p.split2) {      // second destination (two convolutions of one map as one launch: the decoders' low-level projections)
    *reinterpret_cast<float4*>(p.out2 + (size_t)m * p.out2_ld + (co - p.split2)) = make_float4(v[0], v[1], v[2], v[3]);
    return;
  }
  if (p.out_fmt) {
    half_t* op = reinterpret_cast<half_t*>(p.out) + (size_t)m * (2 * p.out_ld) + (co >> 5) * 64 + (co & 31);
    f16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (half_t)v[e];
      l[e] = (half_t)(v[e] - (float)h[e]);
    }
    *reinterpret_cast<f16x4*>(op) = h;
    *reinterpret_cast<f16x4*>(op + 32) = l;
  } else {
    *reinterpret_cast<float4*>(p.out + (size_t)m * p.out_ld + co) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int ACT, int BM, int BN, bool WPAIR, bool DB, bool HEAD = false, bool IN2 = false>
__global__ void __launch_bounds__(256, DB ? 2 : 3) conv16x3_kernel(const Conv32 p) {
  constexpr int WC = BN / 64;          // waves along the couts (64 couts each)
  constexpr int WP = 4 / WC;           // waves along the pixels
  constexpr int NJ = BM / WP / 16;     // 16-pixel fragments per wave
  constexpr int NXS = BM / 64;         // (row, 8-channel chunk) pieces of the pixel tile per thread and step
  constexpr int NWS = BN / 64;         // ... of the weight tile
  // DB: two LDS buffers: step k computes from buffer k % 2 while the operands of step k + 1 (loaded during step k - 1) are
  // split into the other one and the loads of step k + 2 are in flight -- one barrier per step, two steps of latency
  // cover.  DB = false (short K: the launch is its prologue and epilogue): one buffer, two barriers per step, three
  // workgroups per CU
  constexpr int NB = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) half_t Xh[NB][BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Xl[NB][BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wh[NB][BN * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wl[NB][BN * X_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware raster (consecutive workgroup ids go round-robin over the 8 XCDs): XCD x owns the pixel tiles
  // [x * mtx, (x + 1) * mtx) and walks them with the cout tile fastest, so the nt workgroups that read the same pixel tile
  // run side by side on ONE XCD and share it through that XCD's L2 (with blockIdx.y = cout tile they were a whole grid row
  // apart: every cout tile re-read its pixels from HBM)
  // x3_mtx == 0 (short K or one cout tile, where the raster measured slower): pixel tile fastest, the cout tiles a grid row apart
  const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
  const int mtile = p.x3_mtx ? xcd * p.x3_mtx + jj / p.x3_nt : bid % p.x3_mt;
  const int ntile = p.x3_mtx ? jj % p.x3_nt : bid / p.x3_mt;
  if (mtile >= p.x3_mt) return;
  const int m0 = mtile * BM, n0 = ntile * BN;
  // grouped convolution (RegNet's 3x3): blockIdx.z = group, as conv32_kernel (ref32.hip)
  const int g = blockIdx.z, gco = g * p.Cout;
  const float* gin = p.in + (size_t)g * p.cin_g;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.N * HoWo;
  const int K1 = p.KH * p.KW * p.Cin;                 // K of the main source
  const int K = K1 + (IN2 ? p.Cin2 : 0);              // + the second source's channels (Conv32::in2; instantiations of their own)
  // staging roles: piece s of thread t = row t / 4 + 64 s, 8-channel chunk t % 4 -- eight consecutive lanes store 128
  // contiguous bytes of two rows (the ds_write_b128 lane group), four lanes load 128 contiguous bytes of one row
  const int srow = tid >> 2, sch = tid & 3;
  int an[NXS], aoy[NXS], aox[NXS];
  bool xok[NXS];
#pragma unroll
  for (int s2 = 0; s2 < NXS; ++s2) {
    const int am = m0 + srow + 64 * s2;
    xok[s2] = am < M;
    an[s2] = aoy[s2] = aox[s2] = 0;
    if (xok[s2]) {
      an[s2] = am / HoWo;
      const int r = am - an[s2] * HoWo;
      aoy[s2] = r / p.Wo;
      aox[s2] = r - aoy[s2] * p.Wo;
    }
  }
  const float* wrow[NWS];
#pragma unroll
  for (int s2 = 0; s2 < NWS; ++s2) {
    const int co = n0 + srow + 64 * s2;
    const float* base = WPAIR ? reinterpret_cast<const float*>(p.wpair) : p.w;
    wrow[s2] = co < p.Cout ? base + (size_t)(gco + co) * K + sch * 8 : nullptr;
  }
  const int wc = (wave % WC) * 64, wp = (wave / WC) * (NJ * 16);
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (a second register set -- loads issued three steps ahead instead of two -- measured 3 % slower: 204 VGPRs)
  constexpr int NR = 1;
  float4 rx[NR][NXS][2], rw[NR][NWS][2];
  // the loader walks K in order (every call is the next step): tap and channel of this thread's chunk advance by
  // increments -- no division inside the loop (round 5: the two integer divisions and the 64-bit index arithmetic of a
  // per-step recomputation were a third of the kernel's vector instructions)
  int g_k = sch * 8, g_c0 = sch * 8, g_ky = 0, g_kx = 0;
  while (g_c0 >= p.Cin) { g_c0 -= p.Cin; if (++g_kx == p.KW) { g_kx = 0; ++g_ky; } }      // (a chunk of 8 channels lies inside one tap: Cin % 16 == 0)
  int oy0[NXS], ox0[NXS], pix0[NXS], pix2[NXS];
#pragma unroll
  for (int s2 = 0; s2 < NXS; ++s2) {
    oy0[s2] = aoy[s2] * p.stride - p.pad;
    ox0[s2] = aox[s2] * p.stride - p.pad;
    pix0[s2] = an[s2] * p.H * p.W;
    pix2[s2] = IN2 ? (an[s2] * p.H2 + aoy[s2] * p.stride2) * p.W2 + aox[s2] * p.stride2 : 0;
  }
  auto gload = [&](int k0, auto RS) {
    constexpr int rs = decltype(RS)::value;
    // every piece is loaded unconditionally: an out-of-range one (padding tap, row beyond M / Cout, K tail) reads the zero
    // page instead -- no zero-filled registers, no divergent branch around the loads
    const bool kok = g_k < K;
    if (!IN2 || g_k < K1) {
#pragma unroll
      for (int s2 = 0; s2 < NXS; ++s2) {
        const int iy = oy0[s2] + g_ky * p.dil, ix = ox0[s2] + g_kx * p.dil;
        const bool ok = xok[s2] && (IN2 || kok) && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float4* src = reinterpret_cast<const float4*>(ok ? gin + (size_t)(pix0[s2] + iy * p.W + ix) * p.in_ld + g_c0 : p.zero);
        rx[rs][s2][0] = src[0];
        rx[rs][s2][1] = src[1];
      }
    } else {      // the second source (Conv32::in2): its pixel at stride2, channel g_k - K1; or the K tail: zeros
#pragma unroll
      for (int s2 = 0; s2 < NXS; ++s2) {
        const float4* src = reinterpret_cast<const float4*>((xok[s2] && kok) ? p.in2 + (size_t)pix2[s2] * p.in2_ld + (g_k - K1) : p.zero);
        rx[rs][s2][0] = src[0];
        rx[rs][s2][1] = src[1];
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < NWS; ++s2) {
      const float4* src = reinterpret_cast<const float4*>((wrow[s2] && kok) ? wrow[s2] + k0 : p.zero);
      rw[rs][s2][0] = src[0];
      rw[rs][s2][1] = src[1];
    }
    g_k += X_BK;
    g_c0 += X_BK;
    while (g_c0 >= p.Cin) { g_c0 -= p.Cin; if (++g_kx == p.KW) { g_kx = 0; ++g_ky; } }
  };
  const int soff = srow * X_LD + ((sch ^ swz(srow)) << 3);      // (row + 64 s keeps the row's swizzle key)
  auto stage = [&](int b, auto RS) {
    constexpr int rs = decltype(RS)::value;
#pragma unroll
    for (int s2 = 0; s2 < NXS; ++s2) split_store(rx[rs][s2], Xh[b] + soff + s2 * 64 * X_LD, Xl[b] + soff + s2 * 64 * X_LD);
#pragma unroll
    for (int s2 = 0; s2 < NWS; ++s2) {
      if (WPAIR) pair_store(rw[rs][s2], Wh[b] + soff + s2 * 64 * X_LD, Wl[b] + soff + s2 * 64 * X_LD);
      else split_store(rw[rs][s2], Wh[b] + soff + s2 * 64 * X_LD, Wl[b] + soff + s2 * 64 * X_LD);
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * X_LD + ((fq ^ swz(fr)) << 3);           // fragment: row fr of a 16-row block, chunk fq
  using R0 = std::integral_constant<int, 0>;
  auto mma = [&](int b) {
    // the wave's weight fragments stay in registers for the step (32 VGPRs); the pixel fragments stream through
    f16x8 wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const f16x8*>(Wh[b] + (wc + i * 16) * X_LD + foff);
      wl[i] = *reinterpret_cast<const f16x8*>(Wl[b] + (wc + i * 16) * X_LD + foff);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f16x8 xh = *reinterpret_cast<const f16x8*>(Xh[b] + (wp + j * 16) * X_LD + foff);
      const f16x8 xl = *reinterpret_cast<const f16x8*>(Xl[b] + (wp + j * 16) * X_LD + foff);
      // consecutive MFMAs write different accumulators (a dependent one would wait for its predecessor's passes)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl, acc[i][j], 0, 0, 0);
    }
  };
  if (DB) {
    // step k computes from LDS buffer k % 2 while the registers (step k + 1, loaded during step k - 1) are split into the
    // other buffer and the loads of step k + 2 go out: one barrier per step, two steps of MFMAs between a load and its use
    gload(0, R0());
    stage(0, R0());
    if (X_BK < K) gload(X_BK, R0());
    int b = 0;
    for (int k0 = 0; k0 < K; k0 += X_BK, b ^= 1) {
      lds_barrier();        // buffer b is complete; nobody reads buffer b ^ 1 (step k - 1) any more (the loads stay in flight)
      if (k0 + X_BK < K) {
        stage(b ^ 1, R0());
        if (k0 + 2 * X_BK < K) gload(k0 + 2 * X_BK, R0());
      }
      mma(b);
    }
  } else {
    gload(0, R0());
    for (int k0 = 0; k0 < K; k0 += X_BK) {
      lds_barrier();        // the previous step's fragment reads are done
      stage(0, R0());
      lds_barrier();
      if (k0 + X_BK < K) gload(k0 + X_BK, R0());
      mma(0);
    }
  }
  if (HEAD) {
    // fused 1x1 head: per pixel, head_c dot products of the activated outputs with head_w over this workgroup's couts.
    // Fixed order: a lane's 16 couts ascending, the four lane groups by two butterfly steps, the cout-halves of the tile
    // (waves wc = 0, 64) in ascending order through LDS, the cout tiles in launch_head_finish_f32 -- deterministic.
    __syncthreads();                       // the last step's fragment reads are done: the X tiles become scratch
    float* red = reinterpret_cast<float*>(&Xh[0][0]);      // [WC][BM][4] floats = 4 KiB per cout half
    const int hcn = p.head_c;
    float bi[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) bi[i][e] = (p.bias && col + e < p.Cout) ? p.bias[gco + col + e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = col + e < p.Cout ? x_act<ACT>(acc[i][j][e] + bi[i][e]) : 0.f;
          const int cw = col + e < p.Cout ? col + e : 0;
#pragma unroll
          for (int h = 0; h < 4; ++h)
            if (h < hcn) sum[h] = fmaf(v, p.head_w[(size_t)h * p.Cout + cw], sum[h]);
        }
      }
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        sum[h] += __shfl_xor(sum[h], 16);
        sum[h] += __shfl_xor(sum[h], 32);
      }
      if (lane < 16) {
#pragma unroll
        for (int h = 0; h < 4; ++h) red[((wave % WC) * BM + wp + j * 16 + fr) * 4 + h] = sum[h];
      }
    }
    __syncthreads();
    if (tid < BM) {
      const int m = m0 + tid;
      if (m < M) {
        for (int h = 0; h < hcn; ++h) {
          float t = red[tid * 4 + h];
          if (WC == 2) t += red[(BM + tid) * 4 + h];
          p.head_part[((size_t)ntile * M + m) * hcn + h] = t;
        }
      }
    }
    return;
  }
  // epilogue (conv32_kernel's): bias (+ per-image bias) (+ residual), activation, store (NHWC slice or k2s2 pixel shuffle).
  // acc[i][j][e] <-> cout n0 + wc + 16 i + 4 (lane / 16) + e, pixel m0 + wp + 16 j + lane % 16
  const bool vec = p.ps_cout == 0 && (p.Cout & 3) == 0 && (gco & 3) == 0 && (p.out_ld & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                   (!p.bias || (((uintptr_t)p.bias) & 15) == 0) && (!p.bias_n || (((uintptr_t)p.bias_n) & 15) == 0) &&
                   (!p.res || ((p.res_ld & 3) == 0 && (((uintptr_t)p.res) & 15) == 0));
  if (vec) {
    // four consecutive couts per lane: every operand of the epilogue is one 16-byte access, all of them independent
    float4 bi[4];
    bool cok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
      cok[i] = col < p.Cout;
      bi[i] = (cok[i] && p.bias) ? *reinterpret_cast<const float4*>(p.bias + gco + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + wp + j * 16 + fr;
      if (m >= M) continue;
      const int n = m / HoWo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!cok[i]) continue;
        const int co = gco + n0 + wc + i * 16 + (lane >> 4) * 4;
        float4 t = make_float4(acc[i][j][0] + bi[i].x, acc[i][j][1] + bi[i].y, acc[i][j][2] + bi[i].z, acc[i][j][3] + bi[i].w);
        if (p.bias_n) {
          const float4 b = *reinterpret_cast<const float4*>(p.bias_n + (size_t)n * p.Cout + co);
          t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        if (p.res) {
          const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + co);
          t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
        }
        store4<ACT>(p, m, co, t);
      }
    }
    return;
  }
  // ragged channel counts, unaligned slices, the pixel-shuffle store: element by element
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wp + j * 16 + fr;
    if (m >= M) continue;
    const int n = m / HoWo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
      if (col >= p.Cout) continue;
      const int r = m - n * HoWo;
      const int y = r / p.Wo, x = r - y * p.Wo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (col + e >= p.Cout) continue;
        const int co = gco + col + e;
        float t = acc[i][j][e] + (p.bias ? p.bias[co] : 0.f);
        if (p.bias_n) t += p.bias_n[(size_t)n * p.Cout + co];
        if (p.res) t += p.res[(size_t)m * p.res_ld + co];
        t = x_act<ACT>(t);
        if (p.ps_cout == 0) {
          p.out[(size_t)m * p.out_ld + co] = t;
        } else {      // ConvTranspose2d(k=2, s=2) as four sub-pixel 1x1 convs: (n,y,x,q*C+c) -> (n, 2y+dy, 2x+dx, c)
          const int q = co / p.ps_cout, c = co - q * p.ps_cout;
          p.out[(((size_t)n * (2 * p.Ho) + 2 * y + (q >> 1)) * (2 * p.Wo) + 2 * x + (q & 1)) * p.out_ld + c] = t;
        }
      }
    }
  }
}

// ---- long-K variant with split roles (round 5) ---------------------------------------------------------------------
// 128 pixels x 128 couts, EIGHT waves: waves 0-3 stage (global -> registers -> split -> LDS), waves 4-7 only read fragments
// and issue MFMAs -- one of each per SIMD, so the split's vector work and LDS stores run beside the matrix pipe instead of in
// front of it.  Three LDS slots of 32 KiB (hi / lo tiles of pixels and weights); in step k the stagers fill slot (k + 2) % 3
// (read last in step k - 1) while the multipliers consume slot k % 3; ONE barrier per step.  One workgroup per CU.
constexpr int XS_TILE = 128 * X_LD;                     // halfs per tile
constexpr int XS_SLOT_HALFS = 4 * XS_TILE;

// WIMG (round 5, late): the weights come from a packed IMAGE (Conv32::wimg, x3_weight_image_kernel: per cout tile and step the
// hi tile and the lo tile exactly as they lie in LDS, 16 KiB contiguous) by LDS-DMA, three steps deep in a ring of their own --
// no registers, no split, no LDS stores for half of the operand bytes, and twice the bytes in flight (finding 51: register
// staging bounded them).  LDS then: pixel slots 2 x [Xh | Xl] + cout slots 3 x [Wh | Wl] = 80 KiB, two workgroups per CU.
// KSPLIT (round 6, late): split s = XCD % S walks steps [s * x3_ksteps, ...) of the K loop only and stores raw partial sums (Conv32::kpart)
template <int ACT, bool WPAIR, int XS_SLOTS, bool WIMG = false, bool KSPLIT = false>
__global__ void __launch_bounds__(512, XS_SLOTS == 2 ? 4 : 2) conv16x3s_kernel(const Conv32 p) {
  static_assert(!WIMG || XS_SLOTS == 2, "the image variant has two pixel slots");
  extern __shared__ __attribute__((aligned(1024))) char xs_lds[];
  half_t* const L = reinterpret_cast<half_t*>(xs_lds);      // slot s: Xh | Xl | Wh | Wl at L + s * XS_SLOT_HALFS
  // (WIMG: the cout ring is an LDS object of its own, so that the compiler can tell the stagers' ds_writes -- pixel slots -- from
  // the LDS-DMA's destination: with one array it drained vmcnt in front of every staging pass)
  __shared__ __attribute__((aligned(1024))) half_t xs_wring[WIMG ? 3 * 2 * XS_TILE : 8];
  auto xslot = [&](int s) { return WIMG ? L + s * 2 * XS_TILE : L + s * XS_SLOT_HALFS; };                                   // Xh | Xl
  auto wslot = [&](int s) { return WIMG ? xs_wring + s * 2 * XS_TILE : L + s * XS_SLOT_HALFS + 2 * XS_TILE; };            // Wh | Wl
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool stager = wave < 4;
  const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
  int mtile = p.x3_mtx ? xcd * p.x3_mtx + jj / p.x3_nt : bid % p.x3_mt;
  int ntile = p.x3_mtx ? jj % p.x3_nt : bid / p.x3_mt;
  int ksplit_idx = 0;
  if constexpr (KSPLIT) {      // S in {2, 4, 8}: a split (its K range of the weights, its channel range of the map) lives on 8 / S XCDs
    const int S = p.ksplit;
    ksplit_idx = xcd % S;
    const int tile = jj * (8 / S) + xcd / S;
    mtile = tile / p.x3_nt;
    ntile = tile - mtile * p.x3_nt;
  }
  if (mtile >= p.x3_mt) return;
  const int m0 = mtile * 128, n0 = ntile * 128;
  const int g = blockIdx.z, gco = g * p.Cout;
  const float* gin = p.in + (size_t)g * p.cin_g;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.N * HoWo;
  const int K = p.KH * p.KW * p.Cin;
  const int nsteps_all = (K + X_BK - 1) / X_BK;
  const int t0 = KSPLIT ? ksplit_idx * p.x3_ksteps : 0;                                       // first step of this workgroup
  const int nsteps = KSPLIT ? min(p.x3_ksteps, nsteps_all - t0) : nsteps_all;                 // ... and how many it walks
  if (stager) {
    // ---- waves 0-3: global -> registers -> split -> LDS.  Piece i of thread t = row t / 8 + 32 i, FOUR-channel chunk t % 8:
    // eight consecutive lanes read one whole 128-byte line (a row's 32 fp32 channels of the step).  (The four-wave kernel's
    // piece -- 8 channels as two 16-byte loads, four lanes per row -- touches every line in two instructions, half of it
    // each: the texture path then spends its cycles on the line, not the bytes: 33 TD-busy cycles per load instruction
    // measured against 16 for whole lines, and that path, not the matrix pipe, is what this kernel waits for.)
    const int srow = tid >> 3, sc4 = tid & 7;
    const float* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = n0 + srow + 32 * i;
      const float* base = WPAIR ? reinterpret_cast<const float*>(p.wpair) : p.w;
      wrow[i] = co < p.Cout ? base + (size_t)(gco + co) * K + sc4 * 4 + (size_t)t0 * X_BK : nullptr;
    }
    // TWO register sets: the loads of step k + A + 1 and k + A + 2 are in flight while step k + A is split and stored (the
    // stagers' registers are free -- the allocation is the multipliers' -- and one step of the multipliers is shorter than a
    // loaded HBM round trip).  Every step issues its eight loads unconditionally (past K: the zero page), so that the wait
    // in front of a split is a counted vmcnt(8), not a drain.
    // Addresses advance by increments (Cin % 32 == 0 here: a tap is a whole number of steps and the tap changes at the same
    // step in every lane, i.e. on scalar counters): one 64-bit add per pointer and step instead of the tap arithmetic --
    // the stagers share their SIMD's issue port with the multipliers' MFMAs.
    float4 rx[2][4], rw[2][4];
    const char* px[4];
    const char* pw[4];
    uint32_t dx[4], dw[4];                      // bytes per step: 128, or 0 on the zero page
    const int csteps = p.Cin / X_BK;            // steps per tap
    int c_left = csteps, g_t = 0, g_ky = 0, g_kx = 0;
    int c_first = 0;                            // KSPLIT: the channel the first tap of this split starts at
    if constexpr (KSPLIT) {
      const int tap0 = t0 / csteps, s_in = t0 - tap0 * csteps;
      g_ky = tap0 / p.KW; g_kx = tap0 - g_ky * p.KW;
      c_left = csteps - s_in;
      c_first = s_in * X_BK;
    }
    // (the pixel's coordinates are recomputed at every tap change -- KH * KW times per workgroup -- instead of living in
    // registers through the loop)
    auto set_tap = [&](bool live, int c0 = 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int sr = srow;
        asm volatile("" : "+v"(sr));      // (nothing of this hoisted out of the loop as an invariant: it would be spilled there)
        const int am = m0 + sr + 32 * i;
        const int an = am / HoWo, r = am - an * HoWo;
        const int aoy = r / p.Wo, aox = r - aoy * p.Wo;
        const int iy = aoy * p.stride - p.pad + g_ky * p.dil, ix = aox * p.stride - p.pad + g_kx * p.dil;
        const bool ok = live && am < M && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        int sc = sc4 * 4 + c0;
        asm volatile("" : "+v"(sc));
        px[i] = reinterpret_cast<const char*>(ok ? gin + ((size_t)((an * p.H + iy) * p.W + ix) * p.in_ld + sc) : p.zero);
        dx[i] = ok ? X_BK * 4u : 0u;
      }
    };
    set_tap(true, c_first);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pw[i] = reinterpret_cast<const char*>(wrow[i] ? wrow[i] : p.zero);
      dw[i] = wrow[i] ? X_BK * 4u : 0u;
    }
    auto gload = [&](auto SET) {
      constexpr int R = decltype(SET)::value;
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 1      // diagnostic build: no global loads (stale registers)
      ++g_t;
      return;
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rx[R][i] = *reinterpret_cast<const float4*>(px[i]);
        px[i] += dx[i];
      }
      if constexpr (!WIMG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          rw[R][i] = *reinterpret_cast<const float4*>(pw[i]);
          pw[i] += dw[i];
        }
      }
      ++g_t;
      if (--c_left == 0) {                      // next tap (wave-uniform)
        c_left = csteps;
        if (++g_kx == p.KW) { g_kx = 0; ++g_ky; }
        set_tap(g_t < nsteps);
      }
      if (g_t == nsteps) {                      // past K: everything reads the zero page from here on
#pragma unroll
        for (int i = 0; i < 4; ++i) { pw[i] = reinterpret_cast<const char*>(p.zero); dw[i] = 0u; }
      }
    };
    // LDS position of the piece: row, 16-byte chunk (sc4 / 2) ^ swz(row), 8-byte half sc4 % 2 (rows 32 i apart share swz)
    const int soff = srow * X_LD + ((((sc4 >> 1) ^ swz(srow)) << 3) | ((sc4 & 1) << 2));
    auto stage = [&](auto SET, int slot) {
      constexpr int R = decltype(SET)::value;
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 2      // diagnostic build: no split, no LDS stores
      asm volatile("" :: "v"(rx[R][0].x), "v"(rx[R][3].w), "v"(rw[R][0].x), "v"(rw[R][3].w));
      return;
#endif
      half_t* const SX = xslot(slot);
#pragma unroll
      for (int i = 0; i < 4; ++i) split_store4(rx[R][i], SX + soff + i * 32 * X_LD, SX + XS_TILE + soff + i * 32 * X_LD);
      if constexpr (!WIMG) {
        half_t* const SW = wslot(slot);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (WPAIR) pair_store4(rw[R][i], SW + soff + i * 32 * X_LD, SW + XS_TILE + soff + i * 32 * X_LD);
          else split_store4(rw[R][i], SW + soff + i * 32 * X_LD, SW + XS_TILE + soff + i * 32 * X_LD);
        }
      }
    };
    // WIMG: this wave's four 1 KiB pieces of step t's [Wh | Wl] image into cout slot t % 3 (t clamped: the last steps' extra
    // issues rewrite a slot nobody reads any more -- every step issues, so the counted waits below are constants).  Buffer
    // form: descriptor + scalar piece offset + ONE lane-offset register (64-bit per-lane addresses made the register
    // allocator move pending pixel loads out of the way, i.e. wait for them, at the top of every iteration)
    const int dma_voff = lane * 16;
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(WIMG ? p.wimg : nullptr), 0, 0x7fffffff, 0x00020000);
    auto dma_w = [&](int t) {
      const int tc = t < nsteps ? t : nsteps - 1;
      const int soff0 = (((ntile * nsteps_all + t0 + tc) * 16 + wave * 4) * 512) * 2;      // bytes (launcher: the image is below 2 GiB)
      half_t* dst = wslot(t % 3) + wave * 4 * 512;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst + i * 512), 16, dma_voff,
                                                 soff0 + i * 1024, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // the stagers run A = XS_SLOTS - 1 steps ahead of the multipliers in LDS and two more steps ahead in registers;
    // step t lives in register set t & 1
    constexpr int A = XS_SLOTS - 1;
    if constexpr (WIMG) dma_w(0);                           // (older than step 0's pixel loads: their wait covers it)
#pragma unroll
    for (int a = 0; a < A; ++a)
      if (a < nsteps) { gload(S0{}); stage(S0{}, a); }
    if constexpr (WIMG) dma_w(1);
    // sets: step A -> set A & 1, step A + 1 -> the other one
    if ((A & 1) == 0) { gload(S0{}); gload(S1{}); }
    else { gload(S1{}); gload(S0{}); }
    lds_barrier();
    // main loop without a condition inside (the compiler's counted waits need the same loads pending on every path:
    // with a guarded body it drains to vmcnt(0) in front of every split); the tail stages what is left and keeps the
    // multipliers' barrier count
    using SA = std::integral_constant<int, A & 1>;          // set of step k + A for even k
    using SB = std::integral_constant<int, (A + 1) & 1>;
    const int kst = nsteps - A;                             // steps k that still stage one
    int k = 0;
    // WIMG: iteration k issues the cout image of step k + 2 FIRST (its slot was read last in step k - 1), then stages and
    // loads pixels; before the barrier that opens step k + 1 the image of step k + 1 (first thing issued in iteration
    // k - 1) must have landed: younger than it are iteration k - 1's 4 pixel loads and iteration k's 4 + 4 = vmcnt(12)
    for (; k + 1 < kst; k += 2) {
      if constexpr (WIMG) dma_w(k + 2);
      stage(SA{}, (k + A) % XS_SLOTS);                      // its slot was read last in step k - 1
      gload(SA{});                                          // step k + A + 2 (past K: zeros, never staged)
      if constexpr (WIMG) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);                    // (no split arithmetic hoisted over the barrier: 128 registers)
      if constexpr (WIMG) dma_w(k + 3);
      stage(SB{}, (k + 1 + A) % XS_SLOTS);
      gload(SB{});
      if constexpr (WIMG) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      lds_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (WIMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail: everything issued has landed
    if (k < kst) {
      stage(SA{}, (k + A) % XS_SLOTS);
      lds_barrier();
      ++k;
    }
    for (; k < nsteps; ++k) lds_barrier();
    return;
  }
  // ---- waves 4-7: fragments and MFMAs ----
  const int cw = wave - 4;
  const int wc = (cw & 1) * 64, wp = (cw >> 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * X_LD + ((fq ^ swz(fr)) << 3);
  lds_barrier();
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 4      // diagnostic build: no fragment reads (loop-invariant operands)
  f16x8 abw, abx;
#pragma unroll
  for (int e = 0; e < 8; ++e) { abw[e] = (half_t)(float)(lane + e); abx[e] = (half_t)(float)(lane - e); }
  asm volatile("" : "+v"(abw), "+v"(abx));
#endif
  for (int k = 0; k < nsteps; ++k) {
    const half_t* const SX = xslot(k % XS_SLOTS);
    const half_t* const SW = wslot(WIMG ? k % 3 : k % XS_SLOTS);
    f16x8 wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 4
      wh[i] = abw; wl[i] = abx;
#else
      wh[i] = *reinterpret_cast<const f16x8*>(SW + (wc + i * 16) * X_LD + foff);
      wl[i] = *reinterpret_cast<const f16x8*>(SW + XS_TILE + (wc + i * 16) * X_LD + foff);
#endif
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 4
      const f16x8 xh = abx, xl = abw;
#else
      const f16x8 xh = *reinterpret_cast<const f16x8*>(SX + (wp + j * 16) * X_LD + foff);
      const f16x8 xl = *reinterpret_cast<const f16x8*>(SX + XS_TILE + (wp + j * 16) * X_LD + foff);
#endif
#if defined(EMP_X3_ABLATE) && EMP_X3_ABLATE == 3      // diagnostic build: fragment reads, no MFMAs
      asm volatile("" :: "v"(xh), "v"(xl), "v"(wh[j]), "v"(wl[j]));
#else
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl, acc[i][j], 0, 0, 0);
#endif
    }
    lds_barrier();
  }
  if constexpr (KSPLIT) {      // raw partial sums of this K range; ksplit_finish32_kernel does the rest
    float* const part = p.kpart + (size_t)ksplit_idx * M * p.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wp + j * 16 + fr;
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = n0 + wc + i * 16 + (lane >> 4) * 4;
        if (co < p.Cout)
          *reinterpret_cast<float4*>(part + (size_t)m * p.Cout + co) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
  // epilogue: the 16-byte form of conv16x3_kernel (the launcher sends everything else to that kernel)
  float4 bi[4];
  bool cok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
    cok[i] = col < p.Cout;
    bi[i] = (cok[i] && p.bias) ? *reinterpret_cast<const float4*>(p.bias + gco + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wp + j * 16 + fr;
    if (m >= M) continue;
    const int n = m / HoWo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!cok[i]) continue;
      const int co = gco + n0 + wc + i * 16 + (lane >> 4) * 4;
      float4 t = make_float4(acc[i][j][0] + bi[i].x, acc[i][j][1] + bi[i].y, acc[i][j][2] + bi[i].z, acc[i][j][3] + bi[i].w);
      if (p.bias_n) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias_n + (size_t)n * p.Cout + co);
        t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
      }
      if (p.res) {
        const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + co);
        t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
      }
      store4<ACT>(p, m, co, t);
    }
  }
}

// can the split-role kernel take this launch?  (its epilogue is the vector form only)
inline bool xs_epilogue_ok(const Conv32& p, int gco_step) {
  return p.ps_cout == 0 && (p.Cout & 3) == 0 && (gco_step & 3) == 0 && (p.out_ld & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
         (!p.bias || (((uintptr_t)p.bias) & 15) == 0) && (!p.bias_n || (((uintptr_t)p.bias_n) & 15) == 0) &&
         (!p.res || ((p.res_ld & 3) == 0 && (((uintptr_t)p.res) & 15) == 0));
}

// out = act(sum over s (ascending) of part[s] + bias + bias_n + res), four couts per thread
template <int ACT>
__global__ void __launch_bounds__(256) ksplit_finish32_kernel(const Conv32 p, int64_t total) {
  const int C4 = p.Cout >> 2, HoWo = p.Ho * p.Wo;
  const int64_t M = (int64_t)p.N * HoWo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / C4;
    const int co = (int)(i - m * C4) * 4;
    float4 t = *reinterpret_cast<const float4*>(p.kpart + (size_t)m * p.Cout + co);
    for (int sidx = 1; sidx < p.ksplit; ++sidx) {
      const float4 u = *reinterpret_cast<const float4*>(p.kpart + ((size_t)sidx * M + m) * p.Cout + co);
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + co);
      t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    if (p.bias_n) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias_n + (size_t)(m / HoWo) * p.Cout + co);
      t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    if (p.res) {
      const float4 r = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.res_ld + co);
      t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
    }
    store4<ACT>(p, (int)m, co, t);
  }
}

template <bool WPAIR, int SLOTS, bool WIMG = false>
int launch_spec(const Conv32& p, dim3 grid, hipStream_t s) {
  constexpr int LDS = WIMG ? 4 * XS_TILE * 2 : SLOTS * XS_SLOT_HALFS * 2;      // dynamic part (WIMG: + 48 KiB static cout ring)
  auto go = [&](auto kern) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(512), LDS, s, p);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  };
  if (p.ksplit > 1) {
    if (int rc = go(&conv16x3s_kernel<0, WPAIR, SLOTS, WIMG, true>)) return rc;
    const int64_t total = (int64_t)p.N * p.Ho * p.Wo * (p.Cout >> 2);
    const dim3 fg((unsigned)std::min<int64_t>((total + 255) / 256, 4096));
    if (p.act == 1) hipLaunchKernelGGL(ksplit_finish32_kernel<1>, fg, dim3(256), 0, s, p, total);
    else if (p.act == 2) hipLaunchKernelGGL(ksplit_finish32_kernel<2>, fg, dim3(256), 0, s, p, total);
    else hipLaunchKernelGGL(ksplit_finish32_kernel<0>, fg, dim3(256), 0, s, p, total);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  if (p.act == 1) return go(&conv16x3s_kernel<1, WPAIR, SLOTS, WIMG>);
  if (p.act == 2) return go(&conv16x3s_kernel<2, WPAIR, SLOTS, WIMG>);
  return go(&conv16x3s_kernel<0, WPAIR, SLOTS, WIMG>);
}

template <int BM, int BN, bool WPAIR, bool DB>
int launch_tile(const Conv32& p, dim3 grid, hipStream_t s) {
  if (p.act == 1) hipLaunchKernelGGL((conv16x3_kernel<1, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  else if (p.act == 2) hipLaunchKernelGGL((conv16x3_kernel<2, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv16x3_kernel<0, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <bool WPAIR>
int launch_pair(Conv32 p, hipStream_t s) {
  const int G = p.groups > 1 ? p.groups : 1;
  const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
  const int bn = p.Cout > 64 ? 128 : 64;
  const unsigned nt = (unsigned)((p.Cout + bn - 1) / bn);
  // (a 256-pixel tile -- half the weight tile's trips through LDS per product -- measured 1.4-2.4x SLOWER on every shape of
  // the network at batch 8: one workgroup per CU leaves nothing to run behind a barrier; profiles/r05_conv16x3.txt)
  p.x3_mt = (int)((M + 127) / 128);
  p.x3_nt = (int)nt;
  const int kk = p.KH * p.KW * p.Cin;
  static const int rk = [] { const char* e = getenv("EMP_X3_RASTER_K"); return e ? atoi(e) : 64; }();      // (round 6: 64 instead of 256 -- +0.3-1 % at batch 16, neutral below)
  p.x3_mtx = (nt >= 2 && kk >= rk) ? (p.x3_mt + 7) / 8 : 0;
  dim3 grid((unsigned)(p.x3_mtx ? 8 * p.x3_mtx * p.x3_nt : p.x3_mt * p.x3_nt), 1u, (unsigned)G);
  static const int kdb = [] { const char* e = getenv("EMP_X3_KDB"); return e ? atoi(e) : 1024; }();     // K from which the two-buffer pipeline runs (measured: profiles/r05_conv16x3.txt)
  if (p.head_w) {      // fused head: one-buffer variant, ReLU (the separable head blocks: K = the decoder width)
    EMP_REQUIRE(p.head_part && p.head_c >= 1 && p.head_c <= 4 && p.act == 1 && G == 1 && !p.res && !p.bias_n && p.ps_cout == 0,
                "conv16x3: fused head needs ReLU, 1..4 head channels, a plain ungrouped convolution");
    if (bn == 128) hipLaunchKernelGGL((conv16x3_kernel<1, 128, 128, WPAIR, false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv16x3_kernel<1, 128, 64, WPAIR, false, true>), grid, dim3(256), 0, s, p);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  if (p.in2) {      // K-concatenated second source: instantiations of their own (conv3 + projection shortcut: ReLU, Cout >= 256)
    EMP_REQUIRE(p.act == 1 && bn == 128 && G == 1 && p.ps_cout == 0, "conv16x3: a second source needs ReLU and Cout > 64");
    if (p.KH * p.KW * p.Cin + p.Cin2 >= kdb) hipLaunchKernelGGL((conv16x3_kernel<1, 128, 128, WPAIR, true, false, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv16x3_kernel<1, 128, 128, WPAIR, false, false, true>), grid, dim3(256), 0, s, p);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  const bool db = p.KH * p.KW * p.Cin >= kdb;
  static const int spec = [] { const char* e = getenv("EMP_X3_SPEC"); return e ? atoi(e) : 2; }();      // 0: the four-wave kernel everywhere; 2 / 3: LDS slots of the split-role kernel (A/B; 3 slots = one workgroup per CU measured slower)
  if (db && spec && bn == 128 && p.Cin % X_BK == 0 && xs_epilogue_ok(p, p.Cout)) {      // (its stagers step whole taps: Cin % 32 == 0)
    static const bool wimg_on = [] { const char* e = getenv("EMP_X3_WIMG"); return !(e && e[0] == '0'); }();      // A/B runs
    // split-K: S workgroups per tile while S * tiles still fit the chip's 256 CUs and every one keeps >= 16 steps (a launch that
    // fills half the chip or more gains nothing: the finish pass costs what the shorter loop saves)
    p.ksplit = 0;
    if (p.kpart && G == 1 && !p.out2 && spec == 2) {
      const int nsteps = kk / X_BK;
      static const int cap = [] { const char* e = getenv("EMP_X3_KSPLIT_WGS"); return e ? atoi(e) : 512; }();      // workgroups a split launch may reach: two per CU (A/B: 256 is 1-5 % slower at batches 1-8)
      const int tiles = p.x3_mt * p.x3_nt;
      static const int min_per = [] { const char* e = getenv("EMP_X3_KSPLIT_PER"); return e ? std::max(atoi(e), 4) : 16; }();      // steps a split keeps at least (A/B)
      int S = std::min(std::min(8, cap / std::max(tiles, 1)), nsteps / min_per);
      S = S >= 8 ? 8 : (S >= 4 ? 4 : (S >= 2 ? 2 : 0));      // (the kernel's raster: a split per 8 / S XCDs)
      static const int min_steps = [] { const char* e = getenv("EMP_X3_KSPLIT_MINSTEPS"); return e ? atoi(e) : 32; }();      // (A/B: 40 and 64 are 3 % slower at one tile per call)
      if (nsteps < min_steps) S = 0;
      if (S >= 2) {
        const int per = (nsteps + S - 1) / S;
        if ((S - 1) * per < nsteps && (int64_t)S * M * p.Cout * 4 <= p.kpart_bytes && ((uintptr_t)p.kpart & 15) == 0) {
          p.ksplit = S; p.x3_ksteps = per;
          grid.x = (unsigned)(8 * ((tiles * S + 7) / 8));
        }
      }
    }
    if (p.wimg && wimg_on && G == 1 && spec == 2 && (int64_t)nt * (kk / X_BK) * 16384 < (1ll << 31)) return launch_spec<WPAIR, 2, true>(p, grid, s);
    return spec == 3 ? launch_spec<WPAIR, 3>(p, grid, s) : launch_spec<WPAIR, 2>(p, grid, s);
  }
  if (db) return bn == 128 ? launch_tile<128, 128, WPAIR, true>(p, grid, s) : launch_tile<128, 64, WPAIR, true>(p, grid, s);
  return bn == 128 ? launch_tile<128, 128, WPAIR, false>(p, grid, s) : launch_tile<128, 64, WPAIR, false>(p, grid, s);
}

// out[n][h][pix] = b[h] + sum over the cout tiles t (ascending) of part[(t * N * P + n * P + pix) * C + h]
__global__ void __launch_bounds__(256) head_finish32_kernel(const float* __restrict__ part, int tiles, int64_t MP, int P, int C,
                                                            const float* __restrict__ b, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < MP * C; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i / MP);
    const int64_t m = i - (int64_t)h * MP;
    float t = b ? b[h] : 0.f;
    for (int k = 0; k < tiles; ++k) t += part[((size_t)k * MP + m) * C + h];
    const int64_t n = m / P, pix = m - n * P;
    out[((size_t)n * C + h) * P + pix] = t;
  }
}

__global__ void __launch_bounds__(256) split_pairs_kernel(const float* __restrict__ w, uint32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = w[i];
    const half_t h = (half_t)x;
    const half_t l = (half_t)(x - (float)h);
    out[i] = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
  }
}

}  // namespace

// the checks of launch_conv32 (ref32.hip) have run: same contract
int launch_conv16x3(const Conv32& p, hipStream_t s) {
  EMP_REQUIRE((int64_t)p.N * p.H * p.W < (1ll << 31), "conv16x3: input map too large for 32-bit pixel indices");
  EMP_REQUIRE(!p.in2 || (p.Cin2 > 0 && p.Cin2 % 16 == 0 && p.in2_ld % 4 == 0 && p.in2_ld >= p.Cin2 && ((uintptr_t)p.in2 % 16) == 0 &&
                         p.stride2 >= 1 && (p.Ho - 1) * p.stride2 < p.H2 && (p.Wo - 1) * p.stride2 < p.W2 && p.groups <= 1 &&
                         (int64_t)p.N * p.H2 * p.W2 < (1ll << 31)),
              "conv16x3: second source: Cin2=%d (multiple of 16), row %d, %dx%d at stride %d must cover the %dx%d output", p.Cin2,
              p.in2_ld, p.H2, p.W2, p.stride2, p.Ho, p.Wo);
  EMP_REQUIRE(p.in_fmt == 0 && p.res_fmt == 0, "conv16x3: hl32 inputs / residuals belong to conv16x3p");
  EMP_REQUIRE(!p.out2 || (p.split2 > 0 && p.split2 % 4 == 0 && p.split2 < p.Cout && !p.out_fmt && !p.head_w && p.ps_cout == 0 && (p.Cout & 3) == 0 &&
                          p.groups <= 1 && (p.out_ld & 3) == 0 && (p.out2_ld & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 && (((uintptr_t)p.out2) & 15) == 0 &&
                          (!p.bias || (((uintptr_t)p.bias) & 15) == 0) && !p.bias_n && !p.res && p.KH * p.KW * p.Cin < 1024),
              "conv16x3: a second destination needs the four-wave kernel's vector epilogue (split %% 4, aligned fp32 rows, K < 1024)");
  EMP_REQUIRE(!p.out_fmt || (!p.head_w && p.ps_cout == 0 && (p.Cout & 3) == 0 && p.groups <= 1 && p.out_ld % 32 == 0 && (((uintptr_t)p.out) & 127) == 0 &&
                             (!p.bias || (((uintptr_t)p.bias) & 15) == 0) && (!p.bias_n || (((uintptr_t)p.bias_n) & 15) == 0) &&
                             (!p.res || ((p.res_ld & 3) == 0 && (((uintptr_t)p.res) & 15) == 0))),
              "conv16x3: an hl32 output needs the vector epilogue (Cout %% 4, whole 128-byte row blocks, aligned operands)");
  Conv32 q = p;
  q.zero = reinterpret_cast<const float*>(zero_page());
  EMP_REQUIRE(q.zero != nullptr, "conv16x3: no zero page");
  if (q.wpair) {
    EMP_REQUIRE(((uintptr_t)q.wpair % 16) == 0, "conv16x3: misaligned weight pairs");
    return launch_pair<true>(q, s);
  }
  return launch_pair<false>(q, s);
}

int conv16x3_cout_tiles(int Cout) { return Cout > 64 ? (Cout + 127) / 128 : 1; }

int launch_head_finish_f32(const float* part, int tiles, int N, int P, int C, const float* b, float* out, hipStream_t s) {
  EMP_REQUIRE(part && out && tiles >= 1 && C >= 1 && C <= 4, "head_finish: bad arguments");
  const int64_t MP = (int64_t)N * P, total = MP * C;
  const int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(head_finish32_kernel, dim3((unsigned)(g > 65535 ? 65535 : g)), dim3(256), 0, s, part, tiles, MP, P, C, b, out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// Weight IMAGE of the split-role kernel's WIMG variant: [cout tile of 128][step of 32 k][hi tile | lo tile], a tile = 128 rows
// of 64 B with 16-byte chunk c of row r at chunk c ^ swz(r) -- the bytes of the kernel's [Wh | Wl] LDS slot; rows past Cout
// are zeros.  w: [Cout][K] fp32, K % 32 == 0.
__global__ void __launch_bounds__(256) x3_weight_image_kernel(const float* __restrict__ w, half_t* __restrict__ out, int Cout, int K) {
  const int nsteps = K / X_BK, ntiles = (Cout + 127) / 128;
  const int64_t total = (int64_t)ntiles * nsteps * 128 * 4;      // (row, logical chunk) pairs
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i & 3), r = (int)((i >> 2) & 127);
    const int64_t ts = i >> 9;
    const int st = (int)(ts % nsteps), nt = (int)(ts / nsteps);
    const int co = nt * 128 + r;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = co < Cout ? w[(size_t)co * K + st * X_BK + c * 8 + e] : 0.f;
      const half_t a = (half_t)x;
      h[e] = a;
      l[e] = (half_t)(x - (float)a);
    }
    half_t* base = out + (size_t)ts * 2 * XS_TILE + r * X_LD + ((c ^ swz(r)) << 3);
    *reinterpret_cast<f16x8*>(base) = h;
    *reinterpret_cast<f16x8*>(base + XS_TILE) = l;
  }
}

// fp32 weights -> one uint32 per weight: fp16(x) | fp16(x - fp16(x)) << 16 (emp_pdl_finalize in the fp16x3 mode)
int launch_split_pairs(const float* w, uint32_t* out, int64_t n, hipStream_t s) {
  if (n <= 0) return EMP_OK;
  int64_t g = (n + 255) / 256;
  hipLaunchKernelGGL(split_pairs_kernel, dim3((unsigned)(g > 65535 ? 65535 : g)), dim3(256), 0, s, w, out, n);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// halfs of conv16x3's weight image for a [Cout][K] convolution, or 0 where the split-role kernel cannot take it
int64_t x3_weight_image_halfs(int Cout, int K, int Cin) {
  if (Cout <= 64 || K % X_BK != 0 || Cin % X_BK != 0) return 0;
  return (int64_t)((Cout + 127) / 128) * (K / X_BK) * 2 * XS_TILE;
}
int launch_x3_weight_image(const float* w, half_t* out, int Cout, int K, hipStream_t s) {
  EMP_REQUIRE(w && out && Cout > 0 && K > 0 && K % X_BK == 0, "x3_weight_image: bad shape");
  const int64_t total = (int64_t)((Cout + 127) / 128) * (K / X_BK) * 512;
  hipLaunchKernelGGL(x3_weight_image_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s, w, out, Cout, K);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
