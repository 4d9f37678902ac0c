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
// (prefetched two steps ahead) -> split -> LDS (two buffers of fp16 hi / lo tiles, one barrier per step; rows padded to 40
// halfs: conflict-free ds_read_b128 fragments).  Activations are split in the kernel (5 vector ops per element); weights arrive either as fp32 (the C ABI's
// emp_conv2d_nhwc_f16x3) or already split -- `Conv32::wpair`, one uint32 per weight = hi | lo << 16, made once at
// emp_pdl_finalize by split_pairs -- and then only change lanes (v_perm).
// Values must fit fp16's range (|x| <= 65504), as every map of the fp16 engine does.
#include "common.h"

namespace emp {
namespace {

constexpr int X_BK = 32, X_LD = 40;

template <int ACT>
__device__ __forceinline__ float x_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

// 16 fp32 values -> 16 hi halfs + 16 lo halfs, written as two 16-byte LDS stores each
__device__ __forceinline__ void split_store(const float4 (&r)[4], half_t* hi, half_t* lo) {
  f16x8 h[2], l[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float v[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half_t a = (half_t)v[e];
      h[q >> 1][(q & 1) * 4 + e] = a;
      l[q >> 1][(q & 1) * 4 + e] = (half_t)(v[e] - (float)a);
    }
  }
  *reinterpret_cast<f16x8*>(hi) = h[0];
  *reinterpret_cast<f16x8*>(hi + 8) = h[1];
  *reinterpret_cast<f16x8*>(lo) = l[0];
  *reinterpret_cast<f16x8*>(lo + 8) = l[1];
}

// 16 pre-split weights (uint32 = hi | lo << 16, carried in float4 registers) -> the same two pairs of LDS stores
__device__ __forceinline__ void pair_store(const float4 (&r)[4], half_t* hi, half_t* lo) {
  uint4 h[2], l[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t p0 = __float_as_uint(r[q].x), p1 = __float_as_uint(r[q].y), p2 = __float_as_uint(r[q].z), p3 = __float_as_uint(r[q].w);
    const uint32_t h01 = __builtin_amdgcn_perm(p1, p0, 0x05040100u), h23 = __builtin_amdgcn_perm(p3, p2, 0x05040100u);
    const uint32_t l01 = __builtin_amdgcn_perm(p1, p0, 0x07060302u), l23 = __builtin_amdgcn_perm(p3, p2, 0x07060302u);
    if (q & 1) { h[q >> 1].z = h01; h[q >> 1].w = h23; l[q >> 1].z = l01; l[q >> 1].w = l23; }
    else { h[q >> 1].x = h01; h[q >> 1].y = h23; l[q >> 1].x = l01; l[q >> 1].y = l23; }
  }
  *reinterpret_cast<uint4*>(hi) = h[0];
  *reinterpret_cast<uint4*>(hi + 8) = h[1];
  *reinterpret_cast<uint4*>(lo) = l[0];
  *reinterpret_cast<uint4*>(lo + 8) = l[1];
}

template <int ACT, int BM, int BN, bool WPAIR, bool DB>
__global__ void __launch_bounds__(256, DB ? 2 : 3) conv16x3_kernel(const Conv32 p) {
  constexpr int WC = BN / 64;          // waves along the couts (64 couts each)
  constexpr int WP = 4 / WC;           // waves along the pixels
  constexpr int NJ = BM / WP / 16;     // 16-pixel fragments per wave
  constexpr int XC = BM / 128;         // 16-channel chunks of the pixel tile per thread and step (BM rows x 2 chunks / 256)
  // two LDS buffers: step k computes from buffer k % 2 while the operands of step k + 1 (loaded during step k - 1) are split
  // into the other one and the loads of step k + 2 are in flight -- one barrier per step, two steps of latency cover
  // DB = false (short K: the launch is its prologue and epilogue): one buffer, two barriers per step, three workgroups per CU
  constexpr int NB = DB ? 2 : 1;
  __shared__ __attribute__((aligned(16))) half_t Xh[NB][BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Xl[NB][BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wh[NB][BN * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wl[NB][BN * X_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // grouped convolution (RegNet's 3x3): blockIdx.z = group, as conv32_kernel (ref32.hip)
  const int g = blockIdx.z, gco = g * p.Cout;
  const float* gin = p.in + (size_t)g * p.cin_g;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.N * HoWo;
  const int K = p.KH * p.KW * p.Cin;
  // staging roles.  pixels: BM == 256: thread -> row tid, both 16-channel chunks; BM == 128: row tid / 2, chunk tid % 2.
  // weights (BN rows): row tid / 2, chunk tid % 2
  const int xrow = XC == 2 ? tid : tid >> 1, xch0 = XC == 2 ? 0 : (tid & 1);
  const int am = m0 + xrow;
  int an = 0, aoy = 0, aox = 0;
  const bool xrow_ok = am < M;
  if (xrow_ok) {
    an = am / HoWo;
    const int r = am - an * HoWo;
    aoy = r / p.Wo;
    aox = r - aoy * p.Wo;
  }
  const int wr = tid >> 1, wch = tid & 1;
  const bool wrole = wr < BN;
  const bool wok = wrole && n0 + wr < p.Cout;
  const float* wrow = WPAIR ? reinterpret_cast<const float*>(p.wpair) : p.w;
  wrow = wok ? wrow + (size_t)(gco + n0 + wr) * K : nullptr;
  const int wc = (wave % WC) * 64, wp = (wave / WC) * (NJ * 16);
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 rx[XC][4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rx[c][q] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int kk = k0 + (xch0 + c) * 16;
      if (xrow_ok && kk < K) {
        const int tap = kk / p.Cin, c0 = kk - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = aoy * p.stride - p.pad + ky * p.dil, ix = aox * p.stride - p.pad + kx * p.dil;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const float4* src = reinterpret_cast<const float4*>(gin + (((size_t)an * p.H + iy) * p.W + ix) * p.in_ld + c0);
#pragma unroll
          for (int q = 0; q < 4; ++q) rx[c][q] = src[q];
        }
      }
    }
    const int kw = k0 + wch * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) rw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wrow && kw < K) {
      const float4* src = reinterpret_cast<const float4*>(wrow + kw);
#pragma unroll
      for (int q = 0; q < 4; ++q) rw[q] = src[q];
    }
  };
  auto stage = [&](int b) {
#pragma unroll
    for (int c = 0; c < XC; ++c)
      split_store(rx[c], Xh[b] + xrow * X_LD + (xch0 + c) * 16, Xl[b] + xrow * X_LD + (xch0 + c) * 16);
    if (wrole) {
      if (WPAIR) pair_store(rw, Wh[b] + wr * X_LD + wch * 16, Wl[b] + wr * X_LD + wch * 16);
      else split_store(rw, Wh[b] + wr * X_LD + wch * 16, Wl[b] + wr * X_LD + wch * 16);
    }
  };
  const int fr = lane & 15, fk = (lane >> 4) * 8;
  gload(0);
  if (DB) {
    stage(0);
    if (X_BK < K) gload(X_BK);
  }
  int b = 0;
  for (int k0 = 0; k0 < K; k0 += X_BK, b ^= (DB ? 1 : 0)) {
    if (DB) {
      __syncthreads();      // buffer b is complete; nobody reads buffer b ^ 1 (step k - 1) any more
      if (k0 + X_BK < K) {
        stage(b ^ 1);                                   // the registers hold step k + 1
        if (k0 + 2 * X_BK < K) gload(k0 + 2 * X_BK);    // in flight behind this step's and the next step's MFMAs
      }
    } else {
      __syncthreads();      // the previous step's fragment reads are done
      stage(0);
      __syncthreads();
      if (k0 + X_BK < K) gload(k0 + X_BK);
    }
    // the wave's weight fragments stay in registers for the step (32 VGPRs); the pixel fragments stream through
    f16x8 wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const f16x8*>(Wh[b] + (wc + i * 16 + fr) * X_LD + fk);
      wl[i] = *reinterpret_cast<const f16x8*>(Wl[b] + (wc + i * 16 + fr) * X_LD + fk);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f16x8 xh = *reinterpret_cast<const f16x8*>(Xh[b] + (wp + j * 16 + fr) * X_LD + fk);
      const f16x8 xl = *reinterpret_cast<const f16x8*>(Xl[b] + (wp + j * 16 + fr) * X_LD + fk);
      // consecutive MFMAs write different accumulators (a dependent one would wait for its predecessor's passes)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], xh, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xl, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], xh, acc[i][j], 0, 0, 0);
    }
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
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.out_ld + co) =
            make_float4(x_act<ACT>(t.x), x_act<ACT>(t.y), x_act<ACT>(t.z), x_act<ACT>(t.w));
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

template <int BM, int BN, bool WPAIR, bool DB>
int launch_tile(const Conv32& p, dim3 grid, hipStream_t s) {
  if (p.act == 1) hipLaunchKernelGGL((conv16x3_kernel<1, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  else if (p.act == 2) hipLaunchKernelGGL((conv16x3_kernel<2, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv16x3_kernel<0, BM, BN, WPAIR, DB>), grid, dim3(256), 0, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <bool WPAIR>
int launch_pair(const Conv32& p, hipStream_t s) {
  const int G = p.groups > 1 ? p.groups : 1;
  const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
  const int bn = p.Cout > 64 ? 128 : 64;
  const unsigned nt = (unsigned)((p.Cout + bn - 1) / bn);
  // (a 256-pixel tile -- half the weight tile's trips through LDS per product -- measured 1.4-2.4x SLOWER on every shape of
  // the network at batch 8: one workgroup per CU leaves nothing to run behind a barrier; profiles/r05_conv16x3.txt)
  const dim3 grid((unsigned)((M + 127) / 128), nt, (unsigned)G);
  const int kdb = [] { const char* e = getenv("EMP_X3_KDB"); return e ? atoi(e) : 1024; }();     // K from which the two-buffer pipeline runs (measured: profiles/r05_conv16x3.txt)
  const bool db = p.KH * p.KW * p.Cin >= kdb;
  if (db) return bn == 128 ? launch_tile<128, 128, WPAIR, true>(p, grid, s) : launch_tile<128, 64, WPAIR, true>(p, grid, s);
  return bn == 128 ? launch_tile<128, 128, WPAIR, false>(p, grid, s) : launch_tile<128, 64, WPAIR, false>(p, grid, s);
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
  if (p.wpair) {
    EMP_REQUIRE(((uintptr_t)p.wpair % 16) == 0, "conv16x3: misaligned weight pairs");
    return launch_pair<true>(p, s);
  }
  return launch_pair<false>(p, s);
}

// fp32 weights -> one uint32 per weight: fp16(x) | fp16(x - fp16(x)) << 16 (emp_pdl_finalize in the fp16x3 mode)
int launch_split_pairs(const float* w, uint32_t* out, int64_t n, hipStream_t s) {
  if (n <= 0) return EMP_OK;
  int64_t g = (n + 255) / 256;
  hipLaunchKernelGGL(split_pairs_kernel, dim3((unsigned)(g > 65535 ? 65535 : g)), dim3(256), 0, s, w, out, n);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
