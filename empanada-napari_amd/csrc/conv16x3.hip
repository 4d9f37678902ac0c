// Split-fp16 convolution of the `fp16x3` precision mode (round 5; emp_pdl_set_precision(net, 2) / EMP_PRECISION=fp16x3).
//
// The reference computes this path in fp32 (empanada/inference/engines.py:248-255).  The fp16 engine is within 1e-3 of it
// in rms only, and not for every weight draw (tests/test_gpu_parity_stats.py); the fp32 reference mode (ref32.hip) is
// within 1e-4 in the max norm but runs on the exact fp32 matrix pipe at 1/16 of the fp16 rate.  This kernel is the fp32
// mode's convolution on the FP16 matrix pipe: same operands (NHWC fp32 maps, fp32 weights [Cout][KH*KW][Cin16]), same
// epilogue, same launch contract (`Conv32`); every operand is split on its way into LDS,
//     x = hi + lo,   hi = fp16(x),   lo = fp16(x - hi)          (22 significant bits: 2^-22 relative)
// and every product is three MFMAs into one fp32 accumulator:  w_lo . x_hi  +  w_hi . x_lo  +  w_hi . x_hi  (the
// w_lo . x_lo term is 2^-22 of the product: dropped).  3 x 1/16 of the fp32 pipe's time for the same result to ~1e-6
// relative per term -- the heads stay within 1e-3 of the fp32 forward in the MAX norm (tests/test_gpu_fp16x3.py).
//
// Tile: 128 pixels x BN couts (BN = 128 or 64), K walked tap-major in steps of 32 channels (two 16-channel chunks, each
// inside one tap because Cin16 % 16 == 0), four waves; A = weights, B = pixels, so a lane's four accumulator values are
// four consecutive couts of one pixel (float4 stores).  Operands are staged global -> registers (fp32, prefetched one
// step ahead) -> split -> LDS (fp16 hi / lo tiles, rows padded to 40 halfs: conflict-free ds_read_b128 fragments).
// Values must fit fp16's range (|x| <= 65504), as every map of the fp16 engine does.
#include "common.h"

namespace emp {
namespace {

constexpr int X_BM = 128, X_BK = 32, X_LD = 40;

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

template <int ACT, int BN>
__global__ void __launch_bounds__(256, 2) conv16x3_kernel(const Conv32 p) {
  constexpr int WC = BN / 64;          // waves along the couts (64 couts each)
  constexpr int WP = 4 / WC;           // waves along the pixels
  constexpr int NJ = X_BM / WP / 16;   // 16-pixel fragments per wave
  __shared__ __attribute__((aligned(16))) half_t Xh[X_BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Xl[X_BM * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wh[BN * X_LD];
  __shared__ __attribute__((aligned(16))) half_t Wl[BN * X_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * X_BM, n0 = blockIdx.y * BN;
  // grouped convolution (RegNet's 3x3): blockIdx.z = group, as conv32_kernel (ref32.hip)
  const int g = blockIdx.z, gco = g * p.Cout;
  const float* gin = p.in + (size_t)g * p.cin_g;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.N * HoWo;
  const int K = p.KH * p.KW * p.Cin;
  // staging roles: thread -> (row, one 16-channel chunk of the step); the weight tile has BN rows
  const int srow = tid >> 1, shalf = tid & 1;
  const int am = m0 + srow;
  int an = 0, aoy = 0, aox = 0;
  const bool xrow_ok = am < M;
  if (xrow_ok) {
    an = am / HoWo;
    const int r = am - an * HoWo;
    aoy = r / p.Wo;
    aox = r - aoy * p.Wo;
  }
  const bool wrole = srow < BN;
  const float* wrow = (wrole && n0 + srow < p.Cout) ? p.w + (size_t)(gco + n0 + srow) * K : nullptr;
  const int wc = (wave % WC) * 64, wp = (wave / WC) * (NJ * 16);
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 rx[4], rw[4];
  auto gload = [&](int k0) {
    const int kk = k0 + shalf * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) rx[q] = rw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kk >= K) return;
    if (xrow_ok) {
      const int tap = kk / p.Cin, c0 = kk - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int iy = aoy * p.stride - p.pad + ky * p.dil, ix = aox * p.stride - p.pad + kx * p.dil;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        const float4* src = reinterpret_cast<const float4*>(gin + (((size_t)an * p.H + iy) * p.W + ix) * p.in_ld + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) rx[q] = src[q];
      }
    }
    if (wrow) {
      const float4* src = reinterpret_cast<const float4*>(wrow + kk);
#pragma unroll
      for (int q = 0; q < 4; ++q) rw[q] = src[q];
    }
  };
  gload(0);
  const int fr = lane & 15, fk = (lane >> 4) * 8;
  for (int k0 = 0; k0 < K; k0 += X_BK) {
    __syncthreads();      // the previous step's fragment reads are done
    split_store(rx, Xh + srow * X_LD + shalf * 16, Xl + srow * X_LD + shalf * 16);
    if (wrole) split_store(rw, Wh + srow * X_LD + shalf * 16, Wl + srow * X_LD + shalf * 16);
    __syncthreads();
    if (k0 + X_BK < K) gload(k0 + X_BK);      // in flight behind this step's MFMAs
    f16x8 xh[NJ], xl[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xh[j] = *reinterpret_cast<const f16x8*>(Xh + (wp + j * 16 + fr) * X_LD + fk);
      xl[j] = *reinterpret_cast<const f16x8*>(Xl + (wp + j * 16 + fr) * X_LD + fk);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f16x8 wh = *reinterpret_cast<const f16x8*>(Wh + (wc + i * 16 + fr) * X_LD + fk);
      const f16x8 wl = *reinterpret_cast<const f16x8*>(Wl + (wc + i * 16 + fr) * X_LD + fk);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // epilogue (conv32_kernel's): bias (+ per-image bias) (+ residual), activation, store (NHWC slice or k2s2 pixel shuffle).
  // acc[i][j][e] <-> cout n0 + wc + 16 i + 4 (lane / 16) + e, pixel m0 + wp + 16 j + lane % 16
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + wp + j * 16 + fr;
    if (m >= M) continue;
    const int n = m / HoWo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = n0 + wc + i * 16 + (lane >> 4) * 4;
      if (col >= p.Cout) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = gco + col + e;
        float t = acc[i][j][e];
        if (col + e < p.Cout) {
          t += p.bias ? p.bias[co] : 0.f;
          if (p.bias_n) t += p.bias_n[(size_t)n * p.Cout + co];
          if (p.res) t += p.res[(size_t)m * p.res_ld + co];
        }
        v[e] = x_act<ACT>(t);
      }
      const int co = gco + col;
      if (p.ps_cout == 0) {
        float* o = p.out + (size_t)m * p.out_ld + co;
        if (col + 3 < p.Cout && ((((uintptr_t)o) & 15) == 0)) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < p.Cout) o[e] = v[e];
        }
      } else {      // ConvTranspose2d(k=2, s=2) as four sub-pixel 1x1 convs: (n,y,x,q*C+c) -> (n, 2y+dy, 2x+dx, c)
        const int r = m - n * HoWo;
        const int y = r / p.Wo, x = r - y * p.Wo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col + e >= p.Cout) continue;
          const int q = (co + e) / p.ps_cout, c = (co + e) - q * p.ps_cout;
          p.out[(((size_t)n * (2 * p.Ho) + 2 * y + (q >> 1)) * (2 * p.Wo) + 2 * x + (q & 1)) * p.out_ld + c] = v[e];
        }
      }
    }
  }
}

template <int BN>
int launch_bn(const Conv32& p, dim3 grid, hipStream_t s) {
  if (p.act == 1) hipLaunchKernelGGL((conv16x3_kernel<1, BN>), grid, dim3(256), 0, s, p);
  else if (p.act == 2) hipLaunchKernelGGL((conv16x3_kernel<2, BN>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv16x3_kernel<0, BN>), grid, dim3(256), 0, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace

// the checks of launch_conv32 (ref32.hip) have run: same contract
int launch_conv16x3(const Conv32& p, hipStream_t s) {
  const int G = p.groups > 1 ? p.groups : 1;
  const int64_t M = (int64_t)p.N * p.Ho * p.Wo;
  const unsigned mt = (unsigned)((M + X_BM - 1) / X_BM);
  if (p.Cout > 64) return launch_bn<128>(p, dim3(mt, (unsigned)((p.Cout + 127) / 128), (unsigned)G), s);
  return launch_bn<64>(p, dim3(mt, 1u, (unsigned)G), s);
}

}  // namespace emp
