// Stem on the matrix pipe: 7x7 stride-2 pad-3 conv (1 -> 64 channels) + bias + ReLU + 3x3 stride-2 pad-1 max-pool
// in one launch (reference: encoders/resnet.py:164-168,219-222: conv1 + bn1 + relu + maxpool, BN folded), with the
// uint8/uint16 -> normalise -> zero-pad preprocessing fused in like layers.hip's VALU stem.
//
// The VALU stem is bound by its 49 x 64 fp32 FMAs per output pixel (0.70 ms per 32 tiles) and writes a 1 GiB
// half-resolution map that the max-pool (0.32 ms) reads straight back.  Here the conv is a GEMM
//   D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],   k = ky * 8 + kx  (7 x 7 taps padded to 8 x 8 = 64),
// and fp32 accuracy is kept by splitting both operands into fp16 hi + lo parts (x = hi + lo, |lo| <= 2^-11 |x|):
//   W.X ~= Wh.Xh + Wh.Xl + Wl.Xh        (three MFMAs, error ~2^-21 relative: below the fp16 rounding of the output).
// The X fragment of a lane is 8 consecutive input pixels of one patch row, i.e. four ds_read_b32 from the LDS patch.
//
// One workgroup = one 8 x 8 tile of POOLED outputs: the 17 x 17 conv outputs it needs (stored fp16 in LDS, -inf
// outside the conv map so that the pool's padding never wins) from a 40 x 40 input patch; the conv map itself is
// never written to HBM.
#include "common.h"

namespace emp {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef half_t f16x2 __attribute__((ext_vector_type(2)));

constexpr int PT = 8;                 // pooled tile side
constexpr int ST = 2 * PT + 1;        // conv ("stem") tile side: 17
constexpr int NSP = ST * ST;          // 289 conv pixels
constexpr int NMT = (NSP + 15) / 16;  // 19 MFMA pixel tiles
constexpr int IP = 40;                // input patch side (2*16 + 7 = 39 rows/cols used, padded to 40)

template <typename T>
__global__ void __launch_bounds__(256) stem_pool_kernel(const T* __restrict__ img, float sub, float mul, int N, int H,
                                                        int W, int vh, int vw, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, half_t* __restrict__ out,
                                                        int normalise) {
  __shared__ __attribute__((aligned(16))) half_t ph[IP][IP];          // patch, hi part
  __shared__ __attribute__((aligned(16))) half_t pl[IP][IP];          // patch, lo part
  __shared__ __attribute__((aligned(16))) half_t st[NMT * 16][64];    // conv tile, fp16 after bias + ReLU
  const int Hs = H >> 1, Ws = W >> 1;       // conv map
  const int Hp = Hs >> 1, Wp = Ws >> 1;     // pooled map
  const int tiles_x = (Wp + PT - 1) / PT, tiles_y = (Hp + PT - 1) / PT;
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  const int total_tiles = N * tiles_y * tiles_x;
  // weight fragments (MFMA A operand: row = cout, 8 consecutive k per lane), hi / lo -- once per workgroup, the
  // workgroups are persistent over tiles
  const int fr = l & 15, fq = l >> 4;
  f16x8 wh[4][2], wl[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ky = fq + 4 * ks;
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) {
        const float w = (ky < 7 && kx < 7) ? wgt[(ky * 7 + kx) * 64 + ct * 16 + fr] : 0.f;
        const half_t h = (half_t)w;
        wh[ct][ks][kx] = h;
        wl[ct][ks][kx] = (half_t)(w - (float)h);
      }
    }
  float bv[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[ct][r] = bias[ct * 16 + fq * 4 + r];

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
  int b = tile;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; b /= tiles_y;
  const int n = b;
  const int py0 = ty * PT, px0 = tx * PT;
  const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;        // conv tile origin (may be -1)
  const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;        // input patch origin
  const T* src = img + (size_t)n * vh * vw;
  // branch-free (clamped address, masked value): the 7 loads of a thread are in flight together instead of one per
  // loop-carried branch
#pragma unroll
  for (int k = 0; k < (IP * IP + 255) / 256; ++k) {
    const int i = tid + 256 * k;
    const int r = i / IP, c = i - r * IP;
    const int iy = iy0 + r, ix = ix0 + c;
    const bool ok = i < IP * IP && iy >= 0 && iy < vh && ix >= 0 && ix < vw;
    const int cy = min(max(iy, 0), vh - 1), cx = min(max(ix, 0), vw - 1);
    float v = (float)src[(size_t)cy * vw + cx];
    if (normalise) { v -= sub; v *= mul; }
    v = ok ? v : 0.f;
    const half_t h = (half_t)v;
    if (i < IP * IP) {
      ph[r][c] = h;
      pl[r][c] = (half_t)(v - (float)h);
    }
  }
  __syncthreads();

  for (int mt = wave; mt < NMT; mt += 4) {
    const int q = mt * 16 + fr;                       // this lane's conv pixel (B operand column)
    const int qq = q < NSP ? q : NSP - 1;
    const int sy = qq / ST, sx = qq - sy * ST;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r = 2 * sy + fq + 4 * ks;             // patch row of tap ky = fq + 4 ks  (<= 39)
      const uint32_t* rh = reinterpret_cast<const uint32_t*>(&ph[r][2 * sx]);
      const uint32_t* rl = reinterpret_cast<const uint32_t*>(&pl[r][2 * sx]);
      union { uint32_t u[4]; f16x8 v; } xh, xl;
#pragma unroll
      for (int i = 0; i < 4; ++i) { xh.u[i] = rh[i]; xl.u[i] = rl[i]; }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ct][ks], xh.v, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][ks], xl.v, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][ks], xh.v, acc[ct], 0, 0, 0);
      }
    }
    // conv pixel outside the conv map -> -inf (max-pool padding); lane owns couts ct*16 + fq*4 + [0,4) of pixel q
    const int gy = sy0 + sy, gx = sx0 + sx;
    const bool inside = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
    const uint32_t keep = inside ? 0xFFFFFFFFu : 0u;
    // packed epilogue: fp32 bias add on pairs, one conversion per pair, ReLU on the packed fp16 pair (rounding is monotone
    // and keeps 0: the same values as ReLU before the rounding), border select on the packed word
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      union { f16x2 h[2]; uint32_t u[2]; f16x4 v; } o;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x2 s2 = f32x2{acc[ct][2 * r], acc[ct][2 * r + 1]} + f32x2{bv[ct][2 * r], bv[ct][2 * r + 1]};
        const f16x2 h2 = __builtin_convertvector(s2, f16x2);
        o.h[r] = __builtin_elementwise_max(h2, f16x2{(half_t)0.f, (half_t)0.f});
        o.u[r] = (o.u[r] & keep) | (0xFC00FC00u & ~keep);      // outside the conv map: (-inf, -inf); a bit select, no branch
      }
      // 16-byte chunk c of conv pixel q lives at chunk c ^ (q & 7): the rows are 128 B = all 32 banks apart, so the 16 lanes of a
      // ds_write_b64 group (16 consecutive pixels, same couts) hit ONE pair of banks without it (round 5 census: 81 % of this
      // kernel's LDS cycles were bank conflicts)
      *reinterpret_cast<f16x4*>(&st[q][(((ct * 2 + (fq >> 1)) ^ (q & 7)) << 3) + (fq & 1) * 4]) = o.v;
    }
  }
  __syncthreads();

  // 3x3 stride-2 max-pool of the 17x17 tile -> 8x8 pooled pixels x 8 channel groups of 8
  for (int i = tid; i < PT * PT * 8; i += 256) {
    const int cg = i & 7, pp = i >> 3;
    const int py = pp / PT, px = pp - py * PT;
    const int oy = py0 + py, ox = px0 + px;
    if (oy >= Hp || ox >= Wp) continue;
    float m[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int sq = (2 * py + dy) * ST + 2 * px + dx;
        const f16x8 v = *reinterpret_cast<const f16x8*>(&st[sq][(cg ^ (sq & 7)) * 8]);
#pragma unroll
        for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], (float)v[c]);
      }
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (half_t)m[c];
    *reinterpret_cast<f16x8*>(out + (((size_t)n * Hp + oy) * Wp + ox) * 64 + cg * 8) = o;
  }
  __syncthreads();      // the LDS tiles are reused by the next tile
  }
}


// ---- the same launch for the fp32 graph (fp32 reference mode's schedule in the fp16x3 mode; round 6) ---------------------------------
// conv1 + bn1 + relu + maxpool with an fp32 conv tile in LDS and an fp32 NHWC output: the products are the three-MFMA split above
// (fp32 result to ~2^-21 relative, what every convolution of the fp16x3 mode delivers), bias / ReLU / max in fp32.  Replaces the
// VALU stem7x7 launch + the max-pool launch of run32 (0.56 + 0.30 ms at batch 16 -> one launch).  Conv tile: 304 rows of 64 floats
// (256 B); 16-byte chunk c (0..15) of pixel q at chunk c ^ (q & 15): the 16 lanes of a store group (16 pixels, same couts) and the
// nine reads of a pooling thread spread over all banks.
template <typename T>
__global__ void __launch_bounds__(256) stem_pool32_kernel(const T* __restrict__ img, float sub, float mul, int N, int H,
                                                          int W, int vh, int vw, const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int normalise) {
  extern __shared__ __attribute__((aligned(16))) char sp_lds[];
  half_t (*ph)[IP] = reinterpret_cast<half_t (*)[IP]>(sp_lds);                                  // patch, hi part
  half_t (*pl)[IP] = reinterpret_cast<half_t (*)[IP]>(sp_lds + IP * IP * 2);                    // patch, lo part
  float (*st)[64] = reinterpret_cast<float (*)[64]>(sp_lds + 2 * IP * IP * 2);                  // conv tile, fp32 after bias + ReLU
  const int Hs = H >> 1, Ws = W >> 1;
  const int Hp = Hs >> 1, Wp = Ws >> 1;
  const int tiles_x = (Wp + PT - 1) / PT, tiles_y = (Hp + PT - 1) / PT;
  const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
  const int total_tiles = N * tiles_y * tiles_x;
  const int fr = l & 15, fq = l >> 4;
  f16x8 wh[4][2], wl[4][2];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ky = fq + 4 * ks;
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) {
        const float w = (ky < 7 && kx < 7) ? wgt[(ky * 7 + kx) * 64 + ct * 16 + fr] : 0.f;
        const half_t h = (half_t)w;
        wh[ct][ks][kx] = h;
        wl[ct][ks][kx] = (half_t)(w - (float)h);
      }
    }
  float bv[4][4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[ct][r] = bias[ct * 16 + fq * 4 + r];

  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int n = b;
    const int py0 = ty * PT, px0 = tx * PT;
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;
    const T* src = img + (size_t)n * vh * vw;
#pragma unroll
    for (int k = 0; k < (IP * IP + 255) / 256; ++k) {
      const int i = tid + 256 * k;
      const int r = i / IP, c = i - r * IP;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = i < IP * IP && iy >= 0 && iy < vh && ix >= 0 && ix < vw;
      const int cy = min(max(iy, 0), vh - 1), cx = min(max(ix, 0), vw - 1);
      float v = (float)src[(size_t)cy * vw + cx];
      if (normalise) { v -= sub; v *= mul; }
      v = ok ? v : 0.f;
      const half_t h = (half_t)v;
      if (i < IP * IP) {
        ph[r][c] = h;
        pl[r][c] = (half_t)(v - (float)h);
      }
    }
    __syncthreads();
    for (int mt = wave; mt < NMT; mt += 4) {
      const int q = mt * 16 + fr;
      const int qq = q < NSP ? q : NSP - 1;
      const int sy = qq / ST, sx = qq - sy * ST;
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r = 2 * sy + fq + 4 * ks;
        const uint32_t* rh = reinterpret_cast<const uint32_t*>(&ph[r][2 * sx]);
        const uint32_t* rl = reinterpret_cast<const uint32_t*>(&pl[r][2 * sx]);
        union { uint32_t u[4]; f16x8 v; } xh, xl;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xh.u[i] = rh[i]; xl.u[i] = rl[i]; }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ct][ks], xh.v, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][ks], xh.v, acc[ct], 0, 0, 0);
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ct][ks], xl.v, acc[ct], 0, 0, 0);
        }
      }
      const int gy = sy0 + sy, gx = sx0 + sx;
      const bool inside = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = inside ? fmaxf(acc[ct][r] + bv[ct][r], 0.f) : -INFINITY;      // outside the conv map: the pool's padding
        *reinterpret_cast<f32x4*>(&st[q][(((ct * 4 + fq) ^ (q & 15)) << 2)]) = o;
      }
    }
    __syncthreads();
    // 3x3 stride-2 max-pool of the 17x17 tile -> 8x8 pooled pixels x 16 channel groups of 4
    for (int i = tid; i < PT * PT * 16; i += 256) {
      const int cg = i & 15, pp = i >> 4;
      const int py = pp / PT, px = pp - py * PT;
      const int oy = py0 + py, ox = px0 + px;
      if (oy >= Hp || ox >= Wp) continue;
      f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int sq = (2 * py + dy) * ST + 2 * px + dx;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&st[sq][((cg ^ (sq & 15)) << 2)]);
#pragma unroll
          for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], v[c]);
        }
      *reinterpret_cast<f32x4*>(out + (((size_t)n * Hp + oy) * Wp + ox) * 64 + cg * 4) = m;
    }
    __syncthreads();
  }
}

}  // namespace

// img (N,vh,vw) -> pooled (N,H/4,W/4,64) fp16; H, W = padded size (multiples of 4), pixels outside vh x vw are zero
int launch_stem_pool(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                     const float* w, const float* b, half_t* out, hipStream_t s) {
  EMP_REQUIRE(H % 4 == 0 && W % 4 == 0, "stem: H, W must be multiples of 4");
  const int Hp = H / 4, Wp = W / 4;
  const int64_t tiles = (int64_t)N * cdiv(Hp, PT) * cdiv(Wp, PT);
  EMP_REQUIRE(tiles < (1ll << 31), "stem: too many tiles");
  const int64_t grid = tiles < 256 * 12 ? tiles : 256 * 12;      // persistent: 3 workgroups per CU x 4 rounds of slack
  switch (dtype) {
    case EMP_IMG_F32:
      hipLaunchKernelGGL(stem_pool_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, (const float*)img, sub, mul, N, H, W,
                         vh, vw, w, b, out, 0);
      break;
    case EMP_IMG_U8:
      hipLaunchKernelGGL(stem_pool_kernel<uint8_t>, dim3((unsigned)grid), dim3(256), 0, s, (const uint8_t*)img, sub, mul, N,
                         H, W, vh, vw, w, b, out, 1);
      break;
    case EMP_IMG_U16:
      hipLaunchKernelGGL(stem_pool_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0, s, (const uint16_t*)img, sub, mul,
                         N, H, W, vh, vw, w, b, out, 1);
      break;
    default:
      EMP_REQUIRE(false, "stem: unknown image dtype %d", dtype);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// the fp32-graph form: img (N,vh,vw) -> pooled (N,H/4,W/4,64) fp32 (stem_pool32_kernel)
int launch_stem_pool_f32(const void* img, int dtype, float sub, float mul, int N, int H, int W, int vh, int vw,
                         const float* w, const float* b, float* out, hipStream_t s) {
  EMP_REQUIRE(H % 4 == 0 && W % 4 == 0, "stem: H, W must be multiples of 4");
  const int Hp = H / 4, Wp = W / 4;
  const int64_t tiles = (int64_t)N * cdiv(Hp, PT) * cdiv(Wp, PT);
  EMP_REQUIRE(tiles < (1ll << 31), "stem: too many tiles");
  constexpr int LDS = 2 * IP * IP * 2 + NMT * 16 * 64 * 4;      // 6400 + 77824
  const int64_t grid = tiles < 256 * 4 ? tiles : 256 * 4;       // persistent: one workgroup per CU (84 KiB of LDS) x 4 rounds of slack
  auto go = [&](auto kern, auto ptr, int norm) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, s, ptr, sub, mul, N, H, W, vh, vw, w, b, out, norm);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  };
  switch (dtype) {
    case EMP_IMG_F32: return go(&stem_pool32_kernel<float>, (const float*)img, 0);
    case EMP_IMG_U8: return go(&stem_pool32_kernel<uint8_t>, (const uint8_t*)img, 1);
    case EMP_IMG_U16: return go(&stem_pool32_kernel<uint16_t>, (const uint16_t*)img, 1);
    default: EMP_REQUIRE(false, "stem: unknown image dtype %d", dtype);
  }
  return EMP_OK;
}

}  // namespace emp
