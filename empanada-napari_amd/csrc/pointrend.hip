// PointRend subdivision inference for gfx950 (reference:
// empanada/models/point_rend.py:241-269, eval branch):
//   per step: x2 bilinear upsample (align_corners=False) of the logits,
//   uncertainty (-|logit| or top2 difference, :11-31), the num_points most
//   uncertain grid cells (torch.topk, :123-137), bilinear point sampling of the
//   coarse logits and of the decoder features (grid_sample, :33-60), a 3-layer
//   point MLP (runs on the implicit-GEMM kernel as a 1x1 conv over the point
//   list) and a scatter of the refined logits back into the grid.
// Top-k is an exact 4-pass radix select on the fp32 bit pattern of a
// non-negative key (smaller key = more uncertain) followed by an ORDERED
// compaction, so the selected set is deterministic: all keys below the k-th
// value, plus the lowest-index cells among those equal to it.
#include "common.h"

namespace emp {
namespace {

constexpr int CHUNK = 2048;  // elements per block in the select / compact passes

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
// exclusive scan over the 256 threads of a block; *total gets the block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sh /*[5]*/, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v, lane);
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < w) base += sh[i];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  return base + inc - v;
}

__global__ void __launch_bounds__(256) upsample2x_keys_kernel(const float* __restrict__ in, int N, int C, int h, int w,
                                                              float* __restrict__ out, uint32_t* __restrict__ keys,
                                                              int64_t total) {
  const int H = 2 * h, W = 2 * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int ox = (int)(i % W);
    int64_t p = i / W;
    int oy = (int)(p % H);
    int n = (int)(p / H);
    float fy = 0.5f * ((float)oy + 0.5f) - 0.5f;
    float fx = 0.5f * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    float ly = fy - (float)y0, lx = fx - (float)x0;
    float hy = 1.f - ly, hx = 1.f - lx;
    float top1 = -INFINITY, top2 = -INFINITY, v0 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* b = in + ((size_t)n * C + c) * h * w;
      float v = hy * (hx * b[y0 * w + x0] + lx * b[y0 * w + x1]) + ly * (hx * b[y1 * w + x0] + lx * b[y1 * w + x1]);
      out[(((size_t)n * C + c) * H + oy) * W + ox] = v;
      if (c == 0) v0 = v;
      if (v > top1) { top2 = top1; top1 = v; } else if (v > top2) { top2 = v; }
    }
    float key = (C == 1) ? fabsf(v0) : (top1 - top2);
    keys[i] = __float_as_uint(key);
  }
}

// state[n] = {prefix, k_remaining, unused, unused}
__global__ void init_state_kernel(uint32_t* state, uint32_t* hist, int N, uint32_t k) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) { state[4 * i] = 0; state[4 * i + 1] = k; state[4 * i + 2] = 0; state[4 * i + 3] = 0; }
  if (i < N * 256) hist[i] = 0;
}

__global__ void __launch_bounds__(256) radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t plane,
                                                         const uint32_t* __restrict__ state,
                                                         uint32_t* __restrict__ hist, int shift, uint32_t himask) {
  __shared__ uint32_t lh[256];
  const int n = blockIdx.y;
  lh[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t prefix = state[4 * n];
  const uint32_t* kp = keys + (size_t)n * plane;
  const int64_t base = (int64_t)blockIdx.x * CHUNK;
#pragma unroll
  for (int j = 0; j < CHUNK / 256; ++j) {
    int64_t i = base + j * 256 + threadIdx.x;
    if (i < plane) {
      uint32_t k = kp[i];
      if ((k & himask) == prefix) atomicAdd(&lh[(k >> shift) & 255u], 1u);
    }
  }
  __syncthreads();
  uint32_t c = lh[threadIdx.x];
  if (c) atomicAdd(&hist[n * 256 + threadIdx.x], c);
}

__global__ void __launch_bounds__(256) radix_select_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ state,
                                                           int shift) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.x;
  uint32_t c = hist[n * 256 + threadIdx.x];
  hist[n * 256 + threadIdx.x] = 0;
  uint32_t total;
  uint32_t ex = block_excl_scan(c, sh, &total);
  const uint32_t krem = state[4 * n + 1];
  const uint32_t prefix = state[4 * n];
  __syncthreads();
  if (c > 0 && ex < krem && krem <= ex + c) {
    state[4 * n] = prefix | ((uint32_t)threadIdx.x << shift);
    state[4 * n + 1] = krem - ex;
  }
}

__device__ __forceinline__ uint32_t count_flags(const uint32_t* kp, int64_t plane, int64_t e0, uint32_t T,
                                                uint32_t* flags_less, uint32_t* flags_eq) {
  // thread handles 8 consecutive elements starting at e0; returns less | eq<<16
  uint32_t fl = 0, fe = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int64_t i = e0 + j;
    if (i < plane) {
      uint32_t k = kp[i];
      fl |= (uint32_t)(k < T) << j;
      fe |= (uint32_t)(k == T) << j;
    }
  }
  *flags_less = fl;
  *flags_eq = fe;
  return (uint32_t)__popc(fl) | ((uint32_t)__popc(fe) << 16);
}

__global__ void __launch_bounds__(256) compact_count_kernel(const uint32_t* __restrict__ keys, int64_t plane,
                                                            const uint32_t* __restrict__ state,
                                                            uint32_t* __restrict__ blockcnt, int nb) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  const uint32_t T = state[4 * n];
  uint32_t fl, fe;
  uint32_t c = count_flags(keys + (size_t)n * plane, plane, (int64_t)blockIdx.x * CHUNK + threadIdx.x * 8, T, &fl, &fe);
  uint32_t total;
  block_excl_scan(c, sh, &total);
  if (threadIdx.x == 0) blockcnt[(size_t)n * nb + blockIdx.x] = total;
}

// exclusive scan of the per-block (less | eq<<16) counts -> 2 x uint32 offsets
__global__ void __launch_bounds__(256) compact_scan_kernel(const uint32_t* __restrict__ blockcnt,
                                                           uint32_t* __restrict__ blockoff, int nb) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.x;
  uint32_t carry_l = 0, carry_e = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    int b = b0 + threadIdx.x;
    uint32_t c = b < nb ? blockcnt[(size_t)n * nb + b] : 0;
    uint32_t tl, te;
    uint32_t el = block_excl_scan(c & 0xffffu, sh, &tl);
    uint32_t ee = block_excl_scan(c >> 16, sh, &te);
    if (b < nb) {
      blockoff[((size_t)n * nb + b) * 2] = carry_l + el;
      blockoff[((size_t)n * nb + b) * 2 + 1] = carry_e + ee;
    }
    carry_l += tl;
    carry_e += te;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) compact_write_kernel(const uint32_t* __restrict__ keys, int64_t plane,
                                                            const uint32_t* __restrict__ state,
                                                            const uint32_t* __restrict__ blockoff, int nb, int k,
                                                            int32_t* __restrict__ idx_out) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  const uint32_t T = state[4 * n];
  const uint32_t krem = state[4 * n + 1];
  const uint32_t nless = (uint32_t)k - krem;
  const int64_t e0 = (int64_t)blockIdx.x * CHUNK + threadIdx.x * 8;
  uint32_t fl, fe;
  uint32_t c = count_flags(keys + (size_t)n * plane, plane, e0, T, &fl, &fe);
  uint32_t tl, te;
  uint32_t pl = block_excl_scan(c & 0xffffu, sh, &tl) + blockoff[((size_t)n * nb + blockIdx.x) * 2];
  uint32_t pe = block_excl_scan(c >> 16, sh, &te) + blockoff[((size_t)n * nb + blockIdx.x) * 2 + 1];
  int32_t* o = idx_out + (size_t)n * k;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (fl & (1u << j)) { o[pl++] = (int32_t)(e0 + j); }
    if (fe & (1u << j)) {
      if (pe < krem) o[nless + pe] = (int32_t)(e0 + j);
      ++pe;
    }
  }
}

__global__ void iota_kernel(int32_t* idx, int N, int k) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N * k) idx[i] = i % k;
}

// ---------------------------------------------------------------------------
// point sampling: half a wave per point, lane = 8 feature channels
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) point_features_kernel(const half_t* __restrict__ feat, int N, int fh, int fw,
                                                             int C, int feat_ld, const float* __restrict__ coarse,
                                                             int ncls, const int32_t* __restrict__ idx, int P, int H2,
                                                             int W2, half_t* __restrict__ x0, half_t* __restrict__ x1,
                                                             int ld) {
  const int hl = threadIdx.x & 31;
  const int64_t pt = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  if (pt >= (int64_t)N * P) return;
  const int n = (int)(pt / P);
  const int id = idx[pt];
  const int iy = id / W2, ix = id - iy * W2;
  // point_rend.py:131-135 (fp32): coord = 0.5*step + step*index, step = 1/size
  const float w_step = 1.0f / (float)W2, h_step = 1.0f / (float)H2;
  const float cx = 0.5f * w_step + w_step * (float)ix;
  const float cy = 0.5f * h_step + h_step * (float)iy;
  // point_sample: grid = 2*coord - 1; grid_sample(align_corners=False): ((g+1)*size - 1)/2
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
  const int CG = C >> 3;
  half_t* r0 = x0 + (size_t)pt * ld;
  half_t* r1 = x1 + (size_t)pt * ld;
  const half_t* fb = feat + (size_t)n * fh * fw * feat_ld;
  for (int cg = hl; cg < CG; cg += 32) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto accum = [&](bool ok, int yy, int xx, float wt) {
      if (ok) {
        f16x8 v = *reinterpret_cast<const f16x8*>(fb + ((size_t)yy * fw + xx) * feat_ld + cg * 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = fmaf((float)v[c], wt, a[c]);
      }
    };
    accum(ok00, ya, xa, w00);
    accum(ok01, ya, xb, w01);
    accum(ok10, yb, xa, w10);
    accum(ok11, yb, xb, w11);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (half_t)a[c];
    *reinterpret_cast<f16x8*>(r0 + cg * 8) = o;
  }
  // tail chunks [C, ld): coarse logits in the first ncls slots, zeros elsewhere
  const int tail = (ld - C) >> 3;
  if (hl < tail) {
    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hl == 0) {
      for (int c = 0; c < ncls && c < 8; ++c) {
        const float* cb = coarse + ((size_t)n * ncls + c) * fh * fw;
        float v = 0.f;
        if (ok00) v = fmaf(cb[ya * fw + xa], w00, v);
        if (ok01) v = fmaf(cb[ya * fw + xb], w01, v);
        if (ok10) v = fmaf(cb[yb * fw + xa], w10, v);
        if (ok11) v = fmaf(cb[yb * fw + xb], w11, v);
        o[c] = (half_t)v;
      }
    }
    *reinterpret_cast<f16x8*>(r0 + C + hl * 8) = o;
    *reinterpret_cast<f16x8*>(r1 + C + hl * 8) = o;
  }
}

inline int grid_for(int64_t total, int per_block = 256, int cap = 256 * 16) {
  int64_t g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

int launch_upsample2x_keys(const float* in, int N, int C, int h, int w, float* out, uint32_t* keys, hipStream_t s) {
  int64_t total = (int64_t)N * 4 * h * w;
  hipLaunchKernelGGL(upsample2x_keys_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, N, C, h, w, out, keys, total);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

size_t topk_work_bytes(int N, int64_t plane) {
  int nb = (int)cdiv64(plane, CHUNK);
  return (size_t)N * (256 + 4 + (size_t)nb * 3) * sizeof(uint32_t) + 256;
}

// keys: (N, plane) non-negative-float bit patterns; selects the k smallest per image.
int launch_topk_smallest(const uint32_t* keys, int N, int64_t plane, int k, void* work, size_t work_bytes,
                         int32_t* idx_out, hipStream_t s) {
  EMP_REQUIRE(k > 0 && plane > 0, "topk: k=%d plane=%lld", k, (long long)plane);
  if ((int64_t)k >= plane) {
    EMP_REQUIRE((int64_t)k == plane, "topk: k > plane");
    hipLaunchKernelGGL(iota_kernel, dim3(cdiv(N * k, 256)), dim3(256), 0, s, idx_out, N, k);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  EMP_REQUIRE(work_bytes >= topk_work_bytes(N, plane), "topk: workspace too small");
  const int nb = (int)cdiv64(plane, CHUNK);
  uint32_t* hist = (uint32_t*)work;
  uint32_t* state = hist + (size_t)N * 256;
  uint32_t* blockcnt = state + (size_t)N * 4;
  uint32_t* blockoff = blockcnt + (size_t)N * nb;
  hipLaunchKernelGGL(init_state_kernel, dim3(cdiv(N * 256, 256)), dim3(256), 0, s, state, hist, N, (uint32_t)k);
  EMP_LAUNCH_CHECK();
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    hipLaunchKernelGGL(radix_hist_kernel, dim3(nb, N), dim3(256), 0, s, keys, plane, state, hist, shift, himask);
    EMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(radix_select_kernel, dim3(N), dim3(256), 0, s, hist, state, shift);
    EMP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(compact_count_kernel, dim3(nb, N), dim3(256), 0, s, keys, plane, state, blockcnt, nb);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(compact_scan_kernel, dim3(N), dim3(256), 0, s, blockcnt, blockoff, nb);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(compact_write_kernel, dim3(nb, N), dim3(256), 0, s, keys, plane, state, blockoff, nb, k, idx_out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_point_features(const half_t* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse, int ncls,
                          const int32_t* idx, int P, int H2, int W2, half_t* x0, half_t* x1, int ld, hipStream_t s) {
  EMP_REQUIRE(C % 8 == 0 && ld % 8 == 0 && ld >= C + 8 && ncls <= 8, "point_features: bad channel layout");
  EMP_REQUIRE((ld - C) / 8 <= 32, "point_features: tail too wide");
  int64_t pts = (int64_t)N * P;
  hipLaunchKernelGGL(point_features_kernel, dim3((unsigned)cdiv64(pts, 8)), dim3(256), 0, s, feat, N, fh, fw, C, feat_ld,
                     coarse, ncls, idx, P, H2, W2, x0, x1, ld);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
