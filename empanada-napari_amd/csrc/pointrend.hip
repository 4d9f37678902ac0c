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

// ---------------------------------------------------------------------------
// Fused point head (point_rend.py:181-188,241-269, eval): point sampling -> num_fc x (Conv1d k=1 + ReLU, coarse logits
// concatenated at every layer) -> predictor -> scatter, for a tile of 256 points per workgroup.  Replaces
// point_features_kernel + num_fc implicit-GEMM launches + head1x1_kernel: the (points x 320) rows never reach HBM.
//   LDS: X[256 points][LD channels] fp16 (all 160 KiB at LD = 320), rows of LD*2 bytes, 16-byte chunk c of each 128-byte segment stored at
//        c ^ (row & 7) (conflict-free ds_read_b128 fragment reads, as in conv_igemm.hip).
//   gather    : half a wave per point, lane = 8 channels; arithmetic of point_features_kernel (fmaf chain per tap).
//   layer     : wave w owns couts [MT*16*w, +MT*16) x all 256 points; K walked in ascending 32-channel steps from a zero
//               accumulator (the order of conv_igemm_kernel: results are bit-identical to the unfused launches); the
//               wave's weight fragments of the WHOLE layer (MT x LD/32 x 4 VGPRs) are loaded from L2 in one burst,
//               the next layer's during this layer's epilogue, so the K loop waits for LDS only [first version:
//               fragments fetched per K-step at two workgroups per CU -- 302 us per launch, 8x its MFMA time, every
//               K-step exposed an L2 round trip]; bias + ReLU + ONE fp16 rounding, written back over X[:, 0:C) after
//               a barrier.  Two passes of 128 points per layer (64 accumulator VGPRs beside 72 weight VGPRs).
//   Measured (32 x 8192 points, MI355X): 257 us per launch = gather 88 (0.54 GB of scattered 512-byte pixel rows: at the
//   memory system's rate) + layers ~115 (MFMA time 60) + predictor ~40, against 120 + 330 + 55 us for the launches it
//   replaces; bit-identical sem_logits (tests/test_gpu_model.py).
//   predictor : half a wave per point on the fp16 row, fp32 weights, the reduction order of head1x1_kernel.
template <int C>
struct PrCfg {
  static constexpr int MT = C / 128;          // 16-cout tiles per wave (8 waves cover C couts)
};

struct PrParams {
  const half_t* feat; int N, fh, fw, feat_ld;
  const float* coarse; int ncls;
  const int32_t* idx; int P, H2, W2;
  const half_t* w[4]; const float* b[4]; int num_fc;    // fc layers: [C][LD] fp16, bias fp32
  const float* pw; const float* pb;                     // predictor [ncls][LD] fp32
  float* out; int64_t plane;
  int64_t npts; int tiles;
};

template <int C, int LD>
__global__ void __launch_bounds__(512, 1) pr_mlp_kernel(const PrParams p) {
  constexpr int MT = PrCfg<C>::MT;
  constexpr int LDB = LD * 2;                 // bytes per row
  // K-steps: channels [C + 8, LD) of a row are zeros (ncls <= 8) and so are the weights' pad columns, so the steps beyond
  // C/32 + 1 add exact zeros: skipping them leaves every sum bit-identical
  constexpr int KSTEPS = C / 32 + 1;
  constexpr int KC = LD / 8;                  // 16-byte chunks per row
  constexpr int PT = 256, QT = PT / 16;       // points per tile, 16-point MFMA tiles
  extern __shared__ __attribute__((aligned(1024))) char X[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hl = lane & 31, hw = tid >> 5;    // half-wave index 0..15
  const int fr = lane & 15, fq = lane >> 4;
  auto xoff = [](int row, int chunk) { return row * LDB + (chunk >> 3) * 128 + (((chunk & 7) ^ (row & 7)) << 4); };

  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    const int64_t g0 = (int64_t)tile * PT;
    // ---------------- gather ----------------
    // 16 points per half-wave, four at a time: the sixteen 16-byte feature taps of a group (and lane 0's coarse-logit
    // taps) are in flight before the first is used.  [measured: this phase moves 0.54 GB of scattered 512-byte pixel
    // rows per launch in ~105 us = 5 TB/s -- it is at the memory system's rate, deeper batching (8 points) was slower]
    auto sample_geometry = [&](int row, int& n, int& xa, int& ya, float* wt, bool* ok) {
      const int64_t pt = g0 + row;
      const bool live = pt < p.npts;
      const int id = live ? p.idx[pt] : 0;
      n = live ? (int)(pt / p.P) : 0;
      const int iy = id / p.W2, ix = id - iy * p.W2;
      // point_rend.py:131-135 (fp32): coord = 0.5*step + step*index, step = 1/size; point_sample: grid = 2*coord - 1;
      // grid_sample(align_corners=False): ((g+1)*size - 1)/2
      const float w_step = 1.0f / (float)p.W2, h_step = 1.0f / (float)p.H2;
      const float cx = 0.5f * w_step + w_step * (float)ix;
      const float cy = 0.5f * h_step + h_step * (float)iy;
      const float gx = 2.0f * cx - 1.0f, gy = 2.0f * cy - 1.0f;
      const float sx = ((gx + 1.f) * (float)p.fw - 1.f) * 0.5f;
      const float sy = ((gy + 1.f) * (float)p.fh - 1.f) * 0.5f;
      const float fx0 = floorf(sx), fy0 = floorf(sy);
      xa = (int)fx0; ya = (int)fy0;
      const int xb = xa + 1, yb = ya + 1;
      const float lx = sx - fx0, ly = sy - fy0;
      wt[0] = (1.f - lx) * (1.f - ly); wt[1] = lx * (1.f - ly); wt[2] = (1.f - lx) * ly; wt[3] = lx * ly;
      ok[0] = live && xa >= 0 && xa < p.fw && ya >= 0 && ya < p.fh;
      ok[1] = live && xb >= 0 && xb < p.fw && ya >= 0 && ya < p.fh;
      ok[2] = live && xa >= 0 && xa < p.fw && yb >= 0 && yb < p.fh;
      ok[3] = live && xb >= 0 && xb < p.fw && yb >= 0 && yb < p.fh;
    };
    {
      constexpr int GB = 4;
#pragma unroll 1
      for (int j0 = 0; j0 < PT / 16; j0 += GB) {
        int nn[GB], xa[GB], ya[GB];
        float wt[GB][4];
        bool ok[GB][4];
#pragma unroll
        for (int u = 0; u < GB; ++u) sample_geometry(hw + 16 * (j0 + u), nn[u], xa[u], ya[u], wt[u], ok[u]);
        if (hl < C / 8) {
          f16x8 v[GB][4];
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            const half_t* fb = p.feat + (size_t)nn[u] * p.fh * p.fw * p.feat_ld + hl * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int yy = ya[u] + (t >> 1), xx = xa[u] + (t & 1);
              const half_t* src = ok[u][t] ? fb + ((size_t)yy * p.fw + xx) * p.feat_ld : p.feat;
              v[u][t] = *reinterpret_cast<const f16x8*>(src);
            }
          }
#pragma unroll
          for (int u = 0; u < GB; ++u) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (ok[u][t]) {
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = fmaf((float)v[u][t][c], wt[u][t], a[c]);
              }
            f16x8 o;
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] = (half_t)a[c];
            *reinterpret_cast<f16x8*>(X + xoff(hw + 16 * (j0 + u), hl)) = o;
          }
        }
        // tail chunks [C, LD): coarse logits in the first ncls slots, zeros elsewhere
        if (hl < (LD - C) / 8) {
          f16x8 o[GB];
#pragma unroll
          for (int u = 0; u < GB; ++u) o[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
          if (hl == 0) {
            for (int c = 0; c < p.ncls && c < 8; ++c) {
              float cv[GB][4];
#pragma unroll
              for (int u = 0; u < GB; ++u) {
                const float* cb = p.coarse + ((size_t)nn[u] * p.ncls + c) * p.fh * p.fw;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  cv[u][t] = ok[u][t] ? cb[(ya[u] + (t >> 1)) * p.fw + xa[u] + (t & 1)] : 0.f;
              }
#pragma unroll
              for (int u = 0; u < GB; ++u) {
                float vv = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  if (ok[u][t]) vv = fmaf(cv[u][t], wt[u][t], vv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[u][e] = (e == c) ? (half_t)vv : o[u][e];     // runtime c: select, no dynamic index
              }
            }
          }
#pragma unroll
          for (int u = 0; u < GB; ++u) *reinterpret_cast<f16x8*>(X + xoff(hw + 16 * (j0 + u), C / 8 + hl)) = o[u];
        }
      }
    }
    __syncthreads();
    // ---------------- fc layers ----------------
    f16x8 wf[KSTEPS][MT];
    auto load_weights = [&](int f) {
      const half_t* wb = p.w[f] + (size_t)(wave * MT * 16 + fr) * LD + fq * 8;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int c = 0; c < MT; ++c) wf[ks][c] = *reinterpret_cast<const f16x8*>(wb + (size_t)c * 16 * LD + ks * 32);
    };
    load_weights(0);
    for (int f = 0; f < p.num_fc; ++f) {
      // two passes of 128 points (64 accumulator VGPRs beside the layer's 80 weight VGPRs): a pass's rows are rewritten
      // with the layer's output as soon as every wave has finished reading them
#pragma unroll 1
      for (int hf = 0; hf < PT / 128; ++hf) {
        f32x4 acc[MT][8];
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
          for (int q = 0; q < 8; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f16x8 pf = *reinterpret_cast<const f16x8*>(X + xoff(hf * 128 + q * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int c = 0; c < MT; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][c], pf, acc[c][q], 0, 0, 0);
          }
        }
        if (hf + 1 == PT / 128 && f + 1 < p.num_fc) load_weights(f + 1);      // in flight during the barrier and the epilogue
        __syncthreads();      // every wave is done reading these rows
#pragma unroll
        for (int c = 0; c < MT; ++c) {
          const int co = wave * MT * 16 + c * 16 + fq * 4;       // 4 consecutive couts of this lane
          const float4 bv = *reinterpret_cast<const float4*>(p.b[f] + co);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int row = hf * 128 + q * 16 + fr;
            f16x4 o;
            o[0] = (half_t)fmaxf(acc[c][q][0] + bv.x, 0.f);
            o[1] = (half_t)fmaxf(acc[c][q][1] + bv.y, 0.f);
            o[2] = (half_t)fmaxf(acc[c][q][2] + bv.z, 0.f);
            o[3] = (half_t)fmaxf(acc[c][q][3] + bv.w, 0.f);
            *reinterpret_cast<f16x4*>(X + xoff(row, co >> 3) + (co & 7) * 2) = o;
          }
        }
      }
      __syncthreads();
    }
    // ---------------- predictor + scatter ----------------
    // class by class: the class's fp32 weights (two 8-channel chunks per lane) are loaded once, then the half-wave walks
    // its 16 rows; per row the arithmetic and the reduction order of head1x1_kernel
    for (int c = 0; c < p.ncls; ++c) {
      float wv[2][8];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int kc = hl + 32 * jj;
        if (kc < KC) {
          const float4* wp = reinterpret_cast<const float4*>(p.pw + (size_t)c * LD + kc * 8);
          const float4 w0 = wp[0], w1 = wp[1];
          wv[jj][0] = w0.x; wv[jj][1] = w0.y; wv[jj][2] = w0.z; wv[jj][3] = w0.w;
          wv[jj][4] = w1.x; wv[jj][5] = w1.y; wv[jj][6] = w1.z; wv[jj][7] = w1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[jj][e] = 0.f;
        }
      }
      const float bias = p.pb[c];
#pragma unroll 4
      for (int j = 0; j < PT / 16; ++j) {
        const int row = hw + 16 * j;
        const int64_t pt = g0 + row;
        const bool live = pt < p.npts;
        float sacc = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int kc = hl + 32 * jj;
          if (kc < KC) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(X + xoff(row, kc));
#pragma unroll
            for (int e = 0; e < 8; ++e) sacc = fmaf((float)v[e], wv[jj][e], sacc);
          }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (hl == 0 && live) {
          const int n = (int)(pt / p.P);
          p.out[((size_t)n * p.ncls + c) * p.plane + (int64_t)p.idx[pt]] = sacc + bias;
        }
      }
    }
    __syncthreads();        // X is rewritten by the next tile's gather
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

bool pr_mlp_supported(int C, int ld, int ncls, int num_fc) {
  return ((C == 256 && ld == 320) || (C == 128 && ld == 192)) && ncls >= 1 && ncls <= 8 && num_fc >= 1 && num_fc <= 4;
}

int launch_pr_mlp(const half_t* feat, int N, int fh, int fw, int C, int feat_ld, const float* coarse, int ncls,
                  const int32_t* idx, int P, int H2, int W2, const half_t* const* fc_w, const float* const* fc_b,
                  int num_fc, int ld, const float* pred_w, const float* pred_b, float* out, int64_t plane,
                  hipStream_t s) {
  EMP_REQUIRE(pr_mlp_supported(C, ld, ncls, num_fc), "pr_mlp: unsupported shape C=%d ld=%d", C, ld);
  PrParams p{};
  p.feat = feat; p.N = N; p.fh = fh; p.fw = fw; p.feat_ld = feat_ld;
  p.coarse = coarse; p.ncls = ncls; p.idx = idx; p.P = P; p.H2 = H2; p.W2 = W2;
  for (int i = 0; i < num_fc; ++i) { p.w[i] = fc_w[i]; p.b[i] = fc_b[i]; }
  p.num_fc = num_fc; p.pw = pred_w; p.pb = pred_b; p.out = out; p.plane = plane;
  p.npts = (int64_t)N * P;
  p.tiles = (int)cdiv64(p.npts, 256);
  const int grid = p.tiles < 256 ? p.tiles : 256;       // one workgroup per CU
  const size_t lds = (size_t)256 * ld * 2;
  if (C == 256) {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&pr_mlp_kernel<256, 320>), (int)lds)) return rc;
    hipLaunchKernelGGL((pr_mlp_kernel<256, 320>), dim3(grid), dim3(512), lds, s, p);
  } else {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&pr_mlp_kernel<128, 192>), (int)lds)) return rc;
    hipLaunchKernelGGL((pr_mlp_kernel<128, 192>), dim3(grid), dim3(512), lds, s, p);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
