// Instance post-processing of the Panoptic-DeepLab render engines for gfx950.
// Integer outputs are bit-exact with the reference given identical fp32 head
// tensors (tests/test_gpu_postprocess.py).  Reference:
//   logits_to_prob / _harden_seg            empanada/inference/engines.py:22-30,114-121
//   _MedianQueue.get_median                 engines.py:59-66
//   find_instance_center                    empanada/inference/postprocess.py:38-76
//   group_pixels / chunked_pixel_grouping   postprocess.py:78-169
//   get_instance_cells                      engines.py:257-275
//   get_panoptic_seg + merge_semantic_and_instance   engines.py:277-292, postprocess.py:223-296
// Build flags that matter for bit-exactness (build.py sets them for this file):
//   -ffp-contract=off  -fhip-fp32-correctly-rounded-divide-sqrt
#include "common.h"

namespace emp {
namespace {

inline int grid_for(int64_t total, int per_block = 256, int cap = 256 * 16) {
  int64_t g = (total + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* sh, uint32_t* total) {
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

// ---------------------------------------------------------------------------
// probabilities
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sigmoid_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    out[i] = 1.0f / (1.0f + expf(-in[i]));
}
__global__ void __launch_bounds__(256) softmax_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                      int64_t plane, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t n = i / plane, p = i - n * plane;
    const float* b = in + n * C * plane + p;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, b[c * plane]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(b[c * plane] - m);
    float* o = out + n * C * plane + p;
    for (int c = 0; c < C; ++c) o[c * plane] = expf(b[c * plane] - m) / s;
  }
}

// ---------------------------------------------------------------------------
// per-pixel median of ks maps (exact selection of the middle order statistic)
// ---------------------------------------------------------------------------
constexpr int MAX_KS = 15;
struct MedianPtrs { const float* p[MAX_KS]; };
__global__ void __launch_bounds__(256) median_kernel(MedianPtrs ptrs, int ks, float* __restrict__ out, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    float v[MAX_KS];
#pragma unroll
    for (int k = 0; k < MAX_KS; ++k) v[k] = k < ks ? ptrs.p[k][i] : INFINITY;
    // full insertion sort of MAX_KS registers (padding = +inf stays at the end)
#pragma unroll
    for (int a = 1; a < MAX_KS; ++a) {
#pragma unroll
      for (int b = a; b > 0; --b) {
        float lo = fminf(v[b - 1], v[b]), hi = fmaxf(v[b - 1], v[b]);
        v[b - 1] = lo;
        v[b] = hi;
      }
    }
    const int mid = (ks - 1) >> 1;
    float r = v[0];
#pragma unroll
    for (int k = 1; k < MAX_KS; ++k) r = (k == mid) ? v[k] : r;
    out[i] = r;
  }
}

// recursive median over a run of consecutive slices in ONE launch (the z recursion is per pixel, so
// a thread walks its pixel through the slices, carrying the filtered history in registers):
//   hist: (mid, count) filtered maps preceding the run (only if n_hist == mid), raw: (n_raw, count) raw maps
//   out[j] = median(hist/out[j-mid..j-1], raw[j..j+mid]) for j in [0, n_out); needs n_raw >= n_out + mid
// out MAY BE raw (in place: map j is in registers before out[j] is stored, maps beyond j + mid are read later), so
// neither pointer is __restrict__: a whole slab is filtered without a second slab-sized buffer (multigpu.py)
template <int KS>
__global__ void __launch_bounds__(256) median_recursive_kernel(const float* __restrict__ hist,
                                                               const float* raw, int n_out,
                                                               float* out, size_t count) {
  constexpr int MID = (KS - 1) / 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    float h[MID > 0 ? MID : 1];
#pragma unroll
    for (int k = 0; k < MID; ++k) h[k] = hist[(size_t)k * count + i];
    float ahead[MID + 1];  // raw[j .. j+mid]
#pragma unroll
    for (int k = 0; k <= MID; ++k) ahead[k] = raw[(size_t)k * count + i];
    for (int j = 0; j < n_out; ++j) {
      float v[KS];
#pragma unroll
      for (int k = 0; k < MID; ++k) v[k] = h[k];
#pragma unroll
      for (int k = 0; k <= MID; ++k) v[MID + k] = ahead[k];
#pragma unroll
      for (int a = 1; a < KS; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
          float lo = fminf(v[b - 1], v[b]), hi = fmaxf(v[b - 1], v[b]);
          v[b - 1] = lo;
          v[b] = hi;
        }
      }
      const float m = v[MID];
      out[(size_t)j * count + i] = m;
#pragma unroll
      for (int k = 0; k + 1 < MID; ++k) h[k] = h[k + 1];
      if (MID > 0) h[MID - 1] = m;
#pragma unroll
      for (int k = 0; k < MID; ++k) ahead[k] = ahead[k + 1];
      if (j + 1 < n_out) ahead[MID] = raw[(size_t)(j + 1 + MID) * count + i];
    }
  }
}

// ---------------------------------------------------------------------------
// centre NMS -> bitmask (one bit per heat-map pixel, row-major)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nms_mask_kernel(const float* __restrict__ ctr, int h, int w, float thr, int k,
                                                       uint32_t* __restrict__ mask, int words_per_img) {
  const int n = blockIdx.y;
  const int hw = h * w;
  const float* c = ctr + (size_t)n * hw;
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (i < hw) {
    const int y = i / w, x = i - y * w;
    float v = c[i];
    v = v > thr ? v : -1.0f;  // F.threshold(x, thr, -1)
    if (v > 0.f) {
      const int p = k >> 1;
      float m = -INFINITY;
      for (int dy = 0; dy < k; ++dy) {
        int yy = y - p + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = 0; dx < k; ++dx) {
          int xx = x - p + dx;
          if (xx < 0 || xx >= w) continue;
          float u = c[yy * w + xx];
          u = u > thr ? u : -1.0f;
          m = fmaxf(m, u);
        }
      }
      keep = (v == m);
    }
  }
  const unsigned long long b = __ballot(keep);
  const int lane = threadIdx.x & 63;
  const int word = (blockIdx.x * 256 + (threadIdx.x & ~63)) >> 5;
  if (lane == 0) {
    uint32_t* mw = mask + (size_t)n * words_per_img;
    if (word < words_per_img) mw[word] = (uint32_t)b;
    if (word + 1 < words_per_img) mw[word + 1] = (uint32_t)(b >> 32);
  }
}

// one block per image: ordered expansion of the bitmask into (y,x) centres
__global__ void __launch_bounds__(256) centers_kernel(const uint32_t* __restrict__ mask, int words_stride,
                                                      int words_per_img, int w, int32_t* __restrict__ centers,
                                                      int32_t* __restrict__ num, int max_centers) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.x;
  const uint32_t* mw = mask + (size_t)n * words_stride;
  int32_t* out = centers + (size_t)n * max_centers * 2;
  const int per = (words_per_img + 255) / 256;
  const int w0 = threadIdx.x * per, w1 = min(words_per_img, w0 + per);
  uint32_t cnt = 0;
  for (int i = w0; i < w1; ++i) cnt += __popc(mw[i]);
  uint32_t total;
  uint32_t pos = block_excl_scan(cnt, sh, &total);
  for (int i = w0; i < w1; ++i) {
    uint32_t m = mw[i];
    while (m) {
      int b = __ffs(m) - 1;
      m &= m - 1;
      if (pos < (uint32_t)max_centers) {
        int pix = i * 32 + b;
        int y = pix / w;
        out[2 * pos] = y;
        out[2 * pos + 1] = pix - y * w;
      }
      ++pos;
    }
  }
  if (threadIdx.x == 0) num[n] = (int32_t)total;  // un-clamped: caller checks against max_centers
}

// ---------------------------------------------------------------------------
// nearest-centre voting + nearest up-sampling of the id map.
// distance arithmetic = torch.norm over (dy,dx): sqrt(fma(dx,dx,fl(dy*dy))) in
// fp32 (DESIGN.md "distance arithmetic"), strict '<' so the lowest centre index
// wins ties; K > 20 reproduces the chunked path's 1e5 initial distance.
// ---------------------------------------------------------------------------
constexpr int CTR_TILE = 1024;

// ---- dense scenes (round 6, late): a uniform grid over the centres ----------------------------------------------------
// The scan below costs pixels x centres: a 4096^2 slice of BASELINE configs[3] with 4 276 objects is 4.5 ms of voting next to a
// 12 ms forward.  From GRID_MIN centres per image the centres are binned (bins of B x B in the scaled coordinates, B a power of two
// with ~2+ centres per bin and at most GRID_BINS bins) and a pixel searches the rings of bins around its voted position outwards
// until the next ring cannot hold a centre at the best distance found (votes that would walk a quarter of the grid: one pass over all centres).  The RESULT IS THE SCAN'S, bit for bit: the same fp32
// expressions per candidate, the winner = the lowest index among the candidates of minimal ROUNDED distance below the 1e5
// start value (the scan's strict '<' in index order), and a ring is skipped only when its exact lower bound on the distance
// exceeds the best fp32 sum by more than the arithmetic's rounding (a relative 4e-6 against ~3e-7).  Non-finite votes keep id 0
// as in the scan.  Per image in the work buffer: hdr {log2 B, GH, GW, used} | bin_start[GRID_BINS + 1] | sorted {cy, cx, id, -}[GRID_KMAX].
constexpr int GRID_MIN = 192, GRID_KMAX = 16384, GRID_BINS = 4096;
constexpr size_t GRID_HDR_INTS = 4 + GRID_BINS + 4;      // hdr | bin_start (+ pad to 16 B)
constexpr size_t GRID_IMG_BYTES = GRID_HDR_INTS * 4 + (size_t)GRID_KMAX * 16;

__global__ void __launch_bounds__(1024) ctr_grid_build_kernel(const int32_t* __restrict__ centers, const int32_t* __restrict__ num,
                                                              int max_centers, int step, int Hs, int Ws, char* __restrict__ gridbuf) {
  __shared__ int cnt[GRID_BINS];
  __shared__ uint32_t sh[16];
  const int n = blockIdx.x, tid = threadIdx.x;
  int K = num[n];
  K = K < max_centers ? K : max_centers;
  int* hdr = reinterpret_cast<int*>(gridbuf + (size_t)n * GRID_IMG_BYTES);
  if (K < GRID_MIN || K > GRID_KMAX) {
    if (tid == 0) hdr[3] = 0;
    return;
  }
  int lb = 2;      // bins of 4 x 4 at the densest (every second cell of a coarse map a centre)
  while ((int64_t)((Hs + (1 << lb) - 1) >> lb) * ((Ws + (1 << lb) - 1) >> lb) > GRID_BINS || ((int64_t)K << (2 * lb)) < 2ll * Hs * Ws) ++lb;
  const int GH = (Hs + (1 << lb) - 1) >> lb, GW = (Ws + (1 << lb) - 1) >> lb, nb = GH * GW;
  int* start = hdr + 4;
  float4* sorted = reinterpret_cast<float4*>(hdr + GRID_HDR_INTS);
  const int32_t* cn = centers + (size_t)n * max_centers * 2;
  for (int b = tid; b < GRID_BINS; b += 1024) cnt[b] = 0;
  __syncthreads();
  for (int k = tid; k < K; k += 1024) atomicAdd(&cnt[((cn[2 * k] * step) >> lb) * GW + ((cn[2 * k + 1] * step) >> lb)], 1);
  __syncthreads();
  // exclusive scan over the bins: four consecutive bins per thread, a block scan of the per-thread sums
  const int b0 = tid * 4;
  int c[4], tsum = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { c[e] = cnt[b0 + e]; tsum += c[e]; }
  const int lane = tid & 63, wv = tid >> 6;
  uint32_t inc = wave_incl_scan((uint32_t)tsum, lane);
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int e = 0; e < wv; ++e) base += sh[e];
  int pos = (int)(base + inc) - tsum;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (b0 + e < nb) start[b0 + e] = pos;
    cnt[b0 + e] = pos;                          // the bin's write cursor
    pos += c[e];
  }
  if (tid == 0) { hdr[0] = lb; hdr[1] = GH; hdr[2] = GW; hdr[3] = 1; start[nb] = K; }
  __syncthreads();
  for (int k = tid; k < K; k += 1024) {
    const int cyi = cn[2 * k] * step, cxi = cn[2 * k + 1] * step;
    const int at = atomicAdd(&cnt[(cyi >> lb) * GW + (cxi >> lb)], 1);      // (order inside a bin is free: ties go by the stored id)
    sorted[at] = make_float4((float)cyi, (float)cxi, __int_as_float(k + 1), 0.f);
  }
}

__global__ void __launch_bounds__(256) group_pixels_kernel(const float* __restrict__ offsets, int h, int w,
                                                           const int32_t* __restrict__ centers,
                                                           const int32_t* __restrict__ num, int max_centers, int step,
                                                           int up, int32_t* __restrict__ cells, const char* __restrict__ gridbuf) {
  __shared__ float cy[CTR_TILE], cx[CTR_TILE];
  const int n = blockIdx.y;
  const int hw = h * w;
  int K = num[n];
  K = K < max_centers ? K : max_centers;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool active = i < hw;
  const int y = active ? i / w : 0, x = active ? i - (i / w) * w : 0;
  float ly = 0.f, lx = 0.f;
  if (active) {
    const float* off = offsets + (size_t)n * 2 * hw;
    ly = (float)(y * step) + off[i];
    lx = (float)(x * step) + off[hw + i];
  }
  const int* ghdr = gridbuf ? reinterpret_cast<const int*>(gridbuf + (size_t)n * GRID_IMG_BYTES) : nullptr;
  if (ghdr && ghdr[3]) {      // (block-uniform: one image per blockIdx.y)
    if (!active) return;
    int id = 0;
    if (ly - ly == 0.f && lx - lx == 0.f) {      // finite votes only: a NaN / inf vote matches no centre in the scan either
      const int lb = ghdr[0], GH = ghdr[1], GW = ghdr[2];
      const int* start = ghdr + 4;
      const float4* sorted = reinterpret_cast<const float4*>(ghdr + GRID_HDR_INTS);
      const int Hs = h * step, Ws = w * step;
      const int by = (int)floorf(fminf(fmaxf(ly, 0.f), (float)(Hs - 1))) >> lb;
      const int bx = (int)floorf(fminf(fmaxf(lx, 0.f), (float)(Ws - 1))) >> lb;
      float best = 1e5f, best_s = 1e10f, best_hi = 1e10f * 1.000004f;
      auto scan = [&](int a, int b) {
        for (int t = a; t < b; ++t) {
          const float4 c = sorted[t];
          const float dy = c.x - ly, dx = c.y - lx;
          const float dy2 = dy * dy;
          const float s2 = __builtin_fmaf(dx, dx, dy2);
          if (s2 > best_hi) continue;
          const float d = __builtin_sqrtf(s2);
          const int idx = __float_as_int(c.z);
          if (d < best || (d == best && idx < id)) { best = d; best_s = s2; best_hi = s2 * 1.000004f; id = idx; }
        }
      };
      for (int r = 0;; ++r) {
        const int y0 = by - r, y1 = by + r, x0 = bx - r, x1 = bx + r;
        if (y0 < 0 && y1 >= GH && x0 < 0 && x1 >= GW) break;      // the ring lies outside the grid: every bin has been seen
        const int xa = x0 > 0 ? x0 : 0, xb = x1 < GW - 1 ? x1 : GW - 1;
        if (y0 >= 0) scan(start[y0 * GW + xa], start[y0 * GW + xb + 1]);                 // top row of the ring: contiguous bins
        if (r > 0 && y1 < GH) scan(start[y1 * GW + xa], start[y1 * GW + xb + 1]);        // bottom row
        const int ya = y0 + 1 > 0 ? y0 + 1 : 0, yb = y1 - 1 < GH - 1 ? y1 - 1 : GH - 1;
        for (int yy = ya; yy <= yb; ++yy) {                                              // the two side bins of the inner rows
          if (x0 >= 0) scan(start[yy * GW + x0], start[yy * GW + x0 + 1]);
          if (x1 < GW) scan(start[yy * GW + x1], start[yy * GW + x1 + 1]);
        }
        // a centre of ring r + 1 or beyond is at least r * B away from the (clamped) vote along one axis
        const float lbn = (float)(r << lb);
        if (lbn * lbn * (1.f - 4e-6f) > best_s) break;
        if (r >= 2) {
          // a vote far outside the image, or with no centre below 1e5 at all, would walk every ring: once the bins still to visit
          // (up to the ring the best distance so far calls for; nothing found yet: the rings seen so far) are a quarter of the
          // grid, ONE pass over every centre is cheaper (the same rule: candidates seen twice change nothing)
          const int side = 2 * r + 1;
          int todo = side * side;
          if (id) {
            const int need = min(((int)(__builtin_sqrtf(best_s) * 1.00001f) >> lb) + 1, 1 << 12);
            todo = (2 * need + 1) * (2 * need + 1) - side * side;
          }
          if (todo * 4 >= GH * GW) {
            scan(0, start[GH * GW]);
            break;
          }
        }
      }
      (void)best_s;
    }
    const int W = w * up;
    int32_t* o = cells + (size_t)n * hw * up * up + (size_t)(y * up) * W + x * up;
    for (int dy = 0; dy < up; ++dy)
      for (int dx = 0; dx < up; ++dx) o[(size_t)dy * W + dx] = id;
    return;
  }
  float best = (K > 20) ? 1e5f : INFINITY;
  // squared-distance screen (round 5; exact): sqrt is monotone, so a centre whose fp32 sum s = fma(dx,dx,fl(dy*dy)) is
  // strictly greater than the sum behind the running best cannot have a smaller rounded distance -- it is skipped without
  // the correctly rounded sqrt (most of this loop at K ~ 1 700-4 500 centres per 1024^2 tile, the fine-boundary case).  A
  // sum that is not greater goes through the reference's comparison of the rounded distances unchanged (two sums may round
  // to one distance: the first centre keeps the pixel).  1e10f = (1e5f)^2 exactly; NaN sums take the exact path.
  float best_s = (K > 20) ? 1e10f : INFINITY;
  int id = 0;
  const int32_t* cn = centers + (size_t)n * max_centers * 2;
  for (int k0 = 0; k0 < K; k0 += CTR_TILE) {
    const int kn = min(CTR_TILE, K - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < kn; t += 256) {
      cy[t] = (float)(cn[2 * (k0 + t)] * step);
      cx[t] = (float)(cn[2 * (k0 + t) + 1] * step);
    }
    __syncthreads();
    if (active) {
      for (int t = 0; t < kn; ++t) {
        const float dy = cy[t] - ly, dx = cx[t] - lx;
        const float dy2 = dy * dy;  // rounded on its own: this file is built with -ffp-contract=off
        const float s = __builtin_fmaf(dx, dx, dy2);
        if (s > best_s) continue;
        const float d = __builtin_sqrtf(s);  // correctly rounded sqrt (build flag)
        const bool first = (K <= 20) && (k0 + t == 0);  // argmin always yields an index
        if (d < best || first) { best = d; best_s = s; id = k0 + t + 1; }
      }
    }
  }
  if (!active) return;
  const int W = w * up;
  int32_t* o = cells + (size_t)n * hw * up * up + (size_t)(y * up) * W + x * up;
  for (int dy = 0; dy < up; ++dy)
    for (int dx = 0; dx < up; ++dx) o[(size_t)dy * W + dx] = id;
}

// ---------------------------------------------------------------------------
// panoptic merge
// work layout per image: counts[(max_ids+1) * CLS] int32 | stuff[CLS] int32 | map[max_ids+1] int64
// ---------------------------------------------------------------------------
struct ThingList { int32_t n; int32_t cls[16]; };

__device__ __forceinline__ int harden(const float* __restrict__ sem, int C, int64_t plane, int64_t pix, float thr) {
  if (C == 1) return sem[pix] >= thr ? 1 : 0;
  int best = 0;
  float bv = sem[pix];
  for (int c = 1; c < C; ++c) {
    float v = sem[c * plane + pix];
    if (v > bv) { bv = v; best = c; }
  }
  return best;
}
__device__ __forceinline__ bool is_thing(const ThingList& tl, int cls) {
  bool r = false;
  for (int i = 0; i < tl.n; ++i) r |= (tl.cls[i] == cls);
  return r;
}

// counts[(id*CLS + cls)] (things) and counts[(max_ids+1)*CLS + cls] (pixels outside every
// instance) per image.  Each thread run-length-compresses 8 consecutive pixels, the block
// aggregates its runs in a 256-entry LDS hash table and only distinct keys reach global atomics
// (one block sees a handful of instances; the background key would otherwise serialise).
__global__ void __launch_bounds__(256) merge_count_kernel(const float* __restrict__ sem, const int32_t* __restrict__ cells,
                                                          int C, int CLS, int64_t plane, float thr, ThingList tl,
                                                          int max_ids, int32_t* __restrict__ counts_all,
                                                          size_t img_stride_i32) {
  __shared__ int tkey[256];
  __shared__ int tval[256];
  const int n = blockIdx.y;
  const float* s = sem + (size_t)n * C * plane;
  const int32_t* ce = cells + (size_t)n * plane;
  int32_t* counts = counts_all + (size_t)n * img_stride_i32;
  tkey[threadIdx.x] = -1;
  tval[threadIdx.x] = 0;
  __syncthreads();
  const int stuff_base = (max_ids + 1) * CLS;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  int run_key = -1, run_cnt = 0;
  auto flush = [&]() {
    if (run_cnt == 0) return;
    unsigned h = ((unsigned)run_key * 2654435761u) >> 24;
    bool done = false;
    for (int probe = 0; probe < 256 && !done; ++probe) {
      int old = atomicCAS(&tkey[h], -1, run_key);
      if (old == -1 || old == run_key) {
        atomicAdd(&tval[h], run_cnt);
        done = true;
      } else {
        h = (h + 1) & 255u;
      }
    }
    if (!done) atomicAdd(&counts[run_key], run_cnt);  // table full (never seen in practice)
    run_cnt = 0;
  };
  for (int j = 0; j < 8; ++j) {
    int64_t p = p0 + j;
    if (p >= plane) break;
    int cls = harden(s, C, plane, p, thr);
    int id = is_thing(tl, cls) ? ce[p] : 0;
    if (id > max_ids || id < 0) id = 0;  // cannot happen when max_ids bounds the centre count
    int key = id > 0 ? id * CLS + cls : stuff_base + cls;
    if (key != run_key) { flush(); run_key = key; }
    ++run_cnt;
  }
  flush();
  __syncthreads();
  const int k = tkey[threadIdx.x];
  if (k >= 0) atomicAdd(&counts[k], tval[threadIdx.x]);
}

// one block per image: ids in ascending order get per-class consecutive numbers
__global__ void __launch_bounds__(256) merge_assign_kernel(int CLS, int max_ids, int64_t divisor,
                                                           int32_t* __restrict__ counts_all,
                                                           int64_t* __restrict__ map_all, size_t img_stride_i32,
                                                           size_t map_stride) {
  __shared__ int cls_of[256];
  __shared__ int run[32];
  const int n = blockIdx.x;
  const int32_t* counts = counts_all + (size_t)n * img_stride_i32;
  int64_t* map = map_all + (size_t)n * map_stride;
  if (threadIdx.x < 32) run[threadIdx.x] = 1;  // class_id_tracker starts at 1
  __syncthreads();
  for (int id0 = 1; id0 <= max_ids; id0 += 256) {
    const int id = id0 + threadIdx.x;
    int cls = -1;
    if (id <= max_ids) {
      int bc = 0;
      for (int c = 0; c < CLS; ++c) {
        int v = counts[id * CLS + c];
        if (v > bc) { bc = v; cls = c; }  // torch.mode: most frequent, ties -> smallest value
      }
    }
    cls_of[threadIdx.x] = cls;
    __syncthreads();
    if (cls >= 0) {
      int rank = 0;
      for (int t = 0; t < (int)threadIdx.x; ++t) rank += (cls_of[t] == cls);
      map[id] = (int64_t)cls * divisor + (int64_t)(run[cls] + rank);
    } else if (id <= max_ids) {
      map[id] = -1;
    }
    __syncthreads();
    if (threadIdx.x < CLS && threadIdx.x < 32) {
      int tot = 0;
      for (int t = 0; t < 256; ++t) tot += (cls_of[t] == (int)threadIdx.x);
      run[threadIdx.x] += tot;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) merge_write_kernel(const float* __restrict__ sem, const int32_t* __restrict__ cells,
                                                          int C, int CLS, int64_t plane, float thr, ThingList tl,
                                                          int max_ids, int64_t divisor, int64_t stuff_area,
                                                          int64_t void_label, const int32_t* __restrict__ stuff_all,
                                                          const int64_t* __restrict__ map_all, size_t img_stride_i32,
                                                          size_t map_stride, int64_t* __restrict__ pan) {
  const int n = blockIdx.y;
  const float* s = sem + (size_t)n * C * plane;
  const int32_t* ce = cells + (size_t)n * plane;
  const int32_t* stuff = stuff_all + (size_t)n * img_stride_i32;
  const int64_t* map = map_all + (size_t)n * map_stride;
  int64_t* o = pan + (size_t)n * plane;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < plane; p += (int64_t)gridDim.x * 256) {
    int cls = harden(s, C, plane, p, thr);
    int64_t v = void_label;
    if (is_thing(tl, cls)) {
      int id = ce[p];
      if (id > 0 && id <= max_ids) {
        int64_t m = map[id];
        if (m >= 0) v = m;
      }
    } else if ((int64_t)stuff[cls] >= stuff_area) {
      v = (int64_t)cls * divisor;
    }
    o[p] = v;
  }
}

}  // namespace
}  // namespace emp

using namespace emp;

extern "C" {

int emp_logits_to_prob(const float* d_logits, float* d_prob, int N, int C, int H, int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  EMP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "logits_to_prob: bad shape");
  const int64_t plane = (int64_t)H * W;
  if (C == 1) {
    int64_t total = (int64_t)N * plane;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(total)), dim3(256), 0, s, d_logits, d_prob, total);
  } else {
    int64_t total = (int64_t)N * plane;
    hipLaunchKernelGGL(softmax_kernel, dim3(grid_for(total)), dim3(256), 0, s, d_logits, d_prob, C, plane, total);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int emp_median_slices(const float* const* h_slice_ptrs, int ks, float* d_out, size_t count, void* stream) {
  EMP_REQUIRE(ks >= 1 && ks <= MAX_KS && (ks & 1), "median: kernel size %d must be odd and <= %d", ks, MAX_KS);
  MedianPtrs ptrs;
  for (int k = 0; k < MAX_KS; ++k) ptrs.p[k] = k < ks ? h_slice_ptrs[k] : h_slice_ptrs[0];
  hipLaunchKernelGGL(median_kernel, dim3(grid_for((int64_t)count)), dim3(256), 0, (hipStream_t)stream, ptrs, ks, d_out,
                     count);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int emp_median_recursive(const float* d_hist, const float* d_raw, int n_raw, int ks, int n_out, float* d_out,
                         size_t count, void* stream) {
  EMP_REQUIRE(d_hist && d_raw && d_out && count > 0, "median_recursive: null argument");
  EMP_REQUIRE((ks & 1) && ks >= 3 && ks <= MAX_KS, "median_recursive: kernel size %d must be odd, 3..%d", ks, MAX_KS);
  EMP_REQUIRE(n_out >= 1 && n_raw >= n_out + (ks - 1) / 2, "median_recursive: need n_out + mid raw maps");
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for((int64_t)count);
  switch (ks) {
    case 3: hipLaunchKernelGGL(median_recursive_kernel<3>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    case 5: hipLaunchKernelGGL(median_recursive_kernel<5>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    case 7: hipLaunchKernelGGL(median_recursive_kernel<7>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    case 9: hipLaunchKernelGGL(median_recursive_kernel<9>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    case 11: hipLaunchKernelGGL(median_recursive_kernel<11>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    case 13: hipLaunchKernelGGL(median_recursive_kernel<13>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
    default: hipLaunchKernelGGL(median_recursive_kernel<15>, dim3(grid), dim3(256), 0, s, d_hist, d_raw, n_out, d_out, count); break;
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

static inline size_t cells_mask_bytes(int N, int h, int w) {
  const size_t words = ((size_t)h * w + 31) / 32 + 2;
  return ((size_t)N * words * 4 + 255) & ~(size_t)255;
}
size_t emp_instance_cells_work_bytes(int N, int h, int w) {      // NMS bit mask | per-image centre grid (round 6)
  return cells_mask_bytes(N, h, w) + (size_t)N * GRID_IMG_BYTES + 256;
}

int emp_instance_cells(const float* d_ctr_hmp, const float* d_offsets, int N, int h, int w, float nms_threshold,
                       int nms_kernel, int step, int up, int32_t* d_cells, int32_t* d_centers, int32_t* d_num_centers,
                       int max_centers, void* d_work, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  EMP_REQUIRE(N > 0 && h > 0 && w > 0 && nms_kernel >= 1 && up >= 1 && max_centers > 0, "instance_cells: bad args");
  const int hw = h * w;
  const int words = (hw + 31) / 32 + 2;
  uint32_t* mask = (uint32_t*)d_work;
  const int nb = cdiv(hw, 256);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, N), dim3(256), 0, s, d_ctr_hmp, h, w, nms_threshold, nms_kernel, mask,
                     words);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(centers_kernel, dim3(N), dim3(256), 0, s, mask, words, (hw + 31) / 32, w, d_centers,
                     d_num_centers, max_centers);
  EMP_LAUNCH_CHECK();
  // dense scenes: bin the centres (one workgroup per image; images below GRID_MIN centres mark their grid unused and take the scan)
  const char* grid_env = getenv("EMP_VOTE_GRID");      // =0: the scan for every image (A/B; bit-identical)
  char* gridbuf = nullptr;
  if (!(grid_env && grid_env[0] == '0') && (int64_t)h * step < (1 << 20) && (int64_t)w * step < (1 << 20)) {
    gridbuf = (char*)d_work + cells_mask_bytes(N, h, w);
    hipLaunchKernelGGL(ctr_grid_build_kernel, dim3(N), dim3(1024), 0, s, d_centers, d_num_centers, max_centers, step, h * step, w * step, gridbuf);
    EMP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(group_pixels_kernel, dim3(nb, N), dim3(256), 0, s, d_offsets, h, w, d_centers, d_num_centers,
                     max_centers, step, up, d_cells, gridbuf);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

static inline size_t merge_img_stride_i32(int CLS, int max_ids) {
  size_t n = (size_t)(max_ids + 1) * CLS + CLS;
  return (n + 3) & ~(size_t)3;
}
size_t emp_panoptic_merge_work_bytes(int N, int C, int max_ids) {
  const int CLS = C < 2 ? 2 : C;
  return (size_t)N * (merge_img_stride_i32(CLS, max_ids) * 4 + (size_t)(max_ids + 1) * 8) + 256;
}

int emp_panoptic_merge(const float* d_sem, const int32_t* d_cells, int N, int C, int H, int W, float confidence_thr,
                       const int32_t* h_thing_list, int n_things, int64_t label_divisor, int64_t stuff_area,
                       int64_t void_label, int max_ids, int64_t* d_pan, void* d_work, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  EMP_REQUIRE(N > 0 && C > 0 && C <= 32 && H > 0 && W > 0 && max_ids >= 0, "panoptic_merge: bad args");
  EMP_REQUIRE(n_things >= 0 && n_things <= 16, "panoptic_merge: at most 16 thing classes");
  const int CLS = C < 2 ? 2 : C;
  ThingList tl;
  tl.n = n_things;
  for (int i = 0; i < 16; ++i) tl.cls[i] = i < n_things ? h_thing_list[i] : -1;
  const size_t stride = merge_img_stride_i32(CLS, max_ids);
  int32_t* counts = (int32_t*)d_work;
  int32_t* stuff = counts + (size_t)(max_ids + 1) * CLS;  // + n*stride inside the kernels
  int64_t* map = (int64_t*)((char*)d_work + (((size_t)N * stride * 4 + 15) & ~(size_t)15));
  const size_t map_stride = (size_t)max_ids + 1;
  EMP_CHECK_HIP(hipMemsetAsync(d_work, 0, (size_t)N * stride * 4, s));
  const int64_t plane = (int64_t)H * W;
  const int nb = (int)cdiv64(plane, 256 * 8);
  hipLaunchKernelGGL(merge_count_kernel, dim3(nb, N), dim3(256), 0, s, d_sem, d_cells, C, CLS, plane, confidence_thr,
                     tl, max_ids, counts, stride);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge_assign_kernel, dim3(N), dim3(256), 0, s, CLS, max_ids, label_divisor, counts, map, stride,
                     map_stride);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(merge_write_kernel, dim3(grid_for(plane, 256, 2048), N), dim3(256), 0, s, d_sem, d_cells, C, CLS,
                     plane, confidence_thr, tl, max_ids, label_divisor, stuff_area, void_label, stuff, map, stride,
                     map_stride, d_pan);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // extern "C"
