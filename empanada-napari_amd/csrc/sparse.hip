// Label-map <-> run-length kernels and the host-side range algebra of the 3-D
// stitching path.
//
// Device (gfx950):
//   emp_ccl8            8-connected components of EQUAL non-zero label, numbered 1..K per image in
//                       raster order of each component's first pixel
//                       (skimage.measure.label / cc3d as used by empanada/inference/rle.py:18-24,
//                       Engine2d.force_connected empanada_napari/inference.py:263-279)
//   emp_rle_extract     runs of equal non-zero label over the raveled image, in raster order
//                       (regionprops(...).coords -> rle_encode, rle.py:73-81, array_utils.py:213-239)
//   emp_rle_fill        run list -> dense volume (numpy_fill_instances, array_utils.py:754-766)
// Host (C++, sorted-sweep restatements of the numba kernels):
//   emp_rle_pair_intersections   array_utils.py:344-407 (two-pointer merge)
//   emp_ranges_vote              array_utils.py:461-639 (k-of-n vote; thr 1 = join_ranges :658-699)
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

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

// A label map as the kernels read it.  RANGE (round 6): the class isolation of rle.py:46-48 /
// inference.py:270-272 -- "keep the labels of [lo, hi), everything else is background" -- applied at the read, on the int64 (or
// int32) panoptic map itself, instead of a torch.where pass (and an int64 -> int32 pass) per class in front of the kernels.
template <typename T, bool RANGE>
struct Img {
  const T* p;
  int64_t lo, hi;
  __device__ __forceinline__ int32_t operator[](size_t i) const {
    const T v = p[i];
    if (RANGE) return (v >= (T)lo && v < (T)hi) ? (int32_t)v : 0;
    return (int32_t)v;
  }
  __device__ __host__ __forceinline__ Img at(size_t off) const { return Img{p + off, lo, hi}; }
};
typedef Img<int32_t, false> ImgPlain;

// ---------------------------------------------------------------------------
// connected components: lock-free union-find on linear pixel indices; a root is the smallest
// index of its component, so ranking the roots in index order IS raster order of first pixels.
// ---------------------------------------------------------------------------
// parent words are read past the per-CU L1 (agent-scope relaxed loads): other workgroups update them
// with atomicMin while this one walks the tree
__device__ __forceinline__ int uf_load(int* parent, int a) {
  return __hip_atomic_load(&parent[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(int* parent, int a) {
  int p = uf_load(parent, a);
  while (p != a) {
    a = p;
    p = uf_load(parent, a);
  }
  return a;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }   // a > b: hang the larger root under the smaller
    int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;
  }
}

template <class IM>
__global__ void __launch_bounds__(256) ccl_init_kernel(const IM in, int* __restrict__ parent,
                                                       int64_t total, int hw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    parent[i] = in[i] != 0 ? (int)(i % hw) : -1;
}

// 2-D initialisation: a pixel starts under the FIRST pixel of its horizontal run (consecutive equal labels in a row),
// as far as the run lies inside the pixel's wave -- one ballot instead of one union per pixel.  A run that continues
// from the previous wave is linked by ccl_merge_kernel (lane 0).  Same indexing as the merge kernel: 64 consecutive
// image-local pixels per wave.
template <class IM>
__global__ void __launch_bounds__(256) ccl_init_rows_kernel(const IM in, int* __restrict__ parent_all,
                                                            int hw, int w) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const IM im = in.at((size_t)n * hw);
  const bool live = p < hw;
  const int32_t v = live ? im[p] : 0;
  const int x = live ? p % w : 0;
  const bool same = v != 0 && x > 0 && im[p - 1] == v;
  const unsigned long long starts = __ballot(!same);
  const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  const int s = upto ? 63 - __clzll(upto) : 0;        // no start bit at or below me: the run came in from the previous wave
  if (live) parent_all[(size_t)n * hw + p] = v != 0 ? p - (lane - s) : -1;
}

// Unions across rows, only where they are not implied (runs are already joined):
//   up            unless the left neighbour and the up-left neighbour are equal too (the left pixel's `up` union covers it)
//   up-left       only if `up` differs and `left` differs (else the left pixel's `up` union is this one)
//   up-right      only if `up` differs and `right` differs (else the right pixel's `up` union is this one)
// In the interior of a blob no pixel issues a union: the atomics are left to run ends and corners.
template <class IM>
__global__ void __launch_bounds__(256) ccl_merge_kernel(const IM in, int* __restrict__ parent_all,
                                                        int h, int w) {
  const int n = blockIdx.y;
  const int hw = h * w;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const IM im = in.at((size_t)n * hw);
  int* parent = parent_all + (size_t)n * hw;
  const int32_t v = im[p];
  if (v == 0) return;
  const int y = p / w, x = p - y * w;
  const bool left = x > 0 && im[p - 1] == v;
  if ((threadIdx.x & 63) == 0 && left) uf_union(parent, p, p - 1);      // run continuing across the wave boundary
  if (y > 0) {
    const bool up = im[p - w] == v;
    const bool ul = x > 0 && im[p - w - 1] == v;
    if (up) {
      if (!(left && ul)) uf_union(parent, p, p - w);
    } else {
      if (ul && !left) uf_union(parent, p, p - w - 1);
      const bool right = x + 1 < w && im[p + 1] == v;
      if (x + 1 < w && im[p - w + 1] == v && !right) uf_union(parent, p, p - w + 1);
    }
  }
}

// 26-connected components of a volume (skimage.measure.label on a 3-D array, full connectivity): the 13 neighbours
// that precede a voxel in raster order
template <class IM>
__global__ void __launch_bounds__(256) ccl_merge3d_kernel(const IM im, int* __restrict__ parent,
                                                          int d, int h, int w) {
  const int64_t total = (int64_t)d * h * w;
  const int64_t p64 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p64 >= total) return;
  const int p = (int)p64;
  const int32_t v = im[p];
  if (v == 0) return;
  const int hw = h * w;
  const int z = p / hw, r = p - z * hw;
  const int y = r / w, x = r - y * w;
  if (x > 0 && im[p - 1] == v) uf_union(parent, p, p - 1);
  if (y > 0) {
    if (im[p - w] == v) uf_union(parent, p, p - w);
    if (x > 0 && im[p - w - 1] == v) uf_union(parent, p, p - w - 1);
    if (x + 1 < w && im[p - w + 1] == v) uf_union(parent, p, p - w + 1);
  }
  if (z > 0) {
    const int q = p - hw;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      if ((unsigned)(y + dy) >= (unsigned)h) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        if ((unsigned)(x + dx) >= (unsigned)w) continue;
        const int n = q + dy * w + dx;
        if (im[n] == v) uf_union(parent, p, n);
      }
    }
  }
}

// grey erosion (op 0: min) / dilation (op 1: max) of a label volume with the 3-D cross (skimage.morphology.erosion /
// dilation defaults: footprint = generate_binary_structure(3, 1), border mode 'reflect' == the edge voxel itself)
__global__ void __launch_bounds__(256) morph_cross3d_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                            int d, int h, int w, int op) {
  const int64_t total = (int64_t)d * h * w;
  const int64_t hw = (int64_t)h * w;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int z = (int)(p / hw);
    const int r = (int)(p - z * hw);
    const int y = r / w, x = r - y * w;
    uint32_t v = in[p];
    auto acc = [&](uint32_t u) { v = op ? (u > v ? u : v) : (u < v ? u : v); };
    if (x > 0) acc(in[p - 1]);
    if (x + 1 < w) acc(in[p + 1]);
    if (y > 0) acc(in[p - w]);
    if (y + 1 < h) acc(in[p + w]);
    if (z > 0) acc(in[p - hw]);
    if (z + 1 < d) acc(in[p + hw]);
    out[p] = v;
  }
}

__global__ void __launch_bounds__(256) ccl_flatten_kernel(int* __restrict__ parent_all, int hw) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  int* parent = parent_all + (size_t)n * hw;
  if (parent[p] >= 0) parent[p] = uf_find(parent, p);
}

constexpr int CHUNK = 2048;
// per block: number of roots (parent[p] == p) in its chunk
__global__ void __launch_bounds__(256) ccl_count_kernel(const int* __restrict__ parent_all, int hw,
                                                        uint32_t* __restrict__ blockcnt, int nb) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  const int* parent = parent_all + (size_t)n * hw;
  const int e0 = blockIdx.x * CHUNK + threadIdx.x * 8;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int p = e0 + j;
    if (p < hw && parent[p] == p) ++c;
  }
  uint32_t total;
  block_excl_scan(c, sh, &total);
  if (threadIdx.x == 0) blockcnt[(size_t)n * nb + blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) scan_blocks_kernel(const uint32_t* __restrict__ blockcnt,
                                                          uint32_t* __restrict__ blockoff, int nb,
                                                          int32_t* __restrict__ totals) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.x;
  uint32_t carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    int b = b0 + threadIdx.x;
    uint32_t c = b < nb ? blockcnt[(size_t)n * nb + b] : 0;
    uint32_t t;
    uint32_t e = block_excl_scan(c, sh, &t);
    if (b < nb) blockoff[(size_t)n * nb + b] = carry + e;
    carry += t;
    __syncthreads();
  }
  if (threadIdx.x == 0 && totals) totals[n] = (int32_t)carry;
}
// rank[root] = 1-based raster rank, stored in a side array indexed by root pixel
__global__ void __launch_bounds__(256) ccl_rank_kernel(const int* __restrict__ parent_all, int hw,
                                                       const uint32_t* __restrict__ blockoff, int nb,
                                                       int32_t* __restrict__ rank_all) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  const int* parent = parent_all + (size_t)n * hw;
  int32_t* rank = rank_all + (size_t)n * hw;
  const int e0 = blockIdx.x * CHUNK + threadIdx.x * 8;
  uint32_t c = 0, flags = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int p = e0 + j;
    if (p < hw && parent[p] == p) { ++c; flags |= 1u << j; }
  }
  uint32_t total;
  uint32_t pos = block_excl_scan(c, sh, &total) + blockoff[(size_t)n * nb + blockIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (flags & (1u << j)) rank[e0 + j] = (int32_t)(++pos);
}
__global__ void __launch_bounds__(256) ccl_relabel_kernel(const int* __restrict__ parent_all,
                                                          const int32_t* __restrict__ rank_all,
                                                          int32_t* __restrict__ out, int64_t total, int hw) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = parent_all[i];
    out[i] = r < 0 ? 0 : rank_all[(i / hw) * hw + r];
  }
}

// Engine2d.force_connected (empanada_napari/inference.py:263-279) pieces: isolate one class's id range as the CCL
// input (and, for the first class, narrow the int64 map to int32 on the way), then write cc + min_id back
template <typename T>
__global__ void __launch_bounds__(256) fc_select_kernel(const T* __restrict__ pan, int32_t* __restrict__ out,
                                                        int32_t* __restrict__ sel, int64_t total, int64_t lo,
                                                        int64_t hi, int write_out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = (int64_t)pan[i];
    if (write_out) out[i] = (int32_t)v;
    sel[i] = (v >= lo && v < hi) ? (int32_t)v : 0;
  }
}
__global__ void __launch_bounds__(256) fc_apply_kernel(const int32_t* __restrict__ cc, int32_t* __restrict__ out,
                                                       int64_t total, int32_t lo) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int32_t c = cc[i];
    if (c > 0) out[i] = c + lo;
  }
}

// ---------------------------------------------------------------------------
// run extraction over the raveled image
// flags: bit0 = run start (label != 0 and differs from the previous element),
//        bit1 = run end   (label != 0 and differs from the next element)
// ---------------------------------------------------------------------------
template <class IM>
__device__ __forceinline__ uint32_t run_flags(const IM& im, int hw, int e0, uint32_t* fs, uint32_t* fe) {
  uint32_t s = 0, e = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = e0 + j;
    if (p < hw) {
      const int32_t v = im[p];
      if (v != 0) {
        const int32_t pv = p > 0 ? im[p - 1] : 0;
        const int32_t nv = p + 1 < hw ? im[p + 1] : 0;
        s |= (uint32_t)(pv != v) << j;
        e |= (uint32_t)(nv != v) << j;
      }
    }
  }
  *fs = s;
  *fe = e;
  return (uint32_t)__popc(s) | ((uint32_t)__popc(e) << 16);
}
template <class IM>
__global__ void __launch_bounds__(256) rle_count_kernel(const IM in, int hw,
                                                        uint32_t* __restrict__ blockcnt, int nb) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  uint32_t fs, fe;
  uint32_t c = run_flags(in.at((size_t)n * hw), hw, blockIdx.x * CHUNK + threadIdx.x * 8, &fs, &fe);
  uint32_t total;
  block_excl_scan(c & 0xffffu, sh, &total);  // starts and ends are equinumerous per image, not per block
  if (threadIdx.x == 0) blockcnt[((size_t)n * nb + blockIdx.x) * 2] = total;
  uint32_t t2;
  block_excl_scan(c >> 16, sh, &t2);
  if (threadIdx.x == 0) blockcnt[((size_t)n * nb + blockIdx.x) * 2 + 1] = t2;
}
__global__ void __launch_bounds__(256) rle_scan_kernel(const uint32_t* __restrict__ blockcnt,
                                                       uint32_t* __restrict__ blockoff, int nb,
                                                       int32_t* __restrict__ totals) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.x;
  uint32_t cs = 0, ce = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    int b = b0 + threadIdx.x;
    uint32_t a = b < nb ? blockcnt[((size_t)n * nb + b) * 2] : 0;
    uint32_t c = b < nb ? blockcnt[((size_t)n * nb + b) * 2 + 1] : 0;
    uint32_t ta, tc;
    uint32_t ea = block_excl_scan(a, sh, &ta);
    uint32_t ec = block_excl_scan(c, sh, &tc);
    if (b < nb) {
      blockoff[((size_t)n * nb + b) * 2] = cs + ea;
      blockoff[((size_t)n * nb + b) * 2 + 1] = ce + ec;
    }
    cs += ta;
    ce += tc;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[n] = (int32_t)cs;
}
// runs (max_runs, 3) per image: {start, length, label}; ends are written as start + length later
template <class IM>
__global__ void __launch_bounds__(256) rle_write_kernel(const IM in, int hw,
                                                        const uint32_t* __restrict__ blockoff, int nb,
                                                        int32_t* __restrict__ runs_all, int max_runs) {
  __shared__ uint32_t sh[8];
  const int n = blockIdx.y;
  const IM im = in.at((size_t)n * hw);
  int32_t* runs = runs_all + (size_t)n * max_runs * 3;
  const int e0 = blockIdx.x * CHUNK + threadIdx.x * 8;
  uint32_t fs, fe;
  uint32_t c = run_flags(im, hw, e0, &fs, &fe);
  uint32_t t;
  uint32_t ps = block_excl_scan(c & 0xffffu, sh, &t) + blockoff[((size_t)n * nb + blockIdx.x) * 2];
  uint32_t pe = block_excl_scan(c >> 16, sh, &t) + blockoff[((size_t)n * nb + blockIdx.x) * 2 + 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (fs & (1u << j)) {
      if (ps < (uint32_t)max_runs) { runs[ps * 3] = e0 + j; runs[ps * 3 + 2] = im[e0 + j]; }
      ++ps;
    }
    if (fe & (1u << j)) {
      if (pe < (uint32_t)max_runs) runs[pe * 3 + 1] = e0 + j + 1;  // exclusive end, fixed up below
      ++pe;
    }
  }
}
__global__ void rle_fixup_kernel(int32_t* __restrict__ runs_all, const int32_t* __restrict__ totals, int max_runs) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  int cnt = totals[n];
  cnt = cnt < max_runs ? cnt : max_runs;
  if (i >= cnt) return;
  int32_t* r = runs_all + ((size_t)n * max_runs + i) * 3;
  r[1] -= r[0];
}

// one wave per run
template <typename T>
__global__ void __launch_bounds__(256) rle_fill_kernel(const int64_t* __restrict__ starts,
                                                       const int64_t* __restrict__ lens,
                                                       const int64_t* __restrict__ vals, int64_t nruns,
                                                       T* __restrict__ vol, int64_t size) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t r = wave0; r < nruns; r += nw) {
    const int64_t s = starts[r], e = min(size, s + lens[r]);
    const T v = (T)vals[r];
    for (int64_t p = s + lane; p < e; p += 64) vol[p] = v;
  }
}

// ordered fill, pass 1: every voxel of run r remembers the highest order[r] that covers it
__global__ void __launch_bounds__(256) rle_prio_kernel(const int64_t* __restrict__ starts, const int64_t* __restrict__ lens,
                                                       const int32_t* __restrict__ order, int64_t nruns,
                                                       int32_t* __restrict__ prio, int64_t size) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t r = wave0; r < nruns; r += nw) {
    const int64_t s = starts[r], e = min(size, s + lens[r]);
    const int32_t o = order[r];
    for (int64_t p = s + lane; p < e; p += 64) atomicMax(prio + p, o);
  }
}
// pass 2: only the winning run writes its value
template <typename T>
__global__ void __launch_bounds__(256) rle_fill_ordered_kernel(const int64_t* __restrict__ starts,
                                                               const int64_t* __restrict__ lens,
                                                               const int64_t* __restrict__ vals,
                                                               const int32_t* __restrict__ order, int64_t nruns,
                                                               const int32_t* __restrict__ prio, T* __restrict__ vol,
                                                               int64_t size) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  for (int64_t r = wave0; r < nruns; r += nw) {
    const int64_t s = starts[r], e = min(size, s + lens[r]);
    const T v = (T)vals[r];
    const int32_t o = order[r];
    for (int64_t p = s + lane; p < e; p += 64)
      if (prio[p] == o) vol[p] = v;
  }
}

}  // namespace
}  // namespace emp

using namespace emp;

// depth == 0: N images of H x W, 8-connected; depth > 0: ONE volume depth x H x W, 26-connected
template <class IM>
static int ccl_run(const IM d_in, int N, int depth, int H, int W, int32_t* d_out, int32_t* d_num, void* d_work,
                   hipStream_t s) {
  const int64_t hw64 = (int64_t)(depth > 0 ? depth : 1) * H * W;
  EMP_REQUIRE(hw64 < (1ll << 30), "ccl: image / volume too large (< 2^30 elements)");
  const int hw = (int)hw64;
  const int nb = cdiv(hw, CHUNK);
  int* parent = (int*)d_work;
  int32_t* rank = (int32_t*)(parent + (size_t)N * hw);
  uint32_t* blockcnt = (uint32_t*)(rank + (size_t)N * hw);
  uint32_t* blockoff = blockcnt + (size_t)N * nb;
  const int64_t total = (int64_t)N * hw;
  if (depth > 0)
    hipLaunchKernelGGL(ccl_init_kernel<IM>, dim3(grid_for(total)), dim3(256), 0, s, d_in, parent, total, hw);
  else
    hipLaunchKernelGGL(ccl_init_rows_kernel<IM>, dim3(cdiv(hw, 256), N), dim3(256), 0, s, d_in, parent, hw, W);
  EMP_LAUNCH_CHECK();
  if (depth > 0)
    hipLaunchKernelGGL(ccl_merge3d_kernel<IM>, dim3(cdiv(hw, 256)), dim3(256), 0, s, d_in, parent, depth, H, W);
  else
    hipLaunchKernelGGL(ccl_merge_kernel<IM>, dim3(cdiv(hw, 256), N), dim3(256), 0, s, d_in, parent, H, W);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(cdiv(hw, 256), N), dim3(256), 0, s, parent, hw);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ccl_count_kernel, dim3(nb, N), dim3(256), 0, s, parent, hw, blockcnt, nb);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(N), dim3(256), 0, s, blockcnt, blockoff, nb, d_num);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ccl_rank_kernel, dim3(nb, N), dim3(256), 0, s, parent, hw, blockoff, nb, rank);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(ccl_relabel_kernel, dim3(grid_for(total)), dim3(256), 0, s, parent, rank, d_out, total, hw);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <class IM>
static int rle_extract_run(const IM d_labels, int N, int H, int W, int32_t* d_runs, int32_t* d_num_runs, int max_runs,
                           void* d_work, hipStream_t s) {
  EMP_REQUIRE(d_labels.p && d_runs && d_num_runs && d_work && N > 0 && H > 0 && W > 0 && max_runs > 0,
              "rle_extract: bad arguments");
  EMP_REQUIRE((int64_t)H * W < (1ll << 30), "rle_extract: image too large");
  const int hw = H * W;
  const int nb = cdiv(hw, CHUNK);
  uint32_t* blockcnt = (uint32_t*)d_work;
  uint32_t* blockoff = blockcnt + (size_t)N * nb * 2;
  hipLaunchKernelGGL(rle_count_kernel<IM>, dim3(nb, N), dim3(256), 0, s, d_labels, hw, blockcnt, nb);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(rle_scan_kernel, dim3(N), dim3(256), 0, s, blockcnt, blockoff, nb, d_num_runs);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(rle_write_kernel<IM>, dim3(nb, N), dim3(256), 0, s, d_labels, hw, blockoff, nb, d_runs, max_runs);
  EMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(rle_fixup_kernel, dim3(cdiv(max_runs, 256), N), dim3(256), 0, s, d_runs, d_num_runs, max_runs);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

extern "C" {

size_t emp_ccl8_work_bytes(int N, int H, int W) {
  const size_t hw = (size_t)H * W;
  const size_t nb = (hw + CHUNK - 1) / CHUNK;
  return (size_t)N * (hw * 8 + nb * 8) + 512;
}


int emp_ccl8(const int32_t* d_in, int N, int H, int W, int32_t* d_out, int32_t* d_num, void* d_work, void* stream) {
  EMP_REQUIRE(d_in && d_out && d_work && N > 0 && H > 0 && W > 0, "ccl8: bad arguments");
  return ccl_run(ImgPlain{d_in, 0, 0}, N, 0, H, W, d_out, d_num, d_work, (hipStream_t)stream);
}

// emp_ccl8 / emp_ccl26 of the labels of [lo, hi) of an int32 (in_bytes 4) or int64 (8) map, the rest read as background: the
// class isolation of rle.py:46-48 folded into the kernels' reads (depth 0: N images, 8-connected; depth > 0: one volume)
int emp_ccl_range(const void* d_in, int in_bytes, int N, int depth, int H, int W, int64_t lo, int64_t hi, int32_t* d_out, int32_t* d_num,
                  void* d_work, void* stream) {
  EMP_REQUIRE(d_in && d_out && d_work && N > 0 && H > 0 && W > 0 && depth >= 0 && (depth == 0 || N == 1), "ccl_range: bad arguments");
  EMP_REQUIRE((in_bytes == 4 || in_bytes == 8) && lo >= 0 && hi > lo && hi <= (1ll << 31) - 1, "ccl_range: 4- or 8-byte labels, 0 <= lo < hi < 2^31");
  if (in_bytes == 8) return ccl_run(Img<int64_t, true>{(const int64_t*)d_in, lo, hi}, N, depth, H, W, d_out, d_num, d_work, (hipStream_t)stream);
  return ccl_run(Img<int32_t, true>{(const int32_t*)d_in, lo, hi}, N, depth, H, W, d_out, d_num, d_work, (hipStream_t)stream);
}

size_t emp_force_connected_work_bytes(int N, int H, int W) {
  return emp_ccl8_work_bytes(N, H, W) + (size_t)N * H * W * 4 + 256;
}

int emp_force_connected(const int64_t* d_pan, int N, int H, int W, const int32_t* h_thing_list, int n_things,
                        int64_t label_divisor, int32_t* d_out, void* d_work, void* stream) {
  EMP_REQUIRE(d_pan && d_out && d_work && N > 0 && H > 0 && W > 0 && n_things >= 0 && label_divisor > 0,
              "force_connected: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)N * H * W;
  int32_t* sel = (int32_t*)d_work;
  void* ccl_work = (char*)d_work + (((size_t)total * 4 + 255) & ~(size_t)255);
  if (n_things == 0) {   // nothing to relabel: the narrowing alone
    hipLaunchKernelGGL(fc_select_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, s, d_pan, d_out, sel, total,
                       (int64_t)1, (int64_t)0, 1);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  }
  for (int t = 0; t < n_things; ++t) {
    const int64_t lo = (int64_t)h_thing_list[t] * label_divisor, hi = lo + label_divisor;
    EMP_REQUIRE(hi < (1ll << 31), "force_connected: class id range beyond int32");
    // later classes read the map the earlier ones have relabelled (the reference edits pan_seg in place)
    if (t == 0)
      hipLaunchKernelGGL(fc_select_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, s, d_pan, d_out, sel, total, lo, hi, 1);
    else
      hipLaunchKernelGGL(fc_select_kernel<int32_t>, dim3(grid_for(total)), dim3(256), 0, s, (const int32_t*)d_out, d_out, sel,
                         total, lo, hi, 0);
    EMP_LAUNCH_CHECK();
    const int rc = ccl_run(ImgPlain{sel, 0, 0}, N, 0, H, W, sel, nullptr, ccl_work, s);   // relabel does not read its input: in place
    if (rc) return rc;
    hipLaunchKernelGGL(fc_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, sel, d_out, total, (int32_t)lo);
    EMP_LAUNCH_CHECK();
  }
  return EMP_OK;
}

// work buffer: emp_ccl8_work_bytes(1, D * H, W)
int emp_ccl26(const int32_t* d_in, int D, int H, int W, int32_t* d_out, int32_t* d_num, void* d_work, void* stream) {
  EMP_REQUIRE(d_in && d_out && d_work && D > 0 && H > 0 && W > 0, "ccl26: bad arguments");
  return ccl_run(ImgPlain{d_in, 0, 0}, 1, D, H, W, d_out, d_num, d_work, (hipStream_t)stream);
}

int emp_morph_cross3d(const uint32_t* d_in, uint32_t* d_out, int D, int H, int W, int op, void* stream) {
  EMP_REQUIRE(d_in && d_out && d_in != d_out && D > 0 && H > 0 && W > 0 && (op == 0 || op == 1), "morph_cross3d: bad arguments");
  hipLaunchKernelGGL(morph_cross3d_kernel, dim3(grid_for((int64_t)D * H * W)), dim3(256), 0, (hipStream_t)stream, d_in,
                     d_out, D, H, W, op);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

size_t emp_rle_extract_work_bytes(int N, int H, int W) {
  const size_t nb = ((size_t)H * W + CHUNK - 1) / CHUNK;
  return (size_t)N * nb * 16 + 512;
}


int emp_rle_extract(const int32_t* d_labels, int N, int H, int W, int32_t* d_runs, int32_t* d_num_runs, int max_runs,
                    void* d_work, void* stream) {
  return rle_extract_run(ImgPlain{d_labels, 0, 0}, N, H, W, d_runs, d_num_runs, max_runs, d_work, (hipStream_t)stream);
}

// emp_rle_extract of the labels of [lo, hi) of an int32 / int64 map (the rest is background): rle.py:46-48's isolation at the read
int emp_rle_extract_range(const void* d_labels, int in_bytes, int N, int H, int W, int64_t lo, int64_t hi, int32_t* d_runs,
                          int32_t* d_num_runs, int max_runs, void* d_work, void* stream) {
  EMP_REQUIRE((in_bytes == 4 || in_bytes == 8) && lo >= 0 && hi > lo && hi <= (1ll << 31) - 1, "rle_extract_range: 4- or 8-byte labels, 0 <= lo < hi < 2^31");
  if (in_bytes == 8)
    return rle_extract_run(Img<int64_t, true>{(const int64_t*)d_labels, lo, hi}, N, H, W, d_runs, d_num_runs, max_runs, d_work, (hipStream_t)stream);
  return rle_extract_run(Img<int32_t, true>{(const int32_t*)d_labels, lo, hi}, N, H, W, d_runs, d_num_runs, max_runs, d_work, (hipStream_t)stream);
}

int emp_rle_fill(const int64_t* d_starts, const int64_t* d_lens, const int64_t* d_vals, int64_t nruns, void* d_volume,
                 int64_t size, int elem_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  EMP_REQUIRE(d_volume && nruns >= 0 && size > 0, "rle_fill: bad arguments");
  if (nruns == 0) return EMP_OK;
  const int grid = grid_for(nruns * 64, 256, 256 * 32);
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(rle_fill_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, nruns, (uint8_t*)d_volume, size); break;
    case 2: hipLaunchKernelGGL(rle_fill_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, nruns, (uint16_t*)d_volume, size); break;
    case 4: hipLaunchKernelGGL(rle_fill_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, nruns, (uint32_t*)d_volume, size); break;
    case 8: hipLaunchKernelGGL(rle_fill_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, nruns, (uint64_t*)d_volume, size); break;
    default: EMP_REQUIRE(false, "rle_fill: element size %d unsupported", elem_bytes);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// Same with the reference's overwrite rule for overlapping objects (numpy_fill_instances fills the instances one after
// the other, array_utils.py:754-766): where runs overlap the one with the highest d_order wins.  d_prio is a scratch
// volume of `size` int32 that this call initialises.  Runs of equal order must not overlap each other.
int emp_rle_fill_ordered(const int64_t* d_starts, const int64_t* d_lens, const int64_t* d_vals, const int32_t* d_order,
                         int64_t nruns, void* d_volume, int64_t size, int elem_bytes, int32_t* d_prio, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  EMP_REQUIRE(d_volume && d_prio && d_order && nruns >= 0 && size > 0, "rle_fill_ordered: bad arguments");
  if (nruns == 0) return EMP_OK;
  EMP_CHECK_HIP(hipMemsetAsync(d_prio, 0xff, (size_t)size * sizeof(int32_t), s));     // -1
  const int grid = grid_for(nruns * 64, 256, 256 * 32);
  hipLaunchKernelGGL(rle_prio_kernel, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_order, nruns, d_prio, size);
  EMP_LAUNCH_CHECK();
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(rle_fill_ordered_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, d_order, nruns, d_prio, (uint8_t*)d_volume, size); break;
    case 2: hipLaunchKernelGGL(rle_fill_ordered_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, d_order, nruns, d_prio, (uint16_t*)d_volume, size); break;
    case 4: hipLaunchKernelGGL(rle_fill_ordered_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, d_order, nruns, d_prio, (uint32_t*)d_volume, size); break;
    case 8: hipLaunchKernelGGL(rle_fill_ordered_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, d_starts, d_lens, d_vals, d_order, nruns, d_prio, (uint64_t*)d_volume, size); break;
    default: EMP_REQUIRE(false, "rle_fill_ordered: element size %d unsupported", elem_bytes);
  }
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// ---------------------------------------------------------------------------
// host range algebra
// ---------------------------------------------------------------------------
int emp_rle_pair_intersections(const int64_t* h_starts, const int64_t* h_runs, const int64_t* h_off,
                               const int64_t* h_pairs, int64_t n_pairs, int64_t* h_out) {
  EMP_REQUIRE(h_starts && h_runs && h_off && h_pairs && h_out && n_pairs >= 0, "rle_pair_intersections: null argument");
  auto one = [&](int64_t k) {
    const int64_t a = h_pairs[2 * k], b = h_pairs[2 * k + 1];
    int64_t i = h_off[a], ie = h_off[a + 1], j = h_off[b], je = h_off[b + 1];
    int64_t tot = 0;
    while (i < ie && j < je) {
      const int64_t s1 = h_starts[i], e1 = s1 + h_runs[i], s2 = h_starts[j], e2 = s2 + h_runs[j];
      const int64_t lo = s1 > s2 ? s1 : s2, hi = e1 < e2 ? e1 : e2;
      if (hi > lo) tot += hi - lo;
      if (e1 < e2) ++i; else ++j;
    }
    h_out[k] = tot;
  };
  // pairs are independent two-pointer merges: big jobs (consensus of whole-volume trackers) are spread over threads
  int64_t work = 0;
  for (int64_t k = 0; k < n_pairs; ++k)
    work += (h_off[h_pairs[2 * k] + 1] - h_off[h_pairs[2 * k]]) + (h_off[h_pairs[2 * k + 1] + 1] - h_off[h_pairs[2 * k + 1]]);
  unsigned nthr = std::thread::hardware_concurrency();
  nthr = nthr > 16 ? 16 : nthr;
  if ((int64_t)nthr > n_pairs) nthr = (unsigned)n_pairs;
  if (work < (1 << 18) || nthr < 2) {
    for (int64_t k = 0; k < n_pairs; ++k) one(k);
    return EMP_OK;
  }
  std::atomic<int64_t> next{0};
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthr; ++t)
    pool.emplace_back([&] {
      for (int64_t k = next.fetch_add(1); k < n_pairs; k = next.fetch_add(1)) one(k);
    });
  for (auto& th : pool) th.join();
  return EMP_OK;
}

// ranges: (n,2) [start,end) in any order; out: maximal ranges whose every index is covered by >= thr
// input ranges (touching ranges chain, like the reference's running range).  Returns the number of
// output ranges through *n_out; h_out must hold n ranges.
int emp_ranges_vote(const int64_t* h_ranges, int64_t n, int thr, int64_t* h_out, int64_t* n_out) {
  EMP_REQUIRE(h_ranges && h_out && n_out && n >= 0 && thr >= 1, "ranges_vote: bad arguments");
  // starts and ends are swept as two sorted sequences (the coverage right of a coordinate does not depend on the order
  // of the events AT that coordinate).  The callers concatenate a few already sorted run lists, so the sort is a
  // merge of the ascending stretches found in the input; arbitrary input falls back to std::sort.
  std::vector<int64_t> st, en;
  st.reserve((size_t)n);
  en.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (h_ranges[2 * i + 1] <= h_ranges[2 * i]) continue;
    st.push_back(h_ranges[2 * i]);
    en.push_back(h_ranges[2 * i + 1]);
  }
  // many stretches (a tracker appends its runs slice by slice in descending slice order): LSD radix sort, 11 bits per
  // pass over the significant bits of the non-negative keys (3 passes for a 512^3 volume) instead of a comparison sort
  auto radix_sort = [](std::vector<int64_t>& v) {
    int64_t mx = 0;
    for (int64_t x : v) { if (x < 0) { std::sort(v.begin(), v.end()); return; } mx = x > mx ? x : mx; }
    std::vector<int64_t> tmp(v.size());
    int64_t* a = v.data();
    int64_t* b = tmp.data();
    for (int shift = 0; shift < 63 && (mx >> shift) != 0; shift += 11) {
      size_t cnt[2049] = {0};
      for (size_t i = 0; i < v.size(); ++i) ++cnt[((a[i] >> shift) & 2047) + 1];
      for (int k = 0; k < 2048; ++k) cnt[k + 1] += cnt[k];
      for (size_t i = 0; i < v.size(); ++i) b[cnt[(a[i] >> shift) & 2047]++] = a[i];
      std::swap(a, b);
    }
    if (a != v.data()) std::memcpy(v.data(), a, v.size() * sizeof(int64_t));
  };
  auto natural_sort = [&](std::vector<int64_t>& v) {
    std::vector<size_t> cut{0};
    for (size_t i = 1; i < v.size(); ++i)
      if (v[i] < v[i - 1]) {
        cut.push_back(i);
        if (cut.size() > 64) { radix_sort(v); return; }
      }
    cut.push_back(v.size());
    while (cut.size() > 2) {              // merge neighbouring stretches pairwise
      std::vector<size_t> next{0};
      for (size_t k = 0; k + 2 < cut.size(); k += 2) {
        std::inplace_merge(v.begin() + cut[k], v.begin() + cut[k + 1], v.begin() + cut[k + 2]);
        next.push_back(cut[k + 2]);
      }
      if (next.back() != v.size()) next.push_back(v.size());
      cut.swap(next);
    }
  };
  natural_sort(st);
  natural_sort(en);
  int64_t cnt = 0, m = 0, open_at = 0;
  bool open = false;
  size_t i = 0, j = 0;
  const size_t N = st.size();
  while (j < N) {                         // every start is followed by its end, so the ends run out last
    const int64_t pos = (i < N && st[i] <= en[j]) ? st[i] : en[j];
    while (i < N && st[i] == pos) { ++cnt; ++i; }
    while (j < N && en[j] == pos) { --cnt; ++j; }
    const bool ok = cnt >= thr;           // net coverage just right of pos: touching ranges stay chained
    if (ok && !open) { open = true; open_at = pos; }
    else if (!ok && open) {
      open = false;
      h_out[2 * m] = open_at;
      h_out[2 * m + 1] = pos;
      ++m;
    }
  }
  *n_out = m;
  return EMP_OK;
}

// fill_holes_in_segmentation (filters.py:178-210) on a host label volume, slice by slice (slices are independent and
// run on threads).  Bug-compatible with the reference loop: labels of the slice in ascending order (regionprops), each
// with the bounding box it had BEFORE the loop; the box is cut out of the CURRENT slice, every non-zero pixel and every
// hole of the cut-out (scipy.ndimage.binary_fill_holes: background not 4-connected to the cut-out's border) becomes
// that label -- including pixels of other objects inside the box.
int emp_fill_holes_slices(uint32_t* h_vol, int64_t D, int64_t H, int64_t W) {
  EMP_REQUIRE(h_vol && D > 0 && H > 0 && W > 0, "fill_holes_slices: bad arguments");
  auto one_slice = [&](int64_t z) {
    uint32_t* m = h_vol + z * H * W;
    std::vector<uint32_t> labels;
    {
      uint32_t last = 0;
      for (int64_t i = 0; i < H * W; ++i) {
        const uint32_t v = m[i];
        if (v != 0 && v != last) { labels.push_back(v); last = v; }
      }
      std::sort(labels.begin(), labels.end());
      labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
    }
    if (labels.empty()) return;
    const size_t nl = labels.size();
    std::vector<int64_t> box(nl * 4);
    for (size_t k = 0; k < nl; ++k) { box[4 * k] = H; box[4 * k + 1] = W; box[4 * k + 2] = -1; box[4 * k + 3] = -1; }
    for (int64_t y = 0; y < H; ++y)
      for (int64_t x = 0; x < W; ++x) {
        const uint32_t v = m[y * W + x];
        if (v == 0) continue;
        const size_t k = std::lower_bound(labels.begin(), labels.end(), v) - labels.begin();
        int64_t* b = &box[4 * k];
        if (y < b[0]) b[0] = y;
        if (x < b[1]) b[1] = x;
        if (y > b[2]) b[2] = y;
        if (x > b[3]) b[3] = x;
      }
    std::vector<uint8_t> outside;
    std::vector<int64_t> stack;
    for (size_t k = 0; k < nl; ++k) {
      const int64_t y0 = box[4 * k], x0 = box[4 * k + 1], bh = box[4 * k + 2] - y0 + 1, bw = box[4 * k + 3] - x0 + 1;
      outside.assign((size_t)(bh * bw), 0);
      stack.clear();
      auto push = [&](int64_t yy, int64_t xx) {
        if (m[(y0 + yy) * W + x0 + xx] == 0 && !outside[(size_t)(yy * bw + xx)]) {
          outside[(size_t)(yy * bw + xx)] = 1;
          stack.push_back(yy * bw + xx);
        }
      };
      for (int64_t xx = 0; xx < bw; ++xx) { push(0, xx); push(bh - 1, xx); }
      for (int64_t yy = 0; yy < bh; ++yy) { push(yy, 0); push(yy, bw - 1); }
      while (!stack.empty()) {
        const int64_t q = stack.back();
        stack.pop_back();
        const int64_t yy = q / bw, xx = q - yy * bw;
        if (yy > 0) push(yy - 1, xx);
        if (yy + 1 < bh) push(yy + 1, xx);
        if (xx > 0) push(yy, xx - 1);
        if (xx + 1 < bw) push(yy, xx + 1);
      }
      const uint32_t L = labels[k];
      for (int64_t yy = 0; yy < bh; ++yy)
        for (int64_t xx = 0; xx < bw; ++xx)
          m[(y0 + yy) * W + x0 + xx] = outside[(size_t)(yy * bw + xx)] ? 0u : L;
    }
  };
  unsigned nthr = std::thread::hardware_concurrency();
  nthr = nthr > 32 ? 32 : (nthr < 1 ? 1 : nthr);
  if ((int64_t)nthr > D) nthr = (unsigned)D;
  if (nthr < 2) {
    for (int64_t z = 0; z < D; ++z) one_slice(z);
    return EMP_OK;
  }
  std::atomic<int64_t> next{0};
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthr; ++t)
    pool.emplace_back([&] { for (int64_t z = next.fetch_add(1); z < D; z = next.fetch_add(1)) one_slice(z); });
  for (auto& th : pool) th.join();
  return EMP_OK;
}

}  // extern "C"
