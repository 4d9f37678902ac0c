// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet layer1 conv2) for gfx950.
//
// As an implicit GEMM with one K-step per tap (conv_igemm.hip) this layer re-stages its 128-pixel x 64-channel A tile
// nine times per output tile and feeds only 64 couts from it: 24 KB of operands per MFLOP through the vector-memory
// path, 0.53 PFLOP/s.  Here the operands move once:
//   * the whole weight tensor (64 x 9 x 64 fp16 = 72 KiB) lives in REGISTERS for the lifetime of a persistent
//     workgroup: wave (pg, ch) owns couts ch*32..+31 = 2 MFMA row tiles x 18 K-steps = 36 fragments = 144 VGPRs;
//   * an 8 x 16 output tile reads its 10 x 18 input halo (180 pixels x 128 B) from LDS, where LDS-DMA put it two tiles
//     ahead (ring of three, 23 KB each); the nine taps are nine row offsets into the same halo tile.
// 8 waves = 4 pixel groups (two output rows each) x 2 cout halves; per tile and wave 36 ds_read_b128 + 72 MFMAs, one
// LDS-only barrier, two 16-byte stores per lane.  K order (tap-major, two 32-channel steps per tap, ascending) and the
// fp32 epilogue equal conv_igemm_kernel's, so results are bit-identical.
#include "common.h"

namespace emp {
namespace {

constexpr int C64_TH = 8, C64_TW = 16;                      // output tile
constexpr int C64_IH = C64_TH + 2, C64_IW = C64_TW + 2;     // halo 10 x 18
constexpr int C64_NPIX = C64_IH * C64_IW;                   // 180
constexpr int C64_NPC = (C64_NPIX + 7) / 8;                 // 23 DMA pieces of 8 pixels
constexpr int C64_SLOT = (C64_NPC + 1) * 1024;              // 24 KB: 23 pieces + one dump piece for the 24th DMA
constexpr int C64_NDMA = 3;                                 // pieces per wave and tile (8 waves x 3 >= 23)

struct C64Params {
  const half_t* in;
  const half_t* wgt;     // [64][9][64]
  const float* bias;
  half_t* out;
  const half_t* zero;
  int N, H, W, in_ld, out_ld, act;
  int tiles_x, tiles_y, tiles;
};

__device__ __forceinline__ int c64_perm32(int x) {
  const int t = x >> 4, i = x & 15;
  return ((i >> 2) << 3) + (t << 2) + (i & 3);
}
template <int ACT>
__device__ __forceinline__ float c64_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}
__device__ __forceinline__ void c64_barrier() {   // LDS-only: the LDS-DMA of younger tiles stays in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int ACT>
__global__ void __launch_bounds__(512, 1) conv3x3_c64_kernel(const C64Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];     // 3 x C64_SLOT
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 1, ch = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // XCD-aware tile order: at iteration `it` XCD x owns tiles [(it*8+x)*nx, +nx)
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
  auto tile_of = [&](int it) { return (it * 8 + xcd) * nx + jx; };
  int my_tiles = 0;
  while (tile_of(my_tiles) < p.tiles) ++my_tiles;

  // ---- this wave's weights: cout row of MFMA row fr in tile c = ch*32 + perm32(c*16 + fr) ----
  f16x8 wreg[2][9][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const half_t* wr = p.wgt + (size_t)(ch * 32 + c64_perm32(c * 16 + fr)) * (9 * 64) + fq * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wreg[c][t][kk] = *reinterpret_cast<const f16x8*>(wr + t * 64 + kk * 32);
  }
  float bv[8];
  {
    const int co = ch * 32 + fq * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[r] = p.bias ? p.bias[co + r] : 0.f;
  }
  // the weight / bias loads must not be pending when the counted vmcnt waits start
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- halo staging: wave w moves pieces {w, w+8, w+16}; every wave issues three DMAs per tile (constant vmcnt), the
  //      24th lands zeros in a dump block behind the 23 real pieces ----
  auto stage = [&](int it, int slot) {
    const bool valid = it < my_tiles;
    int n = 0, y0 = 0, x0 = 0;
    if (valid) {
      const int tile = tile_of(it);
      const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
      const int ty = r % p.tiles_y;
      n = r / p.tiles_y;
      y0 = ty * C64_TH;
      x0 = tx * C64_TW;
    }
    char* hb = lds + slot * C64_SLOT;
#pragma unroll
    for (int k = 0; k < C64_NDMA; ++k) {
      const int piece = wave + 8 * k;                           // 0..23; piece 23 is the dump row block (zeros)
      const int hp = piece * 8 + (lane >> 3);                   // halo pixel of this lane's LDS row
      const int hy = hp / C64_IW, hx = hp - hy * C64_IW;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = valid && wave + 8 * k < C64_NPC && hp < C64_NPIX && (unsigned)iy < (unsigned)p.H &&
                      (unsigned)ix < (unsigned)p.W;
      const half_t* src = ok ? p.in + (((size_t)n * p.H + iy) * p.W + ix) * p.in_ld + (((lane & 7) ^ (hp & 7)) << 3)
                             : p.zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(hb + piece * 1024), 16, 0, 0);
    }
  };

  // fragment base offsets of this wave's two output rows (pixel tile t = output row 2*pg + t), tap (0,0), kk = 0
  stage(0, 0);
  stage(1, 1);
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // tile 0 landed (tile 1 may be in flight)
  c64_barrier();

  int slot = 0;
  for (int it = 0; it < my_tiles; ++it) {
    stage(it + 2, slot == 0 ? 2 : slot - 1);         // (slot + 2) % 3
    const char* hb = lds + slot * C64_SLOT;
    f32x4 acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 pf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int hp = (2 * pg + t + dy) * C64_IW + fr + dx;
          pf[t] = *reinterpret_cast<const f16x8*>(hb + hp * 128 + (((kk * 4 + fq) ^ (hp & 7)) << 4));
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[c][tap][kk], pf[t], acc[c][t], 0, 0, 0);
      }
    }
    // VM issue order so far: [DMA(it+1), one tile ago] [stores(it-1)] [DMA(it+2), this tile].  Loads retire in order,
    // so vmcnt(3) == "the halo of tile it+1 has landed" (the previous tile's stores had a whole tile to complete)
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    // ---- epilogue: lane (fq, fr) owns couts ch*32 + fq*8 + [0,8) of pixel fr of each of its two rows ----
    {
      const int tile = tile_of(it);
      const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
      const int ty = r % p.tiles_y, n = r / p.tiles_y;
      const int ox = tx * C64_TW + fr;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int oy = ty * C64_TH + 2 * pg + t;
        f16x8 o;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          o[r4] = (half_t)c64_act<ACT>(acc[0][t][r4] + bv[r4]);
          o[4 + r4] = (half_t)c64_act<ACT>(acc[1][t][r4] + bv[4 + r4]);
        }
        if (oy < p.H && ox < p.W)
          *reinterpret_cast<f16x8*>(p.out + (((size_t)n * p.H + oy) * p.W + ox) * p.out_ld + ch * 32 + fq * 8) = o;
      }
    }
    c64_barrier();     // tile it+1 visible to every wave; slot `slot` free for the DMA of tile it+3
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the dummy DMA batches
}

// ---------------------------------------------------------------------------
// 128 -> 128 channels (ResNet layer2 conv2, stride 1): same scheme with one MFMA row tile per wave -- wave w keeps
// couts 16w..16w+15 x 1152 K = 36 fragments = 144 VGPRs and multiplies them with ALL eight pixel rows of the tile.
// A halo row's fragment (120 ds_read_b128 per tile and wave) feeds the up to three output rows it belongs to, in an
// order that keeps every output's K walk tap-major / channel-ascending (bit-identical to conv_igemm_kernel).
// Halo rows are 256 B = 16 chunks of 16 B, chunk c of halo pixel hp at chunk c ^ (hp & 15) (round 5: the XOR was 3 bits wide, so the
// fq = 0 and fq = 1 lanes of a ds_read_b128 group shared eight chunk positions -- 49 % of the kernel's LDS cycles were bank conflicts);
// 4 pixels per LDS-DMA instruction, 45 pieces + 3 dump pieces = 6 per wave; a wave's 16 couts
// are 32 B per pixel, so the finished tile is transposed through the just-consumed halo slot into whole 256-byte rows.
// ---------------------------------------------------------------------------
constexpr int C128_NDMA = 6;                                // per wave and tile: 48 >= 45
constexpr int C128_SLOT = 8 * C128_NDMA * 1024;             // 48 KB incl. 3 dump pieces

template <int ACT>
__global__ void __launch_bounds__(512, 1) conv3x3_c128_kernel(const C64Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];     // 3 x C128_SLOT
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
  auto tile_of = [&](int it) { return (it * 8 + xcd) * nx + jx; };
  int my_tiles = 0;
  while (tile_of(my_tiles) < p.tiles) ++my_tiles;

  f16x8 wreg[9][4];
  {
    const half_t* wr = p.wgt + (size_t)(16 * wave + fr) * (9 * 128) + fq * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wreg[t][kk] = *reinterpret_cast<const f16x8*>(wr + t * 128 + kk * 32);
  }
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[16 * wave + fq * 4 + r] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto tile_xy = [&](int it, int& n, int& y0, int& x0) {
    const int tile = tile_of(it);
    const int tx = tile % p.tiles_x, r = tile / p.tiles_x;
    const int ty = r % p.tiles_y;
    n = r / p.tiles_y;
    y0 = ty * C64_TH;
    x0 = tx * C64_TW;
  };
  auto stage = [&](int it, int slot) {
    const bool valid = it < my_tiles;
    int n = 0, y0 = 0, x0 = 0;
    if (valid) tile_xy(it, n, y0, x0);
    char* hb = lds + slot * C128_SLOT;
#pragma unroll
    for (int k = 0; k < C128_NDMA; ++k) {
      const int piece = wave + 8 * k;                           // 0..47; pieces 45..47 are dump blocks
      const int hp = piece * 4 + (lane >> 4);
      const int hy = hp / C64_IW, hx = hp - hy * C64_IW;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = valid && hp < C64_NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const half_t* src = ok ? p.in + (((size_t)n * p.H + iy) * p.W + ix) * p.in_ld + (((lane & 15) ^ (hp & 15)) << 3)
                             : p.zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(hb + piece * 1024), 16, 0, 0);
    }
  };

  stage(0, 0);
  stage(1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  c64_barrier();

  int slot = 0;
  for (int it = 0; it < my_tiles; ++it) {
    stage(it + 2, slot == 0 ? 2 : slot - 1);
    char* hb = lds + slot * C128_SLOT;
    f32x4 acc[8];
#pragma unroll
    for (int y = 0; y < 8; ++y) acc[y] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 30 groups (halo row r, column shift dx) of four 32-channel fragments, each feeding up to twelve MFMAs; the scheduler
    // must not hoist the reads of later groups (144 VGPRs hold the weights) -- the SIMD's second wave covers their latency
    int frv = fr;                      // laundered per tile: the 120 swizzled fragment offsets are recomputed next to
    asm volatile("" : "+v"(frv));      // their reads instead of being hoisted out of the tile loop into 120 VGPRs
#pragma unroll
    for (int g = 0; g < 3 * C64_IH; ++g) {
      const int r = g / 3, dx = g - r * 3;
      const int hp = r * C64_IW + frv + dx;
      f16x8 pf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        pf[kk] = *reinterpret_cast<const f16x8*>(hb + hp * 256 + (((kk * 4 + fq) ^ (hp & 15)) << 4));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int y = r - dy;
          if (y >= 0 && y < 8) acc[y] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[dy * 3 + dx][kk], pf[kk], acc[y], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    // [DMA(it+1) x6, one tile ago] [stores(it-1)] [DMA(it+2) x6]: loads retire in order -> tile it+1 has landed
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    c64_barrier();                 // every wave is done with the halo of tile `it`: the slot becomes the output tile
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const int px = y * 16 + fr;
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      f16x4 o;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) o[r4] = (half_t)c64_act<ACT>(acc[y][r4] + bv[r4]);
      const int chunk = 2 * wave + (fq >> 1);                   // 16-byte chunk of couts 16w + 4fq .. +3
      *reinterpret_cast<f16x4*>(hb + px * 256 + ((chunk ^ (px & 15)) << 4) + (fq & 1) * 8) = o;
    }
    c64_barrier();
    {
      int n, y0, x0;
      tile_xy(it, n, y0, x0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = k * 512 + tid;                            // 2048 chunks: pixel q >> 4, chunk q & 15
        const int px = q >> 4, c = q & 15;
        const int oy = y0 + (px >> 4), ox = x0 + (px & 15);
        const f16x8 v = *reinterpret_cast<const f16x8*>(hb + px * 256 + ((c ^ (px & 15)) << 4));
        if (oy < p.H && ox < p.W)
          *reinterpret_cast<f16x8*>(p.out + (((size_t)n * p.H + oy) * p.W + ox) * p.out_ld + c * 8) = v;
      }
    }
    c64_barrier();                 // the slot is free for the DMA of tile it+3; tile it+1 is visible
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

bool conv3x3_c64_supported(const ConvParams& p) {
  return ((p.Cin == 64 && p.Cout == 64) || (p.Cin == 128 && p.Cout == 128)) && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.dil == 1 &&
         p.res == nullptr && p.bias_n == nullptr && p.ps_cout == 0 && p.in2 == nullptr && p.out2 == nullptr;
}

int launch_conv3x3_c64(const ConvParams& q, hipStream_t stream) {
  EMP_REQUIRE(conv3x3_c64_supported(q), "conv3x3_c64: unsupported shape");
  C64Params p{};
  p.in = q.in; p.wgt = q.wgt; p.bias = q.bias; p.out = q.out; p.zero = q.zero;
  p.N = q.N; p.H = q.H; p.W = q.W; p.in_ld = q.in_ld; p.out_ld = q.out_ld; p.act = q.act;
  p.tiles_x = cdiv(q.W, C64_TW);
  p.tiles_y = cdiv(q.H, C64_TH);
  p.tiles = q.N * p.tiles_x * p.tiles_y;
  int grid = 256;
  while (grid > 8 && grid > p.tiles) grid -= 8;
  auto go = [&](auto kern) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 3 * C64_SLOT)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 3 * C64_SLOT, stream, p);
    return EMP_OK;
  };
  auto go128 = [&](auto kern) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 3 * C128_SLOT)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 3 * C128_SLOT, stream, p);
    return EMP_OK;
  };
  int rc;
  if (q.Cin == 128) {
    if (q.act == 1) rc = go128(&conv3x3_c128_kernel<1>);
    else if (q.act == 2) rc = go128(&conv3x3_c128_kernel<2>);
    else rc = go128(&conv3x3_c128_kernel<0>);
  } else if (q.act == 1) rc = go(&conv3x3_c64_kernel<1>);
  else if (q.act == 2) rc = go(&conv3x3_c64_kernel<2>);
  else rc = go(&conv3x3_c64_kernel<0>);
  if (rc) return rc;
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

}  // namespace emp
