// Fused depthwise-separable block of the fp16x3 mode (round 6):
//   y = act( pw( dwKxK(x) ) + bias )            [+ optional 1x1 head on y]
// replaces nn.Conv2d(C,C,K,groups=C,pad=K/2,bias=False) -> nn.Conv2d(C,Cout,1) -> folded BN -> ReLU / SiLU
// (models/blocks.py:15-33; called from models/decoders/panoptic_deeplab.py:68-80, models/heads.py:12-15 and bifpn.py) on the
// fp32 graph's maps: fp32 NHWC in, fp32 NHWC out (or the head's fp32 planes).  Round 5 ran the block as a depthwise kernel
// (dwconv32_strip) that writes an M x C fp32 map and a pointwise conv16x3 launch that reads it back: 1.3 GB written and re-read
// per block at batch 8, 17 % of the compliant step.  Here the depthwise output is only ever the B operand of the pointwise
// product: it is produced straight into LDS, already split into its fp16 hi / lo pair.
//
// The role split of sepconv_precise.hip (the fp16 engine's exact block), re-cut for fp32 maps: one persistent workgroup per CU,
// eight waves, one "dw" and one "mma" wave per SIMD.  A step is one 8 x 16-pixel tile x one chunk of 32 channels (a pixel's
// chunk is 128 B of fp32: the halo tile has the geometry of the fp16 kernel's 64-channel chunk, and its LDS-DMA pieces are
// whole lines):
//   dw  waves 0-3: LDS-DMA of the halo tile two steps ahead (ring of three); fp32 KxK taps on the vector pipe (v_pk_fma_f32
//                  over channel pairs, taps in registers), result split hi = fp16(x), lo = fp16(x - hi) and stored as two
//                  128-pixel x 32-channel fp16 tiles (64-byte rows, conv16x3.hip's chunk swizzle)
//   mma waves 4-7: pointwise product of the PREVIOUS step's tiles: per 16 x 16 fragment pair w_lo.x_hi, w_hi.x_hi, w_hi.x_lo
//                  (conv16x3.hip's products, in its order) into accumulators that live across the C / 32 chunks of a tile
//                  and start from the bias; weight fragments hi / lo straight from L2 (fragment-order pack), a step ahead
// One LDS-only barrier per step.  Summation order fixed: taps ky-major / kx-minor from 0 (an fmaf chain: the fp32 mode's
// dwconv32 order), K ascending in steps of 32 from the bias: a batch of N equals N single calls bit for bit.
// Not bit-identical to the unfused pair of launches (there the pointwise conv adds its bias at the end); both are the fp32
// result to ~1e-6 relative (tests/test_gpu_sepconv_x3.py holds the kernel to an fp64 reference).
#include "common.h"

namespace emp {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SX_TH = 8, SX_TW = 16;                  // output tile (rows x cols) = 128 pixels
constexpr int SX_IW = SX_TW + 4;
constexpr int sx_npix(int ks) { return (SX_TH + ks - 1) * SX_IW; }
constexpr int sx_halo_bytes(int ks) { return sx_npix(ks) * 128; }      // 32 fp32 channels per pixel
constexpr int SX_BT_HALF = 128 * 64;                  // 128 pixels x 32 channels fp16
constexpr int SX_BT_BYTES = 2 * SX_BT_HALF;           // hi tile, lo tile
constexpr int SX_NDW = 4, SX_NMW = 4;
constexpr int SX_CH = 32;                             // channels per chunk

struct SxParams {
  const float* in;
  int N, H, W, C, in_ld;
  const float* dww;      // [C/32][KS*KS][32] fp32 (sepx3_pack_dw)
  const half_t* pww;     // pointwise weights hi / lo in MFMA-fragment order (sepx3_pack_pw)
  const float* bias;     // [Cout] (never null: the launcher substitutes zeros)
  float* out;            // (N,H,W,out_ld) or nullptr (head mode)
  int out_ld;
  const float* zero;     // >= 2 KiB of zeros
  int tiles_x, tiles_y;
  int tiles;
  const float* hw;       // head mode: [hc][Cout] fp32
  const float* hb;       // [hc]
  int hc;
  float* hout;           // (N,hc) planes of `plane` floats
  int64_t plane;
};

template <int ACT>
__device__ __forceinline__ float sx_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

__device__ __forceinline__ void sx_tile_coords(const SxParams& p, int tile, int& n, int& y0, int& x0) {
  const int tx = tile % p.tiles_x;
  const int r = tile / p.tiles_x;
  const int ty = r % p.tiles_y;
  n = r / p.tiles_y;
  y0 = ty * SX_TH;
  x0 = tx * SX_TW;
}

__device__ __forceinline__ void sx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// B tiles: 64-byte rows (32 halfs of one pixel), 16-byte chunk c of row r at chunk c ^ sx_swz(r): conflict-free ds_read_b128
// fragments (lane -> row lane % 16, chunk lane / 16) -- conv16x3.hip's layout
__device__ __forceinline__ int sx_swz(int r) { return (-((r & 15) >> 2)) & 3; }

// MT = 16-cout MFMA row tiles per mma wave: Cout = 4 * 16 * MT (128 or 256)
template <int KS, int MT, bool HEAD, int ACT>
__global__ void __launch_bounds__(64 * (SX_NDW + SX_NMW), 1) sepconv_x3_kernel(const SxParams p) {
  constexpr int NDW = SX_NDW, NMW = SX_NMW;
  constexpr int KK = KS * KS, PAD = KS / 2;
  constexpr int NPIX = sx_npix(KS), NST = NPIX / 8, HALO_BYTES = sx_halo_bytes(KS);
  constexpr int NT = 64 * (NDW + NMW);
  constexpr int R = 2;                      // output rows per depthwise thread (16 channel pairs x 4 column groups x 4 row groups)
  constexpr int COUT = NMW * 16 * MT;
  constexpr int NDMA = 32 / NDW;            // LDS-DMA instructions per dw wave and step
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* const halo = lds;                                   // 3 x HALO_BYTES (ring)
  char* const bt = lds + 3 * HALO_BYTES;                    // 2 x SX_BT_BYTES
  float* const hwl = reinterpret_cast<float*>(bt + 2 * SX_BT_BYTES);   // HEAD: [2][Cout] head weights,
  float* const red = hwl + 2 * COUT;                        //       [NMW waves][128 px][2] partial sums

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int NC = p.C / SX_CH;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
  auto tile_of = [&](int it) { return (it * 8 + xcd) * nx + jx; };
  int my_tiles = 0;
  while (tile_of(my_tiles) < p.tiles) ++my_tiles;
  const int S = my_tiles * NC;
  const int LAST = S + (HEAD ? 1 : 0);

  if (HEAD) {
    for (int i = tid; i < p.hc * COUT; i += NT) hwl[i] = p.hw[i];
  }

  if (wave < NDW) {
    // ------------------------------------------------------------------ dw role
    // thread = channel pair cp x columns 4cg..4cg+3 x rows R*rg..R*rg+R-1 of the 8 x 16 tile
    const int cp = lane & 15, combo = wave * 4 + (lane >> 4), cg = combo & 3, rg = combo >> 2;
    const int hoff = ((R * rg) * SX_IW + 4 * cg) * 128 + cp * 8;
    const float* sptr[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) sptr[k] = p.zero;
    auto stage = [&](int tile_it, int ch, int ringslot, bool valid) {
      if (!valid) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) sptr[k] = p.zero;
      } else if (ch == 0) {
        int n, y0, x0;
        sx_tile_coords(p, tile_of(tile_it), n, y0, x0);
        const float* src = p.in + (size_t)n * p.H * p.W * p.in_ld + (lane & 7) * 4;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
          const int i = min(wave + NDW * k, NST - 1);
          const int q = i * 8 + (lane >> 3);
          const int py = q / SX_IW, px = q - py * SX_IW;
          const int iy = y0 + py - PAD, ix = x0 + px - PAD;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          sptr[k] = ok ? src + ((size_t)iy * p.W + ix) * p.in_ld : p.zero;
        }
      }
      char* hb = halo + ringslot * HALO_BYTES;
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int i = min(wave + NDW * k, NST - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sptr[k],
                                         (__attribute__((address_space(3))) void*)(hb + i * 1024), 16, 0, 0);
        sptr[k] += SX_CH;      // next chunk (zero-page pointers stay inside the 2 KiB page: <= NC + 2 increments of 128 B)
      }
    };
    f32x2 w[KK], wn[KK];
    auto load_taps = [&](int ch) {
      const f32x2* b = reinterpret_cast<const f32x2*>(p.dww + (size_t)ch * (KK * SX_CH)) + cp;
#pragma unroll
      for (int t = 0; t < KK; ++t) wn[t] = b[t * 16];
    };
    stage(0, 0, 0, S > 0);
    stage(0, NC > 1 ? 1 : 0, 1, S > 1);
    load_taps(0);
    __syncthreads();
    int ring = 0, chn = 0, c2 = 2 % NC, t2 = 2 / NC, ring2 = 2;
    for (int g = 0; g <= LAST; ++g) {
      chn = chn + 1 == NC ? 0 : chn + 1;
#pragma unroll
      for (int t = 0; t < KK; ++t) w[t] = wn[t];
#pragma unroll
      for (int t = 0; t < KK; ++t) asm volatile("" : "+v"(w[t]));      // pins the wait in front of the DMA issue below
      __builtin_amdgcn_sched_barrier(0);
      load_taps(chn);
      stage(t2, c2, ring2, g + 2 < S);
      __builtin_amdgcn_sched_barrier(0);
      if (g < S) {
        const char* hb = halo + ring * HALO_BYTES + hoff;
        f32x2 acc[R][4];
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[y][j] = f32x2{0.f, 0.f};
        f32x2 nxt[4 + KS - 1];
#pragma unroll
        for (int c = 0; c < 4 + KS - 1; ++c) nxt[c] = *reinterpret_cast<const f32x2*>(hb + c * 128);
#pragma unroll
        for (int r = 0; r < R + KS - 1; ++r) {
          f32x2 x[4 + KS - 1];
#pragma unroll
          for (int c = 0; c < 4 + KS - 1; ++c) x[c] = nxt[c];
          if (r + 1 < R + KS - 1) {
#pragma unroll
            for (int c = 0; c < 4 + KS - 1; ++c) nxt[c] = *reinterpret_cast<const f32x2*>(hb + ((r + 1) * SX_IW + c) * 128);
          }
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            const int y = r - ky;
            if (y < 0 || y >= R) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[y][j] = __builtin_elementwise_fma(x[j + kx], w[ky * KS + kx], acc[y][j]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        char* bb = bt + (g & 1) * SX_BT_BYTES;
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int px = (R * rg + y) * SX_TW + 4 * cg + j;
            f16x2 h, l;
            h[0] = (half_t)acc[y][j][0]; h[1] = (half_t)acc[y][j][1];
            l[0] = (half_t)(acc[y][j][0] - (float)h[0]); l[1] = (half_t)(acc[y][j][1] - (float)h[1]);
            char* q = bb + px * 64 + (((cp >> 2) ^ sx_swz(px)) << 4) + (cp & 3) * 4;
            *reinterpret_cast<f16x2*>(q) = h;
            *reinterpret_cast<f16x2*>(q + SX_BT_HALF) = l;
          }
      }
      if (++c2 == NC) { c2 = 0; ++t2; }
      ring = ring == 2 ? 0 : ring + 1;
      ring2 = ring2 == 2 ? 0 : ring2 + 1;
      sx_barrier();
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) asm volatile("" :: "v"(wn[t]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------------------------------------------ mma role
    const int wm = wave - NDW, g16 = lane >> 4, n16 = lane & 15;
    __builtin_amdgcn_s_setprio(3);
    // weights: [chunk][wave][tile][hi | lo][lane][8 halfs]
    const f16x8* const abase = reinterpret_cast<const f16x8*>(p.pww) + (size_t)wm * MT * 2 * 64 + lane;
    constexpr int A_CHUNK = NMW * MT * 2 * 64;          // fragments (16 B) per 32-channel chunk
    const f32x4* const bbase = reinterpret_cast<const f32x4*>(p.bias + wm * 16 * MT + g16 * 8);
    f32x4 acc[MT][8];
    f16x8 ah[MT], al[MT], nah[MT], nal[MT];
    auto load_a = [&](int ch) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        nah[t] = abase[(size_t)ch * A_CHUNK + t * (2 * 64)];
        nal[t] = abase[(size_t)ch * A_CHUNK + t * (2 * 64) + 64];
      }
    };
    const int fidx = wm * 64 + lane, fh = fidx >> 7, fpx = fidx & 127;
    float fhb = 0.f;
    if (HEAD && fh < p.hc) fhb = p.hb[fh];
    load_a(0);
    __syncthreads();
    int c1 = NC - 1, t1 = -1;
    for (int g = 0; g <= LAST; ++g) {
      if (HEAD && g >= 2 && c1 == 0 && fh < p.hc) {
        int n, y0, x0;
        sx_tile_coords(p, tile_of(t1 - 1), n, y0, x0);
        const int oy = y0 + (fpx >> 4), ox = x0 + (fpx & 15);
        float v = fhb;
#pragma unroll
        for (int wv = 0; wv < NMW; ++wv) v += red[(wv * 128 + fpx) * 2 + fh];
        if (oy < p.H && ox < p.W) p.hout[((size_t)n * p.hc + fh) * p.plane + (size_t)oy * p.W + ox] = v;
      }
      const bool do_mma = g >= 1 && g - 1 < S;
      const bool first = do_mma && c1 == 0;
      if (first) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const f32x4 bv = bbase[(t >> 1) * 8 + (t & 1)];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[t][nt] = bv;
        }
      }
      const char* bb = bt + ((g - 1) & 1) * SX_BT_BYTES + n16 * 64 + ((g16 ^ sx_swz(n16)) << 4);
      auto bfrag = [&](int nt, int part) { return *reinterpret_cast<const f16x8*>(bb + part * SX_BT_HALF + nt * 16 * 64); };
      const int cn = c1 + 1 == NC ? 0 : c1 + 1;     // chunk g % NC, multiplied in step g + 1
#pragma unroll
      for (int t = 0; t < MT; ++t) { ah[t] = nah[t]; al[t] = nal[t]; }
      __builtin_amdgcn_sched_barrier(0);
      load_a(cn);      // a whole step to arrive
      __builtin_amdgcn_sched_barrier(0);
      if (do_mma) {
        f16x8 bh = bfrag(0, 0), bl = bfrag(0, 1);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          f16x8 nh, nl;
          if (nt + 1 < 8) { nh = bfrag(nt + 1, 0); nl = bfrag(nt + 1, 1); }
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t][nt], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][nt], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[t][nt], 0, 0, 0);
          if (nt + 1 < 8) { bh = nh; bl = nl; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_mma && c1 == NC - 1) {
        int n, y0, x0;
        sx_tile_coords(p, tile_of(t1), n, y0, x0);
        const int ox = x0 + n16;
        if (!HEAD) {
#pragma unroll
          for (int blk = 0; blk < MT / 2; ++blk) {
            const int cb = wm * 16 * MT + blk * 32 + g16 * 8;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              const int oy = y0 + nt;
              if (oy < p.H && ox < p.W) {
                float* op = p.out + (((size_t)n * p.H + oy) * p.W + ox) * p.out_ld + cb;
                *reinterpret_cast<float4*>(op) = make_float4(sx_act<ACT>(acc[2 * blk][nt][0]), sx_act<ACT>(acc[2 * blk][nt][1]),
                                                             sx_act<ACT>(acc[2 * blk][nt][2]), sx_act<ACT>(acc[2 * blk][nt][3]));
                *reinterpret_cast<float4*>(op + 4) = make_float4(sx_act<ACT>(acc[2 * blk + 1][nt][0]), sx_act<ACT>(acc[2 * blk + 1][nt][1]),
                                                                 sx_act<ACT>(acc[2 * blk + 1][nt][2]), sx_act<ACT>(acc[2 * blk + 1][nt][3]));
              }
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[t][nt][e] = sx_act<ACT>(acc[t][nt][e]);
#pragma unroll 1
          for (int h = 0; h < p.hc; ++h) {
            float sum[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) sum[nt] = 0.f;
#pragma unroll
            for (int blk = 0; blk < MT / 2; ++blk) {
              const float* hp = hwl + h * COUT + wm * 16 * MT + blk * 32 + g16 * 8;
              float hv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = hp[e];
#pragma unroll
              for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  sum[nt] = fmaf(acc[2 * blk][nt][e], hv[e], sum[nt]);
                  sum[nt] = fmaf(acc[2 * blk + 1][nt][e], hv[4 + e], sum[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              float v = sum[nt];
              v += __shfl_xor(v, 16);
              v += __shfl_xor(v, 32);
              if (g16 == 0) red[(wm * 128 + nt * 16 + n16) * 2 + h] = v;
            }
          }
        }
      }
      if (++c1 == NC) { c1 = 0; }
      if (c1 == 0) ++t1;
      sx_barrier();
    }
  }
}

// depthwise taps (KK, ld) fp32 -> chunk-major [C/32][KK][32] fp32
__global__ void __launch_bounds__(256) sepx3_pack_dw_kernel(const float* __restrict__ w, int KK, int C, int ld, float* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < KK * C; i += gridDim.x * 256) {
    const int c = i & 31, t = (i >> 5) % KK, ch = (i >> 5) / KK;
    out[i] = w[(size_t)t * ld + ch * 32 + c];
  }
}

// (Cout, pw_ld) fp32 row-major -> fp16 hi / lo in the fragment order of the kernel above: [chunk][wave][tile][hi | lo][lane][8]
__global__ void __launch_bounds__(256) sepx3_pack_pw_kernel(const float* __restrict__ w, int pw_ld, int C, int Cout, int MT,
                                                            half_t* __restrict__ out) {
  const int total = 2 * C * Cout / 8;      // 16-byte fragments
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int r = i;
    const int lane = r & 63; r >>= 6;
    const int part = r & 1; r >>= 1;
    const int t = r % MT; r /= MT;
    const int wm = r % SX_NMW;
    const int ch = r / SX_NMW;
    const int n16 = lane & 15, g16 = lane >> 4, tp = t & 1;
    const int co = wm * 16 * MT + (t >> 1) * 32 + ((n16 >> 2) << 3) + (tp << 2) + (n16 & 3);
    const float* src = w + (size_t)co * pw_ld + ch * SX_CH + g16 * 8;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half_t h = (half_t)src[e];
      v[e] = part ? (half_t)(src[e] - (float)h) : h;
    }
    *reinterpret_cast<f16x8*>(out + (size_t)i * 8) = v;
  }
}

template <int KS, int MT, bool HEAD>
int sx_launch(const SxParams& p, int act, size_t lds_bytes, int grid, hipStream_t s) {
  auto go = [&](auto kern) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (SX_NDW + SX_NMW)), lds_bytes, s, p);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  };
  if (act == 1) return go(&sepconv_x3_kernel<KS, MT, HEAD, 1>);
  if (act == 2) return go(&sepconv_x3_kernel<KS, MT, HEAD, 2>);
  return go(&sepconv_x3_kernel<KS, MT, HEAD, 0>);
}

}  // namespace

static size_t sepx3_lds_bytes(int Cout, int head_c, int ks) {
  return 3 * sx_halo_bytes(ks) + 2 * SX_BT_BYTES + (head_c ? (size_t)2 * Cout * 4 + (size_t)SX_NMW * 128 * 2 * 4 : 0);
}

bool sepconv_x3_supported(int C, int Cout, int head_c, int ks) {
  // C / 32 + 2 increments of 128 B must stay inside the 2 KiB zero page
  return C % SX_CH == 0 && C >= 64 && C <= 448 && (Cout == 128 || Cout == 256) && head_c >= 0 && head_c <= 2 && (ks == 5 || (ks == 3 && head_c == 0));
}

int64_t sepx3_pw_halfs(int C, int Cout) { return (int64_t)2 * C * Cout; }

int launch_sepx3_pack_pw(const float* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s) {
  EMP_REQUIRE(w && packed && C % SX_CH == 0 && (Cout == 128 || Cout == 256) && pw_ld >= C, "sepx3 pack_pw: bad shape");
  const int total = 2 * C * Cout / 8;
  hipLaunchKernelGGL(sepx3_pack_pw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, pw_ld, C, Cout, Cout / 64, packed);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_sepx3_pack_dw(const float* w, int ks, int C, int ld, float* packed, hipStream_t s) {
  EMP_REQUIRE(w && packed && (ks == 3 || ks == 5) && C % SX_CH == 0 && C > 0 && ld >= C, "sepx3 pack_dw: bad shape");
  hipLaunchKernelGGL(sepx3_pack_dw_kernel, dim3(cdiv(ks * ks * C, 256)), dim3(256), 0, s, w, ks * ks, C, ld, packed);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// out != nullptr: y = act(pw(dw(x)) + bias) -> (N,H,W,out_ld) fp32.
// head_c > 0   : hout[n][h] = head_w[h] . y + head_b[h] as fp32 planes of `plane` floats; y is not stored.
int launch_sepconv_x3(const float* in, int N, int H, int W, int C, int in_ld, const float* dww, const half_t* pww, const float* bias,
                      int Cout, int act, float* out, int out_ld, const float* head_w, const float* head_b, int head_c, float* hout,
                      int64_t plane, hipStream_t s, int ks) {
  EMP_REQUIRE(sepconv_x3_supported(C, Cout, head_c, ks), "sepconv_x3: unsupported shape C=%d Cout=%d head=%d k=%d", C, Cout, head_c, ks);
  EMP_REQUIRE(act >= 0 && act <= 2, "sepconv_x3: bad activation %d", act);
  EMP_REQUIRE((head_c > 0) != (out != nullptr), "sepconv_x3: exactly one of the feature / head outputs");
  EMP_REQUIRE(in && dww && pww && in_ld % 4 == 0 && in_ld >= C && ((uintptr_t)in % 16) == 0 && (out == nullptr || (out_ld % 4 == 0 && ((uintptr_t)out % 16) == 0)) &&
                  (!bias || ((uintptr_t)bias % 16) == 0), "sepconv_x3: 16-byte row alignment");
  SxParams p{};
  p.zero = reinterpret_cast<const float*>(zero_page());
  EMP_REQUIRE(p.zero != nullptr, "sepconv_x3: no zero page");
  p.in = in; p.N = N; p.H = H; p.W = W; p.C = C; p.in_ld = in_ld;
  p.dww = dww; p.pww = pww;
  p.bias = bias ? bias : p.zero;      // Cout * 4 <= 1 KiB of the 2 KiB zero page
  p.out = out; p.out_ld = out_ld;
  p.tiles_x = cdiv(W, SX_TW); p.tiles_y = cdiv(H, SX_TH);
  const int64_t tiles = (int64_t)N * p.tiles_x * p.tiles_y;
  EMP_REQUIRE(tiles < (1ll << 30), "sepconv_x3: too many tiles");
  p.tiles = (int)tiles;
  p.hw = head_w; p.hb = head_b; p.hc = head_c; p.hout = hout; p.plane = plane;
  int dev = 0, n_cu = 0;
  EMP_CHECK_HIP(hipGetDevice(&dev));
  EMP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = n_cu >= 8 ? (n_cu / 8) * 8 : 8;
  const size_t lds_bytes = sepx3_lds_bytes(Cout, head_c, ks);
  if (ks == 3) return Cout == 256 ? sx_launch<3, 4, false>(p, act, lds_bytes, grid, s) : sx_launch<3, 2, false>(p, act, lds_bytes, grid, s);
  if (Cout == 256) return head_c ? sx_launch<5, 4, true>(p, act, lds_bytes, grid, s) : sx_launch<5, 4, false>(p, act, lds_bytes, grid, s);
  return head_c ? sx_launch<5, 2, true>(p, act, lds_bytes, grid, s) : sx_launch<5, 2, false>(p, act, lds_bytes, grid, s);
}

}  // namespace emp
