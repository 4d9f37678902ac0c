// Fused depthwise-separable convolution for gfx950 (MI355X):
//   y = act( pw( dw5x5(x) ) + bias )            [+ optional 1x1 head on y]
// replaces nn.Conv2d(C,C,5,groups=C,pad=2,bias=False) -> nn.Conv2d(C,Cout,1) -> folded BN -> ReLU
// (the 'depthwise_separable_conv' of the Panoptic-DeepLab decoder / heads, models/blocks.py, called from
// models/decoders/panoptic_deeplab.py and models/heads) -- and, for the heads, the final 1x1 conv
// (Cout -> hc <= 4) so that the Cout-channel map never leaves the chip.
//
// There is no non-linearity between the depthwise and the pointwise conv, so the depthwise output is
// only ever the B operand of the pointwise GEMM: it is produced straight into LDS.
//
// One persistent workgroup per CU, 512 threads = 8 waves with two roles (one wave of each per SIMD, so
// the vector ALU and the matrix pipe of every SIMD are both fed):
//   waves 0-3  "dw":  fp32 depthwise 5x5 on the VALU (v_pk_fma_f32 over channel pairs) from a
//                     12x20-pixel x 64-channel halo tile in LDS into a 128-pixel x 64-channel fp16
//                     B tile in LDS (XOR-swizzled rows of 128 B).
//   waves 4-7  "mma": LDS-DMA (global_load_lds) of the NEXT halo tile, then the pointwise GEMM of the
//                     PREVIOUS B tile: 16x16x32 f16 MFMA, A fragments (pointwise weights) straight
//                     from L2, accumulators (Cout x 128 pixels) resident across the C/64 channel
//                     chunks; epilogue bias + act + 16-byte NHWC stores (or the head reduction).
// The pipeline is flattened over (tile, chunk) steps with ONE barrier per step, so fill and drain are
// paid once per workgroup, not per tile.  Halo tiles are DMA'd TWO steps ahead into a ring of three
// (HBM latency under load is about one step); the barrier fences LDS only (lgkmcnt), and the mma waves
// wait with a counted vmcnt so that the youngest DMA batch stays in flight across it.  [measured: with
// one-step prefetch and a full __syncthreads the mma role alone took 1.07 ms of a 1.15 ms launch.]
//
// Summation order of the depthwise taps (ky-major, kx-minor, fp32 fma chain from 0) and of the GEMM
// (ascending 32-channel K-steps) equals the unfused dwconv_kernel + conv_igemm_kernel pair.
#include "common.h"

namespace emp {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SC_TH = 8, SC_TW = 16;                  // output tile (rows x cols) = 128 pixels
// halo tile of a KS x KS depthwise kernel: (8 + KS - 1) rows x 20 columns (KS = 5: 12 x 20 = 240 pixels = 30 LDS-DMA
// instructions of 8 pixels; KS = 3: 10 x 20 = 200 pixels = 25 instructions, the two right-most columns unused)
constexpr int SC_IW = SC_TW + 4;
constexpr int sc_npix(int ks) { return (SC_TH + ks - 1) * SC_IW; }
constexpr int sc_halo_bytes(int ks) { return sc_npix(ks) * 128; }
constexpr int SC_BT_BYTES = 128 * 128;                // 16384

struct SepParams {
  const half_t* in;
  int N, H, W, C, in_ld;
  const half_t* dww;     // [25][C] fp16
  const half_t* pww;     // pointwise weights in MFMA-fragment order (sepconv5_pack_pw)
  const float* bias;     // [Cout]
  half_t* out;           // (N,H,W,out_ld) or nullptr (head mode)
  int out_ld, act;
  const half_t* zero;    // >= 2 KiB of zeros
  int tiles_x, tiles_y;
  int tiles;
  const float* hw;       // head mode: [hc][Cout] fp32
  const float* hb;       // [hc]
  int hc;
  float* hout;           // (N,hc) planes of `plane` floats
  int64_t plane;
};

template <int ACT>
__device__ __forceinline__ float sc_act(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

__device__ __forceinline__ void tile_coords(const SepParams& p, int tile, int& n, int& y0, int& x0) {
  const int tx = tile % p.tiles_x;
  const int r = tile / p.tiles_x;
  const int ty = r % p.tiles_y;
  n = r / p.tiles_y;
  y0 = ty * SC_TH;
  x0 = tx * SC_TW;
}

// workgroup barrier that orders LDS traffic only: global loads / LDS-DMA stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  // Raw barrier that orders LDS traffic only.  A fence (or __syncthreads) would also drain vmcnt: an in-flight
  // LDS-DMA is a pending LDS write on the VM counter.  The "memory" clobber keeps the compiler from moving
  // memory accesses across it; DMA completion is handled by the counted vmcnt waits of the issuing waves.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NDW = number of depthwise waves (4 or 8: one or two per SIMD), NMW = number of mma waves (4 or 8),
// MT = 16-cout MFMA row tiles per mma wave: Cout = NMW * 16 * MT.
template <int KS, int NDW, int NMW, int MT, bool HEAD, int ACT>
__global__ void __launch_bounds__(64 * (NDW + NMW), 1) sepconv5_kernel(const SepParams p) {
  constexpr int KK = KS * KS, PAD = KS / 2;
  constexpr int SC_NPIX = sc_npix(KS), SC_NST = SC_NPIX / 8, SC_HALO_BYTES = sc_halo_bytes(KS);
  constexpr int NT = 64 * (NDW + NMW);      // threads
  constexpr int R = 16 / NDW;               // output rows per depthwise thread (4 or 2)
  constexpr int COUT = NMW * 16 * MT;
  constexpr int NDMA = 32 / NMW;            // LDS-DMA instructions per mma wave and step (8 or 4)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* const halo = lds;                                   // 3 x SC_HALO_BYTES (ring)
  char* const bt = lds + 3 * SC_HALO_BYTES;                 // 2 x SC_BT_BYTES
  float* const dwl = reinterpret_cast<float*>(bt + 2 * SC_BT_BYTES);   // [C/64][25][64] fp32
  float* const biasl = dwl + p.C * KK;                      // [Cout] epilogue bias
  float* const hwl = biasl + COUT;                          // HEAD: [2][Cout] head weights,
  float* const red = hwl + 2 * COUT;                        //       [NMW waves][128 px][2] partial sums

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int NC = p.C >> 6;
  // XCD-aware tile order: at iteration `it` XCD x owns tiles [(it*8+x)*nx, +nx)
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
  auto tile_of = [&](int it) { return (it * 8 + xcd) * nx + jx; };
  int my_tiles = 0;
  while (tile_of(my_tiles) < p.tiles) ++my_tiles;
  const int S = my_tiles * NC;
  const int LAST = S + (HEAD ? 1 : 0);      // last step index (HEAD: one more to finish the last tile's head)

  // depthwise taps -> LDS, fp32, chunk-major; epilogue constants
  for (int i = tid; i < p.C * KK; i += NT) {
    const int t = i / p.C, c = i - t * p.C;
    dwl[((c >> 6) * KK + t) * 64 + (c & 63)] = (float)p.dww[i];
  }
  for (int i = tid; i < COUT; i += NT) biasl[i] = p.bias ? p.bias[i] : 0.f;
  if (HEAD) {
    for (int i = tid; i < p.hc * COUT; i += NT) hwl[i] = p.hw[i];
  }

  if (wave < NDW) {
    // ------------------------------------------------------------------ dw role
    // thread = channel pair cp x columns 4cg..4cg+3 x rows R*rg..R*rg+R-1 of the 8x16 tile
    const int cp = lane & 31, combo = wave * 2 + (lane >> 5), cg = combo & 3, rg = combo >> 2;
    const int hoff = ((R * rg) * SC_IW + 4 * cg) * 128 + cp * 4;
    const int sw = (cp >> 2), sub = (cp & 3) * 4;
    __syncthreads();
    f32x2 w[KK];            // taps of the chunk of the coming step: read before the barrier, off the critical path
#pragma unroll
    for (int t = 0; t < KK; ++t) w[t] = reinterpret_cast<const f32x2*>(dwl)[t * 32 + cp];
    int ring = 0, chn = 0;   // g % 3, (g + 1) % NC
    for (int g = 0; g <= LAST; ++g) {
      chn = chn + 1 == NC ? 0 : chn + 1;
      if (g < S) {
        const char* hb = halo + ring * SC_HALO_BYTES + hoff;
        f32x2 acc[R][4];
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[y][j] = f32x2{0.f, 0.f};
        f16x2 nxt[4 + KS - 1];
#pragma unroll
        for (int c = 0; c < 4 + KS - 1; ++c) nxt[c] = *reinterpret_cast<const f16x2*>(hb + c * 128);
#pragma unroll
        for (int r = 0; r < R + KS - 1; ++r) {
          f32x2 x[4 + KS - 1];
#pragma unroll
          for (int c = 0; c < 4 + KS - 1; ++c) x[c] = f32x2{(float)nxt[c][0], (float)nxt[c][1]};
          if (r + 1 < R + KS - 1) {
#pragma unroll
            for (int c = 0; c < 4 + KS - 1; ++c)
              nxt[c] = *reinterpret_cast<const f16x2*>(hb + ((r + 1) * SC_IW + c) * 128);
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
        char* bb = bt + (g & 1) * SC_BT_BYTES;
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int px = (R * rg + y) * SC_TW + 4 * cg + j;
            f16x2 h;
            h[0] = (half_t)acc[y][j][0];
            h[1] = (half_t)acc[y][j][1];
            *reinterpret_cast<f16x2*>(bb + px * 128 + ((sw ^ (px & 7)) << 4) + sub) = h;
          }
        const f32x2* wl = reinterpret_cast<const f32x2*>(dwl + chn * KK * 64) + cp;
#pragma unroll
        for (int t = 0; t < KK; ++t) w[t] = wl[t * 32];
      }
      ring = ring == 2 ? 0 : ring + 1;
      lds_barrier();
    }
  } else {
    // ------------------------------------------------------------------ mma role
    const int wm = wave - NDW, g16 = lane >> 4, n16 = lane & 15;
    __builtin_amdgcn_s_setprio(3);      // few instructions, long latencies: issue ahead of the VALU-bound dw wave
    // Pointwise weights, pre-packed in fragment order [chunk][wave][tile][k-half][lane][8]: every load instruction
    // reads 1 KiB of whole cache lines (fragment-shaped loads of 16 rows x 64 B cost the texture addresser twice
    // as much, and the addresser is what bounds this role).
    const half_t* const abase = p.pww + ((size_t)wm * MT * 2 * 64 + lane) * 8;
    constexpr int A_CHUNK = NMW * MT * 2 * 64 * 8;      // halves per 64-channel chunk
    // Halo DMA: wave wm issues slots i = wm + NMW*k.  Always exactly NDMA instructions per step (the two spare
    // slots repeat slot 29 with the same data; a step without a halo to fetch copies the zero page into the free
    // ring slot), so that every step has the same VM issue sequence and the counted wait is a constant.
    const half_t* sptr[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) sptr[k] = p.zero;
    auto stage = [&](int tile_it, int ch, int ringslot, bool valid) {
      if (!valid) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) sptr[k] = p.zero;
      } else if (ch == 0) {
        int n, y0, x0;
        tile_coords(p, tile_of(tile_it), n, y0, x0);
        const half_t* src = p.in + (size_t)n * p.H * p.W * p.in_ld + (lane & 7) * 8;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
          const int i = min(wm + NMW * k, SC_NST - 1);
          const int q = i * 8 + (lane >> 3);
          const int py = q / SC_IW, px = q - py * SC_IW;
          const int iy = y0 + py - PAD, ix = x0 + px - PAD;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          sptr[k] = ok ? src + ((size_t)iy * p.W + ix) * p.in_ld : p.zero;
        }
      }
      char* hb = halo + ringslot * SC_HALO_BYTES;
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int i = min(wm + NMW * k, SC_NST - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sptr[k],
                                         (__attribute__((address_space(3))) void*)(hb + i * 1024), 16, 0, 0);
        sptr[k] += 64;      // next chunk (zero-page pointers stay inside the 2 KiB page: <= NC + 2 increments)
      }
    };
    f32x4 acc[MT][8];
    f16x8 a[MT][2];
    auto load_a = [&](int ch) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const half_t* ap = abase + (size_t)ch * A_CHUNK + t * (2 * 64 * 8);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                     : "=&v"(a[t][0]), "=&v"(a[t][1]) : "v"(ap) : "memory");
      }
    };
    // head finishing lanes: output idx = wm*64 + lane -> (class h = idx >> 7, pixel idx & 127)
    const int fidx = wm * 64 + lane, fh = fidx >> 7, fpx = fidx & 127;
    float fhb = 0.f;
    if (HEAD && fh < p.hc) fhb = p.hb[fh];
    load_a(0);
    stage(0, 0, 0, S > 0);
    stage(0, NC > 1 ? 1 : 0, 1, S > 1);      // S > 1 implies NC >= 2 (launcher) or a second tile; see launcher
    __syncthreads();                          // full wait; dwl / epilogue constants visible
    // step counters: c1 = (g-1) % NC with tile t1, c2 = (g+2) % NC with tile t2, ring2 = (g+2) % 3
    int c1 = NC - 1, t1 = -1, c2 = 2 % NC, t2 = 2 / NC, ring2 = 2;
    for (int g = 0; g <= LAST; ++g) {
      if (HEAD && g >= 2 && c1 == 0 && fh < p.hc) {
        // finish the head of tile t1 - 1 (its last chunk was multiplied in the previous step)
        int n, y0, x0;
        tile_coords(p, tile_of(t1 - 1), n, y0, x0);
        const int oy = y0 + (fpx >> 4), ox = x0 + (fpx & 15);
        float v = fhb;
#pragma unroll
        for (int wv = 0; wv < NMW; ++wv) v += red[(wv * 128 + fpx) * 2 + fh];
        if (oy < p.H && ox < p.W) p.hout[((size_t)n * p.hc + fh) * p.plane + (size_t)oy * p.W + ox] = v;
      }
      const bool do_mma = g >= 1 && g - 1 < S;
      // VM issue order: [weights of this step, issued at the end of the previous step] [stores] [NDMA DMA loads of
      // stage(g+2)].  Loads retire in order, so vmcnt(NDMA) == "the weights AND the halo of step g+1 (DMA'd one
      // step ago) have landed, only the youngest DMA batch may still be in flight"; stores only make the wait
      // stricter.  The weight loads come from inline asm so that the compiler's own wait-count insertion does
      // not see them (it answers a pending load behind LDS-DMA traffic with vmcnt(0), draining the DMA batch
      // that is meant to stay in flight); the counted wait names the registers as operands, which orders it
      // before their first use.
      stage(t2, c2, ring2, g + 2 < S);
      if (MT == 4) {
        if (NDMA == 8)
          asm volatile("s_waitcnt vmcnt(8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[MT - 2][0]), "+v"(a[MT - 2][1]), "+v"(a[MT - 1][0]), "+v"(a[MT - 1][1]) :: "memory");
        else
          asm volatile("s_waitcnt vmcnt(4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                       "+v"(a[MT - 2][0]), "+v"(a[MT - 2][1]), "+v"(a[MT - 1][0]), "+v"(a[MT - 1][1]) :: "memory");
      } else {
        if (NDMA == 8)
          asm volatile("s_waitcnt vmcnt(8)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) :: "memory");
        else
          asm volatile("s_waitcnt vmcnt(4)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]) :: "memory");
      }
      if (do_mma) {
        if (c1 == 0) {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* bb = bt + ((g - 1) & 1) * SC_BT_BYTES + n16 * 128;
        auto bfrag = [&](int i) {   // i = nt*2 + ks
          return *reinterpret_cast<const f16x8*>(bb + (i >> 1) * 16 * 128 + ((((i & 1) * 4 + g16) ^ (n16 & 7)) << 4));
        };
        f16x8 b0 = bfrag(0), b1 = bfrag(1);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int nt = i >> 1;
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][0], b0, acc[t][nt], 0, 0, 0);
          if (i + 2 < 16) b0 = bfrag(i + 2);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][1], b1, acc[t][nt], 0, 0, 0);
          if (i + 3 < 16) b1 = bfrag(i + 3);
        }
      }
      load_a(c1 + 1 == NC ? 0 : c1 + 1);   // weights of chunk g % NC, multiplied in step g+1; `a` is dead until the next wait
      if (do_mma && c1 == NC - 1) {
        int n, y0, x0;
        tile_coords(p, tile_of(t1), n, y0, x0);
        const int ox = x0 + n16;
        if (!HEAD) {
#pragma unroll
          for (int blk = 0; blk < MT / 2; ++blk) {
            const int cb = wm * 16 * MT + blk * 32 + g16 * 8;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = biasl[cb + e];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)sc_act<ACT>(acc[2 * blk][nt][e] + bv[e]);
                o[4 + e] = (half_t)sc_act<ACT>(acc[2 * blk + 1][nt][e] + bv[4 + e]);
              }
              const int oy = y0 + nt;
              if (oy < p.H && ox < p.W)
                *reinterpret_cast<f16x8*>(p.out + (((size_t)n * p.H + oy) * p.W + ox) * p.out_ld + cb) = o;
            }
          }
        } else {
#pragma unroll
          for (int blk = 0; blk < MT / 2; ++blk) {
            const int cb = wm * 16 * MT + blk * 32 + g16 * 8;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = biasl[cb + e];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[2 * blk][nt][e] = sc_act<ACT>(acc[2 * blk][nt][e] + bv[e]);
                acc[2 * blk + 1][nt][e] = sc_act<ACT>(acc[2 * blk + 1][nt][e] + bv[4 + e]);
              }
          }
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
      // advance the step counters
      if (++c1 == NC) { c1 = 0; }
      if (c1 == 0) ++t1;
      if (++c2 == NC) { c2 = 0; ++t2; }
      ring2 = ring2 == 2 ? 0 : ring2 + 1;
      lds_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the last (dummy) DMA batch and weight loads
  }
}

// (Cout, pw_ld) row-major -> fragment order of the kernel above; NMW / MT as chosen by launch_sepconv5 for this Cout
__global__ void __launch_bounds__(256) sepconv5_pack_pw_kernel(const half_t* __restrict__ w, int pw_ld, int C, int Cout,
                                                               int NMW, int MT, half_t* __restrict__ out) {
  const int total = C * Cout / 8;      // 16-byte fragments
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int r = i;
    const int lane = r & 63; r >>= 6;
    const int ks = r & 1; r >>= 1;
    const int t = r % MT; r /= MT;
    const int wm = r % NMW;
    const int ch = r / NMW;
    const int n16 = lane & 15, g16 = lane >> 4, tp = t & 1;
    const int co = wm * 16 * MT + (t >> 1) * 32 + ((n16 >> 2) << 3) + (tp << 2) + (n16 & 3);
    const f16x8 v = *reinterpret_cast<const f16x8*>(w + (size_t)co * pw_ld + ch * 64 + ks * 32 + g16 * 8);
    *reinterpret_cast<f16x8*>(out + (size_t)i * 8) = v;
  }
}

template <int KS, int NDW, int NMW, int MT, bool HEAD, int ACT>
int launch_act(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&sepconv5_kernel<KS, NDW, NMW, MT, HEAD, ACT>), 160 * 1024)) return rc;
  hipLaunchKernelGGL((sepconv5_kernel<KS, NDW, NMW, MT, HEAD, ACT>), dim3(grid), dim3(64 * (NDW + NMW)), lds_bytes, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <int KS, int NDW, int NMW, int MT, bool HEAD>
int launch_one(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  if (p.act == 1) return launch_act<KS, NDW, NMW, MT, HEAD, 1>(p, lds_bytes, grid, s);
  if (p.act == 2) return launch_act<KS, NDW, NMW, MT, HEAD, 2>(p, lds_bytes, grid, s);
  return launch_act<KS, NDW, NMW, MT, HEAD, 0>(p, lds_bytes, grid, s);
}

}  // namespace

static size_t sepconv5_lds_bytes(int C, int Cout, int head_c, int ks = 5) {
  const int nmw = Cout == 256 ? 8 : 4;
  return 3 * sc_halo_bytes(ks) + 2 * SC_BT_BYTES + (size_t)C * ks * ks * 4 + (size_t)Cout * 4 +
         (head_c ? (size_t)2 * Cout * 4 + (size_t)nmw * 128 * 2 * 4 : 0);
}

bool sepconv5_supported(int C, int Cout, int head_c) {
  return C % 64 == 0 && C >= 128 && (Cout == 128 || Cout == 256) && head_c >= 0 && head_c <= 2 &&
         sepconv5_lds_bytes(C, Cout, head_c) <= 160 * 1024;   // C <= 320 (features) / C <= 192 (heads)
}

// out != nullptr: y = act(pw(dw(x)) + bias) -> (N,H,W,out_ld) fp16.
// head_c > 0   : hout[n][h] = head_w[h] . y + head_b[h] as fp32 planes of `plane` floats; y is not stored.
int launch_sepconv5_pack_pw(const half_t* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s) {
  EMP_REQUIRE(C % 64 == 0 && (Cout == 128 || Cout == 256) && pw_ld % 8 == 0 && pw_ld >= C, "sepconv5 pack: bad shape");
  const int nmw = Cout == 256 ? 8 : 4, mt = 2;
  const int total = C * Cout / 8;
  hipLaunchKernelGGL(sepconv5_pack_pw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, pw_ld, C, Cout, nmw, mt, packed);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_sepconv5(const half_t* in, int N, int H, int W, int C, int in_ld, const half_t* dww, const half_t* pww,
                    const float* bias, int Cout, int act, half_t* out, int out_ld, const float* head_w,
                    const float* head_b, int head_c, float* hout, int64_t plane, const half_t* zero, hipStream_t s,
                    int ks) {
  EMP_REQUIRE(ks == 5 || (ks == 3 && head_c == 0), "sepconv: depthwise kernel %d unsupported", ks);
  EMP_REQUIRE(sepconv5_supported(C, Cout, head_c), "sepconv5: unsupported shape C=%d Cout=%d head=%d", C, Cout, head_c);
  EMP_REQUIRE(act >= 0 && act <= 2, "sepconv5: bad activation %d", act);
  EMP_REQUIRE((head_c > 0) != (out != nullptr), "sepconv5: exactly one of the feature / head outputs");
  EMP_REQUIRE(in_ld % 8 == 0 && (out == nullptr || out_ld % 8 == 0), "sepconv5: 16-byte row alignment");
  SepParams p{};
  p.in = in; p.N = N; p.H = H; p.W = W; p.C = C; p.in_ld = in_ld;
  p.dww = dww; p.pww = pww; p.bias = bias;
  p.out = out; p.out_ld = out_ld; p.act = act; p.zero = zero;
  p.tiles_x = cdiv(W, SC_TW); p.tiles_y = cdiv(H, SC_TH);
  const int64_t tiles = (int64_t)N * p.tiles_x * p.tiles_y;
  EMP_REQUIRE(tiles < (1ll << 30), "sepconv5: too many tiles");
  p.tiles = (int)tiles;
  p.hw = head_w; p.hb = head_b; p.hc = head_c; p.hout = hout; p.plane = plane;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    EMP_CHECK_HIP(hipGetDevice(&dev));
    EMP_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = n_cu >= 8 ? (n_cu / 8) * 8 : 8;
  }
  const int grid = n_cu;
  const size_t lds_bytes = sepconv5_lds_bytes(C, Cout, head_c, ks);
  if (ks == 3) {     // BiFPN nodes (depthwise 3x3 -> pointwise -> BN -> SiLU); no head mode
    if (Cout == 256) return launch_one<3, 4, 8, 2, false>(p, lds_bytes, grid, s);
    return launch_one<3, 4, 4, 2, false>(p, lds_bytes, grid, s);
  }
  // (two depthwise waves per SIMD, <8, 8, 2>, measured no faster: the mma role's vector-memory issue bounds the step)
  if (Cout == 256)
    return head_c ? launch_one<5, 4, 8, 2, true>(p, lds_bytes, grid, s) : launch_one<5, 4, 8, 2, false>(p, lds_bytes, grid, s);
  return head_c ? launch_one<5, 4, 4, 2, true>(p, lds_bytes, grid, s) : launch_one<5, 4, 4, 2, false>(p, lds_bytes, grid, s);
}

}  // namespace emp
