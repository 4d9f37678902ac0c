// Fused depthwise-separable convolution for gfx950 (MI355X) with the DEPTHWISE CONV ON THE MATRIX PIPE (round 3):
//   y = act( pw( dwKxK(x) ) + bias )            [+ optional 1x1 head on y]
// same operator and the same two-role, one-barrier-per-step pipeline as sepconv_precise.hip (models/blocks.py:15-33 as
// used by decoders/panoptic_deeplab.py:68-80 and heads.py:12-15) -- only the depthwise stage differs.
//
// sepconv.hip / sepconv_precise.hip compute the depthwise taps on the vector pipe: 400 v_pk_fma_f32 + 128 conversions per
// wave and 64-channel step, issued by ONE wave per SIMD at ~5 cycles each -- the pole of both kernels (DESIGN findings
// 16, 25).  Here a depthwise wave owns a 16-channel block and feeds the MFMA unit instead:
//   D[16 pixels of a tile row][16 channels] += A[16 pixels][K] . B[K][16 channels],   K = 32 = 2 taps x 16 channels,
//   A[m][(s, c)] = x[row + ky(tap_s)][m + kx(tap_s)][c]      -- for a lane 8 consecutive channels of ONE pixel: a single
//                                                               ds_read_b128 from the channels-innermost halo tile,
//   B[(s, c)][n] = w[tap_s][c] if c == n else 0              -- block diagonal: 15 of 16 products are zeros, but the
//                                                               matrix pipe does 16x the vector pipe's MACs per cycle,
// 13 MFMAs per 16 px x 16 ch (25 taps in pairs) instead of ~200 vector instructions: 104 MFMAs = 1 664 matrix cycles per
// step and SIMD next to the pointwise conv's 1 024 (2 048 with the lo part).  The taps are fp16 here (MFMA operands), the
// sum is fp32 in the accumulator; LO = true carries the depthwise result to the pointwise conv as an fp16 hi + lo pair
// (two MFMAs per product, DESIGN finding 24), LO = false rounds it to fp16.
// The A addresses cost no vector instructions: the halo is stored with a column-keyed XOR swizzle (16-byte chunk c of
// halo pixel (py, px) at chunk c ^ ((px >> 1) & 7), applied on the LDS-DMA source side), so a lane's address is one
// per-tap-pair register (13 of them) plus the row as the instruction's immediate offset.
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
constexpr int SC_BT_HALF = 128 * 128;                 // 128 pixels x 64 channels fp16
constexpr int SC_BT_BYTES = 2 * SC_BT_HALF;           // hi tile, lo tile
constexpr int SC_NDW = 4, SC_NMW = 4;                 // depthwise / mma waves (one of each per SIMD)

struct SepParams {
  const half_t* in;
  int N, H, W, C, in_ld;
  const half_t* dww;     // block-diagonal tap fragments [C/64][4][NP][64 lanes][8] fp16 (sepconvm_pack_dw)
  const half_t* pww;     // pointwise weights (fp16) in MFMA-fragment order (sepconvp_pack_pw)
  const float* bias;     // [Cout] (never null: the launcher substitutes zeros)
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

// fp32 pair -> fp16 hi pair + fp16 lo pair with x == hi + lo to 21 bits: hi = rtz(x), lo = rtz(x - hi) (the residual is
// exact in fp32; v_cvt_pkrtz converts two values per instruction, the residual is one v_fma_mix / v_sub per value)
__device__ __forceinline__ void split_hi_lo(const f32x2& x, f16x2& hi, f16x2& lo) {
  typedef __fp16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
  const h2 l = __builtin_amdgcn_cvt_pkrtz(x[0] - (float)h[0], x[1] - (float)h[1]);
  hi = __builtin_bit_cast(f16x2, h);
  lo = __builtin_bit_cast(f16x2, l);
}

// MT = 16-cout MFMA row tiles per mma wave: Cout = 4 * 16 * MT (128 or 256).
//
// Loads and the register allocator.  A first build issued the tap / weight loads from inline asm and ordered their use
// with counted s_waitcnt asm statements (as the round-2 kernel does for its weights).  That is unsafe once registers are
// tight: the compiler believes an asm output is complete when the statement has executed, so it may move the value and
// hand the destination register to something else while the load is still in flight -- here a halo DMA pointer, which
// the late data then overwrote (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, one run in three).  Every load below is an
// ordinary load the compiler tracks.  The cost of that, its vmcnt(0) behind LDS-DMA traffic, is avoided by program
// order instead: the depthwise wave consumes the taps of step g (requested a step earlier) BEFORE it issues the DMA
// batch of step g+2, so the wait it gets covers only loads that are a whole step old; the mma waves issue no DMA.
template <int KS, int MT, bool HEAD, int ACT, bool LO>
__global__ void __launch_bounds__(64 * (SC_NDW + SC_NMW), 1) sepconvm_kernel(const SepParams p) {
  constexpr int NDW = SC_NDW, NMW = SC_NMW;
  constexpr int KK = KS * KS, PAD = KS / 2, NP = (KK + 1) / 2;      // NP tap pairs
  constexpr int SC_NPIX = sc_npix(KS), SC_NST = SC_NPIX / 8, SC_HALO_BYTES = sc_halo_bytes(KS);
  constexpr int NT = 64 * (NDW + NMW);      // threads
  constexpr int COUT = NMW * 16 * MT;
  constexpr int NDMA = 32 / NDW;            // LDS-DMA instructions per dw wave and step
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* const halo = lds;                                   // 3 x SC_HALO_BYTES (ring)
  char* const bt = lds + 3 * SC_HALO_BYTES;                 // 2 x SC_BT_BYTES
  float* const hwl = reinterpret_cast<float*>(bt + 2 * SC_BT_BYTES);   // HEAD: [2][Cout] head weights,
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

  if (HEAD) {
    for (int i = tid; i < p.hc * COUT; i += NT) hwl[i] = p.hw[i];
  }

  if (wave < NDW) {
    // ------------------------------------------------------------------ dw role
    // wave = 16-channel block `wave` of the step's 64-channel chunk, all 8 rows of the 8 x 16 tile;
    // lane (x, g): A fragments = pixel column x, taps 2p + (g >> 1), channels (g & 1) * 8 .. + 7 of the block
    const int x16 = lane & 15, g4 = lane >> 4;
    int LB[NP];          // LDS byte offset of the lane's A fragment of tap pair p for tile row 0 (halo slot 0)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      int t = 2 * pp + (g4 >> 1);
      if (t > KK - 1) t = KK - 1;                    // the odd tap out: its B rows are zeros
      const int ky = t / KS, kx = t - ky * KS, col = x16 + kx, cgk = wave * 2 + (g4 & 1);
      LB[pp] = (ky * SC_IW + col) * 128 + ((cgk ^ ((col >> 1) & 7)) << 4);
    }
    // Halo DMA: wave w issues slots i = w + NDW*k (the two spare slots repeat slot 29 with the same data; a step
    // without a halo to fetch copies the zero page into the free ring slot).
    // [Round 3: the DMA moved here from the mma waves.  With three MFMAs per product and one mma wave per SIMD the
    // mma wave became the pole, and DMA issue (60-185 cycles per instruction) plus the wait for the weights were
    // serial with its MFMAs: 1.52 ms per 32 x 256^2 x 320 -> 256 against 0.95 ms before.]
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
        const half_t* src = p.in + (size_t)n * p.H * p.W * p.in_ld;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
          const int i = min(wave + NDW * k, SC_NST - 1);
          const int q = i * 8 + (lane >> 3);
          const int py = q / SC_IW, px = q - py * SC_IW;
          const int iy = y0 + py - PAD, ix = x0 + px - PAD;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          // LDS chunk (lane & 7) of this halo pixel holds the pixel's 16-byte chunk (lane & 7) ^ ((px >> 1) & 7)
          sptr[k] = ok ? src + ((size_t)iy * p.W + ix) * p.in_ld + (((lane & 7) ^ ((px >> 1) & 7)) << 3) : p.zero;
        }
      }
      char* hb = halo + ringslot * SC_HALO_BYTES;
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int i = min(wave + NDW * k, SC_NST - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sptr[k],
                                         (__attribute__((address_space(3))) void*)(hb + i * 1024), 16, 0, 0);
        sptr[k] += 64;      // next chunk (zero-page pointers stay inside the 2 KiB page: <= NC + 2 increments)
      }
    };
    // block-diagonal tap fragments of this wave's channel block, requested one step before their use (two register
    // sets, as the taps of sepconv_precise.hip): [chunk][block][pair][lane][8] fp16, one 16-byte load per pair
    f16x8 w[NP], wn[NP];
    auto load_taps = [&](int ch) {
      const f16x8* b = reinterpret_cast<const f16x8*>(p.dww) + ((size_t)(ch * 4 + wave) * NP) * 64 + lane;
#pragma unroll
      for (int t = 0; t < NP; ++t) wn[t] = b[t * 64];
    };
    stage(0, 0, 0, S > 0);
    stage(0, NC > 1 ? 1 : 0, 1, S > 1);      // S > 1 implies NC >= 2 (launcher) or a second tile; see launcher
    load_taps(0);
    __syncthreads();                          // full wait; epilogue constants visible
    // step counters: c2 = (g+2) % NC with tile t2, ring = g % 3, ring2 = (g+2) % 3, chn = (g+1) % NC
    int ring = 0, chn = 0, c2 = 2 % NC, t2 = 2 / NC, ring2 = 2;
    for (int g = 0; g <= LAST; ++g) {
      chn = chn + 1 == NC ? 0 : chn + 1;
      // The taps of this step (requested at the top of the previous step) are taken over first: the wait the compiler
      // puts here covers them and every older VM operation -- this wave's share of the halo of step g+1, DMA'd a step
      // ago -- and nothing younger exists yet.  Then the taps of step g+1 and the DMA batch of step g+2 are issued;
      // both stay in flight across the step.  The barrier at the end of the step publishes the halo of step g+1.
#pragma unroll
      for (int t = 0; t < NP; ++t) w[t] = wn[t];
#pragma unroll
      for (int t = 0; t < NP; ++t) asm volatile("" : "+v"(w[t]));      // pins the wait in front of the DMA issue below
      __builtin_amdgcn_sched_barrier(0);
      load_taps(chn);      // (past the last step: a harmless re-load of an existing chunk)
      stage(t2, c2, ring2, g + 2 < S);
      __builtin_amdgcn_sched_barrier(0);
      if (g < S) {
        const char* hb = halo + ring * SC_HALO_BYTES;
        f32x4 acc[SC_TH];
#pragma unroll
        for (int r = 0; r < SC_TH; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          const char* ab = hb + LB[pp];
#pragma unroll
          for (int r = 0; r < SC_TH; ++r) {
            const f16x8 a = *reinterpret_cast<const f16x8*>(ab + r * (SC_IW * 128));
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, w[pp], acc[r], 0, 0, 0);
          }
        }
        // accumulator layout: lane (n = x16 -> channel wave*16 + n, rows 4*g4 + e -> pixel column) of tile row r.
        // B tiles (hi, lo): row = pixel (128 B of fp16), 16-byte slot s of a row stored at slot s ^ (px & 7)
        char* bb = bt + (g & 1) * SC_BT_BYTES;
        const int slot = wave * 2 + (x16 >> 3), sub = (x16 & 7) * 2;
#pragma unroll
        for (int r = 0; r < SC_TH; ++r)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int px = r * SC_TW + 4 * g4 + e;
            char* q0 = bb + px * 128 + ((slot ^ (px & 7)) << 4) + sub;
            char* q1 = bb + (px + 1) * 128 + ((slot ^ ((px + 1) & 7)) << 4) + sub;
            if (LO) {
              f16x2 h, l;
              split_hi_lo(f32x2{acc[r][e], acc[r][e + 1]}, h, l);
              *reinterpret_cast<half_t*>(q0) = h[0];
              *reinterpret_cast<half_t*>(q1) = h[1];
              *reinterpret_cast<half_t*>(q0 + SC_BT_HALF) = l[0];
              *reinterpret_cast<half_t*>(q1 + SC_BT_HALF) = l[1];
            } else {
              *reinterpret_cast<half_t*>(q0) = (half_t)acc[r][e];
              *reinterpret_cast<half_t*>(q1) = (half_t)acc[r][e + 1];
            }
          }
      }
      if (++c2 == NC) { c2 = 0; ++t2; }
      ring = ring == 2 ? 0 : ring + 1;
      ring2 = ring2 == 2 ? 0 : ring2 + 1;
      lds_barrier();
    }
    // the last (dummy) DMA batch and tap loads are drained by the end-of-kernel wait the compiler emits for wn
#pragma unroll
    for (int t = 0; t < NP; ++t) asm volatile("" :: "v"(wn[t]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ------------------------------------------------------------------ mma role
    const int wm = wave - NDW, g16 = lane >> 4, n16 = lane & 15;
    __builtin_amdgcn_s_setprio(3);      // few instructions, long latencies: issue ahead of the VALU-bound dw wave
    // Pointwise weights, pre-packed in fragment order [chunk][wave][tile][k-half][lane][8]: every load
    // instruction reads 1 KiB of whole cache lines.  The fragments of k-step 0 of the NEXT chunk are requested as soon
    // as this chunk's k-step 0 MFMAs are issued, those of k-step 1 at the end of the step: each half has half a step
    // to arrive (with all of them requested at the end of the step a full L2 round trip was exposed every step).
    // These waves issue no LDS-DMA, so the compiler's own counted waits are exact here.
    const f16x8* const abase = reinterpret_cast<const f16x8*>(p.pww) + (size_t)wm * MT * 2 * 64 + lane;
    constexpr int A_CHUNK = NMW * MT * 2 * 64;          // fragments (16 B) per 64-channel chunk
    // bias of this lane's accumulator rows: tile t, element e <-> cout wm*16*MT + (t>>1)*32 + g16*8 + (t&1)*4 + e
    const f32x4* const bbase = reinterpret_cast<const f32x4*>(p.bias + wm * 16 * MT + g16 * 8);
    f32x4 acc[MT][8];
    f16x8 ah[MT][2];                      // fragments of the chunk being multiplied
    f16x8 nah[MT][2];                     // in flight: the next chunk's
    auto load_a = [&](int ch, int ks) {     // MT loads
#pragma unroll
      for (int t = 0; t < MT; ++t) nah[t][ks] = abase[(size_t)ch * A_CHUNK + t * (2 * 64) + ks * 64];
    };
    // head finishing lanes: output idx = wm*64 + lane -> (class h = idx >> 7, pixel idx & 127)
    const int fidx = wm * 64 + lane, fh = fidx >> 7, fpx = fidx & 127;
    float fhb = 0.f;
    if (HEAD && fh < p.hc) fhb = p.hb[fh];
    load_a(0, 0);
    load_a(0, 1);
    __syncthreads();                          // epilogue constants visible
    // step counters: c1 = (g-1) % NC with tile t1
    int c1 = NC - 1, t1 = -1;
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
      const bool first = do_mma && c1 == 0;      // first chunk of a tile: the accumulators start from the bias
      if (first) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const f32x4 bv = bbase[(t >> 1) * 8 + (t & 1)];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[t][nt] = bv;
        }
      }
      const char* bb = bt + ((g - 1) & 1) * SC_BT_BYTES + n16 * 128;
      auto bfrag = [&](int i, int part) {   // i = nt*2 + ks; part 0 = hi, 1 = lo
        return *reinterpret_cast<const f16x8*>(bb + part * SC_BT_HALF + (i >> 1) * 16 * 128 + ((((i & 1) * 4 + g16) ^ (n16 & 7)) << 4));
      };
      auto half = [&](const int ks) {       // the 8 pixel tiles x MT cout tiles of k-step ks: 2 MFMAs per product
        f16x8 bh = bfrag(ks, 0), bl = bh;
        if (LO) bl = bfrag(ks, 1);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          f16x8 nh, nl;
          if (nt + 1 < 8) { nh = bfrag((nt + 1) * 2 + ks, 0); if (LO) nl = bfrag((nt + 1) * 2 + ks, 1); }
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks], bh, acc[t][nt], 0, 0, 0);
          if (LO) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks], bl, acc[t][nt], 0, 0, 0);
          }
          if (nt + 1 < 8) { bh = nh; if (LO) bl = nl; }
        }
      };
      const int cn = c1 + 1 == NC ? 0 : c1 + 1;     // chunk g % NC, multiplied in step g+1
      // k-step 0: take over the fragments requested in the middle of the previous step, multiply, request the next ones
#pragma unroll
      for (int t = 0; t < MT; ++t) ah[t][0] = nah[t][0];
      if (do_mma) half(0);
      __builtin_amdgcn_sched_barrier(0);
      load_a(cn, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MT; ++t) ah[t][1] = nah[t][1];
      if (do_mma) half(1);
      __builtin_amdgcn_sched_barrier(0);
      load_a(cn, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (do_mma && c1 == NC - 1) {
        int n, y0, x0;
        tile_coords(p, tile_of(t1), n, y0, x0);
        const int ox = x0 + n16;
        if (!HEAD) {
#pragma unroll
          for (int blk = 0; blk < MT / 2; ++blk) {
            const int cb = wm * 16 * MT + blk * 32 + g16 * 8;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
              f16x8 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)sc_act<ACT>(acc[2 * blk][nt][e]);
                o[4 + e] = (half_t)sc_act<ACT>(acc[2 * blk + 1][nt][e]);
              }
              const int oy = y0 + nt;
              if (oy < p.H && ox < p.W)
                *reinterpret_cast<f16x8*>(p.out + (((size_t)n * p.H + oy) * p.W + ox) * p.out_ld + cb) = o;
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[t][nt][e] = sc_act<ACT>(acc[t][nt][e]);
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
      lds_barrier();
    }
  }
}

// depthwise taps (KK, C) fp32 -> block-diagonal B fragments [C/64][4 blocks][NP pairs][64 lanes][8] fp16: lane (n, g) of
// pair p holds, for k = 8g .. 8g+7 <-> (tap 2p + (g >> 1), channel (g & 1) * 8 + i), the tap's weight of channel n where
// that channel IS n, zeros elsewhere (and zeros for the tap past the last one)
__global__ void __launch_bounds__(256) sepconvm_pack_dw_kernel(const float* __restrict__ w, int KS_, int C, half_t* __restrict__ out) {
  const int KK = KS_ * KS_, NP = (KK + 1) / 2;
  const int total = (C / 16) * NP * 64;      // fragments
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int lane = i & 63, pp = (i >> 6) % NP, blk = (i >> 6) / NP;      // blk = chunk * 4 + block
    const int n = lane & 15, g = lane >> 4, t = 2 * pp + (g >> 1);
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool hit = t < KK && ((g & 1) * 8 + e) == n;
      v[e] = hit ? (half_t)w[(size_t)t * C + blk * 16 + n] : (half_t)0.f;
    }
    *reinterpret_cast<f16x8*>(out + (size_t)i * 8) = v;
  }
}

template <int KS, int MT, bool HEAD, int ACT, bool LO>
int launch_act(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    EMP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&sepconvm_kernel<KS, MT, HEAD, ACT, LO>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((sepconvm_kernel<KS, MT, HEAD, ACT, LO>), dim3(grid), dim3(64 * (SC_NDW + SC_NMW)), lds_bytes, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <int KS, int MT, bool HEAD, bool LO>
int launch_one(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  if (p.act == 1) return launch_act<KS, MT, HEAD, 1, LO>(p, lds_bytes, grid, s);
  if (p.act == 2) return launch_act<KS, MT, HEAD, 2, LO>(p, lds_bytes, grid, s);
  return launch_act<KS, MT, HEAD, 0, LO>(p, lds_bytes, grid, s);
}

}  // namespace

static size_t sepconvm_lds_bytes(int Cout, int head_c, int ks = 5) {
  return 3 * sc_halo_bytes(ks) + 2 * SC_BT_BYTES + (head_c ? (size_t)2 * Cout * 4 + (size_t)SC_NMW * 128 * 2 * 4 : 0);
}

bool sepconvm_supported(int C, int Cout, int head_c) {
  // C / 64 + 2 increments of 128 B must stay inside the 2 KiB zero page
  return C % 64 == 0 && C >= 128 && C <= 512 && (Cout == 128 || Cout == 256) && head_c >= 0 && head_c <= 2 &&
         sepconvm_lds_bytes(Cout, head_c) <= 160 * 1024;
}

// pointwise weights: the fragment order of sepconv_precise.hip (same mma role)
int launch_sepconvm_pack_pw(const float* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s) {
  return launch_sepconvp_pack_pw(w, pw_ld, C, Cout, packed, s);
}

// packed: (C / 16) * ((ks*ks + 1) / 2) * 512 fp16
int launch_sepconvm_pack_dw(const float* w, int ks, int C, half_t* packed, hipStream_t s) {
  EMP_REQUIRE((ks == 3 || ks == 5) && C % 64 == 0 && C > 0, "sepconvm pack_dw: bad shape");
  const int total = (C / 16) * ((ks * ks + 1) / 2) * 64;
  hipLaunchKernelGGL(sepconvm_pack_dw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, ks, C, packed);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// out != nullptr: y = act(pw(dw(x)) + bias) -> (N,H,W,out_ld) fp16.
// head_c > 0   : hout[n][h] = head_w[h] . y + head_b[h] as fp32 planes of `plane` floats; y is not stored.
int launch_sepconvm(const half_t* in, int N, int H, int W, int C, int in_ld, const half_t* dww, const half_t* pww,
                    const float* bias, int Cout, int act, half_t* out, int out_ld, const float* head_w,
                    const float* head_b, int head_c, float* hout, int64_t plane, const half_t* zero, hipStream_t s,
                    int ks, int lo) {
  EMP_REQUIRE(ks == 5 || (ks == 3 && head_c == 0), "sepconv: depthwise kernel %d unsupported", ks);
  EMP_REQUIRE(sepconvm_supported(C, Cout, head_c), "sepconvm: unsupported shape C=%d Cout=%d head=%d", C, Cout, head_c);
  EMP_REQUIRE(act >= 0 && act <= 2, "sepconvm: bad activation %d", act);
  EMP_REQUIRE((head_c > 0) != (out != nullptr), "sepconvm: exactly one of the feature / head outputs");
  EMP_REQUIRE(in_ld % 8 == 0 && (out == nullptr || out_ld % 8 == 0), "sepconvm: 16-byte row alignment");
  SepParams p{};
  p.in = in; p.N = N; p.H = H; p.W = W; p.C = C; p.in_ld = in_ld;
  p.dww = dww; p.pww = pww;
  p.bias = bias ? bias : reinterpret_cast<const float*>(zero);      // Cout * 4 <= 1 KiB of the 2 KiB zero page
  p.out = out; p.out_ld = out_ld; p.act = act; p.zero = zero;
  p.tiles_x = cdiv(W, SC_TW); p.tiles_y = cdiv(H, SC_TH);
  const int64_t tiles = (int64_t)N * p.tiles_x * p.tiles_y;
  EMP_REQUIRE(tiles < (1ll << 30), "sepconvm: too many tiles");
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
  const size_t lds_bytes = sepconvm_lds_bytes(Cout, head_c, ks);
  if (lo) {
    if (ks == 3) return Cout == 256 ? launch_one<3, 4, false, true>(p, lds_bytes, grid, s) : launch_one<3, 2, false, true>(p, lds_bytes, grid, s);
    if (Cout == 256)
      return head_c ? launch_one<5, 4, true, true>(p, lds_bytes, grid, s) : launch_one<5, 4, false, true>(p, lds_bytes, grid, s);
    return head_c ? launch_one<5, 2, true, true>(p, lds_bytes, grid, s) : launch_one<5, 2, false, true>(p, lds_bytes, grid, s);
  }
  if (ks == 3) return Cout == 256 ? launch_one<3, 4, false, false>(p, lds_bytes, grid, s) : launch_one<3, 2, false, false>(p, lds_bytes, grid, s);
  if (Cout == 256)
    return head_c ? launch_one<5, 4, true, false>(p, lds_bytes, grid, s) : launch_one<5, 4, false, false>(p, lds_bytes, grid, s);
  return head_c ? launch_one<5, 2, true, false>(p, lds_bytes, grid, s) : launch_one<5, 2, false, false>(p, lds_bytes, grid, s);
}

}  // namespace emp
