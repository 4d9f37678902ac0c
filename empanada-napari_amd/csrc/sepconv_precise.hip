// Fused depthwise-separable convolution for gfx950 (MI355X):
//   y = act( pw( dw5x5(x) ) + bias )            [+ optional 1x1 head on y]
// replaces nn.Conv2d(C,C,5,groups=C,pad=2,bias=False) -> nn.Conv2d(C,Cout,1) -> folded BN -> ReLU
// (the 'depthwise_separable_conv' of the Panoptic-DeepLab decoder / heads, models/blocks.py:15-33, called from
// models/decoders/panoptic_deeplab.py:68-80 and models/heads.py:12-15) -- and, for the heads, the final 1x1 conv
// (Cout -> hc <= 2) so that the Cout-channel map never leaves the chip.
//
// There is no non-linearity between the depthwise and the pointwise conv, so the depthwise output is
// only ever the B operand of the pointwise GEMM: it is produced straight into LDS, in fp32.
//
// Round 3: the depthwise half of the block is exact although the block runs on the fp16 matrix pipe.  Its rounding sites
// -- the depthwise taps, the depthwise output and the pointwise weights stored as fp16 -- were a large part of the
// end-to-end error of the centre heat-map against the fp32 reference (profiles/r02_error_budget.csv).  A what-if with
// the format-emulating oracle on the two blocks the heat-map depends on (256^2 tile, ctr rms; DESIGN finding 24):
//     fp16 everywhere 8.03e-4 | fp32 taps 7.99e-4 | + depthwise output kept to 21 bits 7.32e-4 | + pointwise weights
//     as hi + lo pairs too 7.28e-4
// so the kernel keeps
//   * the depthwise taps in fp32 (they only ever meet the fp32 vector pipe: free), and
//   * the depthwise output as an fp16 hi tile + an fp16 lo tile in LDS (hi = rtz(x), lo = rtz(x - hi): 21 significant
//     bits; split ONCE by the depthwise wave that produced it -- a first build that stored fp32 and let each of the four
//     mma waves split its own B fragments spent 4x the conversions and ran at half the speed), multiplied as
//     w*hi + w*lo: two MFMAs per product,
// and leaves the pointwise weights in fp16: their hi + lo form (a third MFMA per product, twice the weight traffic, 32
// more registers) bought 0.5 % of the error of the Panoptic-DeepLab network for a third of the matrix work, and was removed.
//
// Round 4: the hi + lo pointwise weights are back as a template switch (WS) for the 128-cout blocks of the BiFPN network
// (MT = 2: the second fragment set fits the registers).  There the budget is different (tools/error_budget.py --arch
// bifpn): the six pointwise weight matrices of the FPN nodes -- each shared by four nodes -- the decoder's fusion conv and
// the centre head are 65 % of the weight-rounding variance of the centre heat-map, and with them carried as pairs the
// heat-map comes under 1e-3 of its scale (1.13e-3 -> 0.97e-3 at 512^2).  Third MFMA per product: w_lo * x_hi.
//
// One persistent workgroup per CU, 512 threads = 8 waves with two roles (one wave of each per SIMD, so
// the vector ALU and the matrix pipe of every SIMD are both fed):
//   waves 0-3  "dw":  fp32 depthwise 5x5 on the VALU (v_pk_fma_f32 over channel pairs) from a
//                     12x20-pixel x 64-channel halo tile in LDS into two 128-pixel x 64-channel fp16
//                     B tiles in LDS (hi and lo parts, XOR-swizzled rows of 128 B); taps straight from L2 into registers.
//   waves 4-7  "mma": LDS-DMA (global_load_lds) of the NEXT halo tile, then the pointwise GEMM of the
//                     PREVIOUS B tile: 16x16x32 f16 MFMA, A fragments (pointwise weights) straight
//                     from L2, accumulators (Cout/4 x 128 pixels per wave) resident across the C/64 channel
//                     chunks and initialised with the bias; epilogue act + 16-byte NHWC stores (or the head
//                     reduction).
// The pipeline is flattened over (tile, chunk) steps with ONE barrier per step, so fill and drain are
// paid once per workgroup, not per tile.  Halo tiles are DMA'd TWO steps ahead into a ring of three
// (HBM latency under load is about one step); the barrier fences LDS only (lgkmcnt), and the mma waves
// wait with a counted vmcnt so that the youngest DMA batch stays in flight across it.  [measured: with
// one-step prefetch and a full __syncthreads the mma role alone took 1.07 ms of a 1.15 ms launch.]
//
// Summation order: depthwise taps ky-major, kx-minor, fp32 fma chain from 0; GEMM ascending 32-channel K-steps,
// per step w*hi, w*lo (WS: then w_lo*hi), from the bias.  Fixed, so a batch of N equals N batch-1 calls bit for bit.
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
  const float* dww;      // [C/64][KS*KS][64] fp32 (sepconv5_pack_dw)
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
template <int KS, int MT, bool HEAD, int ACT, bool WS>
__global__ void __launch_bounds__(64 * (SC_NDW + SC_NMW), 1) sepconvp_kernel(const SepParams p) {
  constexpr int NDW = SC_NDW, NMW = SC_NMW;
  constexpr int KK = KS * KS, PAD = KS / 2;
  constexpr int SC_NPIX = sc_npix(KS), SC_NST = SC_NPIX / 8, SC_HALO_BYTES = sc_halo_bytes(KS);
  constexpr int NT = 64 * (NDW + NMW);      // threads
  constexpr int R = 16 / NDW;               // output rows per depthwise thread
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
    // thread = channel pair cp x columns 4cg..4cg+3 x rows R*rg..R*rg+R-1 of the 8x16 tile
    const int cp = lane & 31, combo = wave * 2 + (lane >> 5), cg = combo & 3, rg = combo >> 2;
    const int hoff = ((R * rg) * SC_IW + 4 * cg) * 128 + cp * 4;
    const int sw = (cp >> 2), sub = (cp & 3) * 4;
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
        const half_t* src = p.in + (size_t)n * p.H * p.W * p.in_ld + (lane & 7) * 8;
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
          const int i = min(wave + NDW * k, SC_NST - 1);
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
        const int i = min(wave + NDW * k, SC_NST - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sptr[k],
                                         (__attribute__((address_space(3))) void*)(hb + i * 1024), 16, 0, 0);
        sptr[k] += 64;      // next chunk (zero-page pointers stay inside the 2 KiB page: <= NC + 2 increments)
      }
    };
    // taps of a 64-channel chunk: [chunk][tap][64] fp32 (sepconvp_pack_dw), 8 bytes per lane and tap, straight from L2
    // into registers, requested one step before their use
    f32x2 w[KK], wn[KK];
    auto load_taps = [&](int ch) {
      const f32x2* b = reinterpret_cast<const f32x2*>(p.dww + (size_t)ch * (KK * 64)) + cp;
#pragma unroll
      for (int t = 0; t < KK; ++t) wn[t] = b[t * 32];
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
      for (int t = 0; t < KK; ++t) w[t] = wn[t];
#pragma unroll
      for (int t = 0; t < KK; ++t) asm volatile("" : "+v"(w[t]));      // pins the wait in front of the DMA issue below
      __builtin_amdgcn_sched_barrier(0);
      load_taps(chn);      // (past the last step: a harmless re-load of an existing chunk)
      stage(t2, c2, ring2, g + 2 < S);
      __builtin_amdgcn_sched_barrier(0);
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
        // B tiles (hi, lo): row = pixel (128 B of fp16), 16-byte slot s of a row stored at slot s ^ (px & 7)
        char* bb = bt + (g & 1) * SC_BT_BYTES;
#pragma unroll
        for (int y = 0; y < R; ++y)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int px = (R * rg + y) * SC_TW + 4 * cg + j;
            f16x2 h, l;
            split_hi_lo(acc[y][j], h, l);
            char* q = bb + px * 128 + ((sw ^ (px & 7)) << 4) + sub;
            *reinterpret_cast<f16x2*>(q) = h;
            *reinterpret_cast<f16x2*>(q + SC_BT_HALF) = l;
          }
      }
      if (++c2 == NC) { c2 = 0; ++t2; }
      ring = ring == 2 ? 0 : ring + 1;
      ring2 = ring2 == 2 ? 0 : ring2 + 1;
      lds_barrier();
    }
    // the last (dummy) DMA batch and tap loads are drained by the end-of-kernel wait the compiler emits for wn
#pragma unroll
    for (int t = 0; t < KK; ++t) asm volatile("" :: "v"(wn[t]));
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
    // WS: the lo halves of the weights (w = hi + lo to 2^-22), packed behind the hi fragments in the same order
    constexpr int MTL = WS ? MT : 1;
    f16x8 al[MTL][2], nal[MTL][2];
    const size_t lo_off = (size_t)NC * A_CHUNK;
    auto load_a = [&](int ch, int ks) {     // MT (2 MT) loads
#pragma unroll
      for (int t = 0; t < MT; ++t) nah[t][ks] = abase[(size_t)ch * A_CHUNK + t * (2 * 64) + ks * 64];
      if (WS) {
#pragma unroll
        for (int t = 0; t < MT; ++t) nal[t][ks] = abase[lo_off + (size_t)ch * A_CHUNK + t * (2 * 64) + ks * 64];
      }
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
        f16x8 bh = bfrag(ks, 0), bl = bfrag(ks, 1);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          f16x8 nh, nl;
          if (nt + 1 < 8) { nh = bfrag((nt + 1) * 2 + ks, 0); nl = bfrag((nt + 1) * 2 + ks, 1); }
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks], bh, acc[t][nt], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][ks], bl, acc[t][nt], 0, 0, 0);
          if (WS) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t][ks], bh, acc[t][nt], 0, 0, 0);
          }
          if (nt + 1 < 8) { bh = nh; bl = nl; }
        }
      };
      const int cn = c1 + 1 == NC ? 0 : c1 + 1;     // chunk g % NC, multiplied in step g+1
      // k-step 0: take over the fragments requested in the middle of the previous step, multiply, request the next ones
#pragma unroll
      for (int t = 0; t < MT; ++t) ah[t][0] = nah[t][0];
      if (WS) {
#pragma unroll
        for (int t = 0; t < MT; ++t) al[t][0] = nal[t][0];
      }
      if (do_mma) half(0);
      __builtin_amdgcn_sched_barrier(0);
      load_a(cn, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MT; ++t) ah[t][1] = nah[t][1];
      if (WS) {
#pragma unroll
        for (int t = 0; t < MT; ++t) al[t][1] = nal[t][1];
      }
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

// depthwise taps (KK, C) fp32 -> chunk-major [C/64][KK][64] fp32: a 64-channel chunk's taps are one 256-byte row per tap
__global__ void __launch_bounds__(256) sepconvp_pack_dw_kernel(const float* __restrict__ w, int KK, int C, float* __restrict__ out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < KK * C; i += gridDim.x * 256) {
    const int c = i & 63, t = (i >> 6) % KK, ch = (i >> 6) / KK;
    out[i] = w[(size_t)t * C + ch * 64 + c];
  }
}

// (Cout, pw_ld) fp32 row-major -> fp16 in the fragment order of the kernel above (MT = Cout / 64)
// lo != 0: the residuals fp16(w - fp16(w)) instead (the second half of a hi + lo weight pair)
__global__ void __launch_bounds__(256) sepconvp_pack_pw_kernel(const float* __restrict__ w, int pw_ld, int C, int Cout,
                                                               int MT, half_t* __restrict__ out, int lo) {
  const int total = C * Cout / 8;      // 16-byte fragments
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int r = i;
    const int lane = r & 63; r >>= 6;
    const int ks = r & 1; r >>= 1;
    const int t = r % MT; r /= MT;
    const int wm = r % SC_NMW;
    const int ch = r / SC_NMW;
    const int n16 = lane & 15, g16 = lane >> 4, tp = t & 1;
    const int co = wm * 16 * MT + (t >> 1) * 32 + ((n16 >> 2) << 3) + (tp << 2) + (n16 & 3);
    const float* src = w + (size_t)co * pw_ld + ch * 64 + ks * 32 + g16 * 8;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const half_t h = (half_t)src[e];
      v[e] = lo ? (half_t)(src[e] - (float)h) : h;
    }
    *reinterpret_cast<f16x8*>(out + (size_t)i * 8) = v;
  }
}

template <int KS, int MT, bool HEAD, int ACT, bool WS>
int launch_act(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&sepconvp_kernel<KS, MT, HEAD, ACT, WS>), 160 * 1024)) return rc;
  hipLaunchKernelGGL((sepconvp_kernel<KS, MT, HEAD, ACT, WS>), dim3(grid), dim3(64 * (SC_NDW + SC_NMW)), lds_bytes, s, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

template <int KS, int MT, bool HEAD, bool WS = false>
int launch_one(const SepParams& p, size_t lds_bytes, int grid, hipStream_t s) {
  if (p.act == 1) return launch_act<KS, MT, HEAD, 1, WS>(p, lds_bytes, grid, s);
  if (p.act == 2) return launch_act<KS, MT, HEAD, 2, WS>(p, lds_bytes, grid, s);
  return launch_act<KS, MT, HEAD, 0, WS>(p, lds_bytes, grid, s);
}

}  // namespace

static size_t sepconvp_lds_bytes(int Cout, int head_c, int ks = 5) {
  return 3 * sc_halo_bytes(ks) + 2 * SC_BT_BYTES + (head_c ? (size_t)2 * Cout * 4 + (size_t)SC_NMW * 128 * 2 * 4 : 0);
}

bool sepconvp_supported(int C, int Cout, int head_c) {
  // C / 64 + 2 increments of 128 B must stay inside the 2 KiB zero page
  return C % 64 == 0 && C >= 128 && C <= 512 && (Cout == 128 || Cout == 256) && head_c >= 0 && head_c <= 2 &&
         sepconvp_lds_bytes(Cout, head_c) <= 160 * 1024;
}

bool sepconvp_wsplit_supported(int Cout) { return Cout == 128; }      // MT = 2: the lo fragments fit the registers

// packed: C * Cout fp16; wsplit: 2 * C * Cout (the hi fragments, then the lo fragments in the same order)
int launch_sepconvp_pack_pw(const float* w, int pw_ld, int C, int Cout, half_t* packed, hipStream_t s, int wsplit) {
  EMP_REQUIRE(C % 64 == 0 && (Cout == 128 || Cout == 256) && pw_ld >= C, "sepconvp pack_pw: bad shape");
  EMP_REQUIRE(!wsplit || sepconvp_wsplit_supported(Cout), "sepconvp pack_pw: hi + lo weights need Cout == 128");
  const int total = C * Cout / 8;
  hipLaunchKernelGGL(sepconvp_pack_pw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, pw_ld, C, Cout, Cout / 64, packed, 0);
  EMP_LAUNCH_CHECK();
  if (wsplit) {
    hipLaunchKernelGGL(sepconvp_pack_pw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w, pw_ld, C, Cout, Cout / 64,
                       packed + (size_t)C * Cout, 1);
    EMP_LAUNCH_CHECK();
  }
  return EMP_OK;
}

// taps (ks*ks, C) fp32 -> the chunk-major order the kernel reads
int launch_sepconvp_pack_dw(const float* w, int ks, int C, float* packed, hipStream_t s) {
  EMP_REQUIRE((ks == 3 || ks == 5) && C % 64 == 0 && C > 0, "sepconvp pack_dw: bad shape");
  hipLaunchKernelGGL(sepconvp_pack_dw_kernel, dim3(cdiv(ks * ks * C, 256)), dim3(256), 0, s, w, ks * ks, C, packed);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// out != nullptr: y = act(pw(dw(x)) + bias) -> (N,H,W,out_ld) fp16.
// head_c > 0   : hout[n][h] = head_w[h] . y + head_b[h] as fp32 planes of `plane` floats; y is not stored.
int launch_sepconvp(const half_t* in, int N, int H, int W, int C, int in_ld, const float* dww, const half_t* pww,
                    const float* bias, int Cout, int act, half_t* out, int out_ld, const float* head_w,
                    const float* head_b, int head_c, float* hout, int64_t plane, const half_t* zero, hipStream_t s,
                    int ks, int wsplit) {
  EMP_REQUIRE(ks == 5 || (ks == 3 && head_c == 0), "sepconv: depthwise kernel %d unsupported", ks);
  EMP_REQUIRE(!wsplit || sepconvp_wsplit_supported(Cout), "sepconvp: hi + lo pointwise weights need Cout == 128");
  EMP_REQUIRE(sepconvp_supported(C, Cout, head_c), "sepconvp: unsupported shape C=%d Cout=%d head=%d", C, Cout, head_c);
  EMP_REQUIRE(act >= 0 && act <= 2, "sepconvp: bad activation %d", act);
  EMP_REQUIRE((head_c > 0) != (out != nullptr), "sepconvp: exactly one of the feature / head outputs");
  EMP_REQUIRE(in_ld % 8 == 0 && (out == nullptr || out_ld % 8 == 0), "sepconvp: 16-byte row alignment");
  SepParams p{};
  p.in = in; p.N = N; p.H = H; p.W = W; p.C = C; p.in_ld = in_ld;
  p.dww = dww; p.pww = pww;
  p.bias = bias ? bias : reinterpret_cast<const float*>(zero);      // Cout * 4 <= 1 KiB of the 2 KiB zero page
  p.out = out; p.out_ld = out_ld; p.act = act; p.zero = zero;
  p.tiles_x = cdiv(W, SC_TW); p.tiles_y = cdiv(H, SC_TH);
  const int64_t tiles = (int64_t)N * p.tiles_x * p.tiles_y;
  EMP_REQUIRE(tiles < (1ll << 30), "sepconvp: too many tiles");
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
  const size_t lds_bytes = sepconvp_lds_bytes(Cout, head_c, ks);
  if (ks == 3) {     // BiFPN nodes (depthwise 3x3 -> pointwise -> BN -> SiLU); no head mode
    if (Cout == 256) return launch_one<3, 4, false>(p, lds_bytes, grid, s);
    return wsplit ? launch_one<3, 2, false, true>(p, lds_bytes, grid, s) : launch_one<3, 2, false>(p, lds_bytes, grid, s);
  }
  if (Cout == 256)
    return head_c ? launch_one<5, 4, true>(p, lds_bytes, grid, s) : launch_one<5, 4, false>(p, lds_bytes, grid, s);
  if (wsplit) return head_c ? launch_one<5, 2, true, true>(p, lds_bytes, grid, s) : launch_one<5, 2, false, true>(p, lds_bytes, grid, s);
  return head_c ? launch_one<5, 2, true>(p, lds_bytes, grid, s) : launch_one<5, 2, false>(p, lds_bytes, grid, s);
}

}  // namespace emp
