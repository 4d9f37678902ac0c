// 256 x 256 implicit-GEMM convolution tile for the MFMA-bound layers (K = KH*KW*Cin >= 512, Cout % 256 == 0):
// NHWC fp16, MFMA 16x16x32 f16, fp32 accumulate, same operands / epilogue semantics as conv_igemm.hip.
//
// Why a second tile: with 128 x 128 x 64 tiles every byte of both operands crosses the CU's vector-memory
// path (64 B/clk) once per 64 FLOP/B x ... = exactly the MFMA peak, i.e. LDS-DMA issue, LDS and the matrix pipe
// all saturate together and the kernel sits at ~40 % of peak (profiles/r01_layer_roofline.csv).  A 256 x 256
// tile halves the operand bytes (and LDS-DMA instructions) per MFMA.
//
// Structure (one workgroup = 8 waves = 2 groups of 4, one workgroup per CU, 128 KiB LDS):
//   * K is walked in slabs of 32 channels ("K-tile": 256 pixel rows + 256 cout rows of 64 B = 32 KiB) through
//     a ring of FOUR LDS slots filled by LDS-DMA three K-tiles ahead; a wave waits for its own DMA pieces with a
//     counted vmcnt(8) (two younger K-tiles stay in flight) and the barriers fence LDS only.
//   * wave (g, c): pixels g*128..+127 x couts c*64..+63, accumulators 4 x 8 MFMA tiles = 128 VGPRs.
//   * the two groups run half a K-tile apart: between two barriers one group issues its 12 fragment reads
//     and 4 DMA pieces while the other (its SIMD partner) runs its 32 MFMAs, then they swap.
//     slot 2t  : group 0 LOAD(t) + STAGE(t+3)   | group 1 COMPUTE(t-1)
//     slot 2t+1: group 0 COMPUTE(t)             | group 1 LOAD(t) + STAGE(t+3)
//   Hazards: K-tile t is read in slots 2t / 2t+1; its pieces were waited for (vmcnt) by their issuers in LOAD(t-1),
//   i.e. before the barrier that opens slot 2t.  STAGE(t+3) overwrites the ring slot of K-tile t-1, last read
//   (lgkmcnt(0) before the closing barrier) in slot 2t-1.
//   Measured (ASPP 3x3, 2048 -> 256 channels, 32 x 64 x 64 pixels, s_memtime stamps in an experiment build):
//   per K-tile and wave 450 cycles fragment reads + address work, 233 wait + barrier, 580 MFMA + DMA issue,
//   90 barrier = two slots of ~680 cycles against 512 cycles of MFMA issue; the chip holds ~1.45 GHz under this
//   load, so 1.24-1.29 PFLOP/s here is ~80 % of the clock-adjusted matrix peak (128 x 128 tile: 0.96-1.0).
//   Ablations: no DMA 0.73 ms, no MFMA 0.75 ms, no fragment reads 0.80 ms vs 0.99 ms complete.
//   Tried in round 2 and removed (tools/conv256_ab.py, same process, random data, bit-identical results):
//   (a) K-walk groups of 128 / 64 channels instead of 256 (less L2 spill of the dilated taps): 0.960 -> 0.969 / 0.991 ms;
//   (b) 2 or all 4 LDS-DMA pieces issued in the LOAD slot instead of in the MFMAs' shadow (template DMA_EARLY, kept):
//       0.960 vs 0.965 / 0.963 ms;
//   (c) a "pair" schedule -- a slot covers TWO K-tiles, ring of five slots (all 160 KiB), the second tile's fragments
//       read in the shadow of the first tile's MFMAs, COMPUTE slots of 64 back-to-back MFMAs, two barriers per 64
//       channels instead of four (238 VGPRs): 0.998 vs 1.012 ms (ASPP), 0.517 vs 0.521 (layer4 3x3), 0.282 vs 0.289
//       (2048 -> 512 1x1).  Halving the barriers and removing the per-slot pipe bubbles buys nothing: with the effective
//       clock at 1.35-1.7 GHz under this load (profiles/r02_conv256_pmc.txt, GRBM_GUI_ACTIVE) the kernel sits at what the
//       chip sustains, not at an issue-slot limit -- MI355X_MICROARCH.md "DVFS give-back" (3), and the guide's own best
//       256^2 template measures the same 1.32 PFLOP/s on random operands.
//   * LDS rows are 64 B; 16-byte chunk c of row r is stored at chunk c ^ ((-(r >> 2)) & 3): conflict-free for the
//     ds_read_b128 lane groups of gfx950; the swizzle is applied to the DMA source address (lane-linear dest).
#include "common.h"

namespace emp {

namespace {

constexpr int KS = 32;                    // channels per K-tile
constexpr int SLOT_BYTES = 512 * 64;      // 256 pixel rows + 256 cout rows
constexpr int W_OFF = 256 * 64;           // cout rows start
constexpr int NSLOT = 4;

__device__ __forceinline__ int perm32b(int x) {
  const int t = x >> 4, i = x & 15;
  return ((i >> 2) << 3) + (t << 2) + (i & 3);
}
__device__ __forceinline__ void lds_barrier() {
  // Raw barrier that orders LDS traffic only.  A fence (or __syncthreads) would also drain vmcnt: an in-flight
  // LDS-DMA is a pending LDS write on the VM counter.  The "memory" clobber keeps the compiler from moving
  // memory accesses across it; DMA completion is handled by the counted vmcnt waits of the issuing waves.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Epilogue of a 128-pixel x 64-cout wave tile straight from the accumulators: lane (fq, fr) owns couts
// co0 + pair*32 + fq*8 + [0,8) of pixel mrow0 + q*16 + fr.  ACT is a compile-time constant (a run-time switch inside
// the unrolled loops costs ~6 scalar branches per ELEMENT: 850 s_cbranch, ~10 us per 256 x 256 tile [measured]); the
// bias is read once and the 16 residual fragments are requested up front (the K-loop's fragment registers are free).
template <int ACT>
__device__ __forceinline__ float act_c(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}
template <int ACT>
__device__ __forceinline__ void wave_epilogue(const ConvParams& p, f32x4 (&acc)[4][8], int mrow0, int co0, int fr, int fq,
                                              int HoWo) {
  float bv[2][8];
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int co = co0 + P * 32 + fq * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[P][r] = 0.f;
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
      bv[P][0] = b0.x; bv[P][1] = b0.y; bv[P][2] = b0.z; bv[P][3] = b0.w;
      bv[P][4] = b1.x; bv[P][5] = b1.y; bv[P][6] = b1.z; bv[P][7] = b1.w;
    }
  }
  f16x8 rv[8][2];
  if (p.res) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int m = mrow0 + q * 16 + fr;
      const half_t* rp = m < p.M ? p.res + (size_t)m * p.res_ld + co0 + fq * 8 : p.zero;
      rv[q][0] = *reinterpret_cast<const f16x8*>(rp);
      rv[q][1] = *reinterpret_cast<const f16x8*>(m < p.M ? rp + 32 : p.zero);
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int m = mrow0 + q * 16 + fr;
    if (m >= p.M) continue;
    // second destination (ConvParams::out2): whole cout tiles at or past `split` (a multiple of 256) go there
    half_t* orow = (p.out2 && co0 >= p.split) ? p.out2 + (size_t)m * p.out2_ld + (co0 - p.split) + fq * 8
                                              : p.out + (size_t)m * p.out_ld + co0 + fq * 8;
    const float* bn = p.bias_n ? p.bias_n + (size_t)(m / HoWo) * p.Cout + co0 + fq * 8 : nullptr;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * P][q][r] + bv[P][r];
        v[4 + r] = acc[2 * P + 1][q][r] + bv[P][4 + r];
      }
      if (bn) {
        const float4 b0 = *reinterpret_cast<const float4*>(bn + P * 32);
        const float4 b1 = *reinterpret_cast<const float4*>(bn + P * 32 + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[q][P][r];
      }
      f16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (half_t)act_c<ACT>(v[r]);
      *reinterpret_cast<f16x8*>(orow + P * 32) = o;
    }
  }
}
// Fused epilogue + back-to-back 1x1 conv (ConvParams::next_*): the finished fp16 tile (256 pixels x 256 channels =
// 128 KiB) goes to HBM as usual AND into the (now idle) LDS ring as rows of 512 B -- 16-byte chunk c of pixel row r at
// chunk c ^ (r & 15) -- from where wave w multiplies pixel rows [32w, 32w+32) with the next conv's weights (A fragments
// straight from L2, rows permuted like the main kernel's so that a lane ends up with 8 consecutive couts): K ascends in
// 32-channel steps from a zero accumulator, i.e. the result is bit-identical to the separate launch.
template <int ACT>
__device__ __forceinline__ void wave_epilogue_b2b(const ConvParams& p, f32x4 (&acc)[4][8], char* lds, int m0, int grp, int wc,
                                                  int wave, int fr, int fq, int HoWo) {
  const int mrow0 = m0 + grp * 128, co0 = wc * 64;
  float bv[2][8];
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int co = co0 + P * 32 + fq * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[P][r] = 0.f;
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
      bv[P][0] = b0.x; bv[P][1] = b0.y; bv[P][2] = b0.z; bv[P][3] = b0.w;
      bv[P][4] = b1.x; bv[P][5] = b1.y; bv[P][6] = b1.z; bv[P][7] = b1.w;
    }
  }
  f16x8 rv[8][2];
  if (p.res) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int m = mrow0 + q * 16 + fr;
      const half_t* rp = m < p.M ? p.res + (size_t)m * p.res_ld + co0 + fq * 8 : p.zero;
      rv[q][0] = *reinterpret_cast<const f16x8*>(rp);
      rv[q][1] = *reinterpret_cast<const f16x8*>(m < p.M ? rp + 32 : p.zero);
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = grp * 128 + q * 16 + fr;
    const int m = m0 + row;
    half_t* orow = p.out + (size_t)m * p.out_ld + co0 + fq * 8;
#pragma unroll
    for (int P = 0; P < 2; ++P) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * P][q][r] + bv[P][r];
        v[4 + r] = acc[2 * P + 1][q][r] + bv[P][4 + r];
      }
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[q][P][r];
      }
      f16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (half_t)act_c<ACT>(v[r]);
      if (m < p.M) *reinterpret_cast<f16x8*>(orow + P * 32) = o;
      const int chunk = wc * 8 + P * 4 + fq;
      *reinterpret_cast<f16x8*>(lds + row * 512 + ((chunk ^ (row & 15)) << 4)) = o;
    }
  }
  lds_barrier();
  // ---- second GEMM: next_out[32 pixels of this wave][64 couts]; both operands from LDS (the next conv's weights were
  // LDS-DMA'd behind the ring at kernel entry: a global load here would sit behind the tile's 16 stores on the in-order
  // vector-memory counter, and holding 32 fragments in registers next to the accumulators spills) ----
  const char* w2 = lds + NSLOT * SLOT_BYTES;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    f32x4 a2[2][2];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int t = 0; t < 2; ++t) a2[tp][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r0 = pp * 32 + perm32b(fr), r1 = pp * 32 + perm32b(16 + fr);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const f16x8 wa = *reinterpret_cast<const f16x8*>(w2 + r0 * 512 + (((ks * 4 + fq) ^ (r0 & 15)) << 4));
      const f16x8 wb = *reinterpret_cast<const f16x8*>(w2 + r1 * 512 + (((ks * 4 + fq) ^ (r1 & 15)) << 4));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = wave * 32 + t * 16 + fr;
        const f16x8 yb = *reinterpret_cast<const f16x8*>(lds + row * 512 + (((ks * 4 + fq) ^ (row & 15)) << 4));
        a2[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, yb, a2[0][t], 0, 0, 0);
        a2[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, yb, a2[1][t], 0, 0, 0);
      }
    }
    const int co2 = pp * 32 + fq * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(p.next_b + co2);
    const float4 b1 = *reinterpret_cast<const float4*>(p.next_b + co2 + 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = m0 + wave * 32 + t * 16 + fr;
      if (m >= p.M) continue;
      f16x8 o;
      o[0] = (half_t)fmaxf(a2[0][t][0] + b0.x, 0.f); o[1] = (half_t)fmaxf(a2[0][t][1] + b0.y, 0.f);
      o[2] = (half_t)fmaxf(a2[0][t][2] + b0.z, 0.f); o[3] = (half_t)fmaxf(a2[0][t][3] + b0.w, 0.f);
      o[4] = (half_t)fmaxf(a2[1][t][0] + b1.x, 0.f); o[5] = (half_t)fmaxf(a2[1][t][1] + b1.y, 0.f);
      o[6] = (half_t)fmaxf(a2[1][t][2] + b1.z, 0.f); o[7] = (half_t)fmaxf(a2[1][t][3] + b1.w, 0.f);
      *reinterpret_cast<f16x8*>(p.next_out + (size_t)m * p.next_ld + co2) = o;
    }
  }
}
__device__ __forceinline__ void wave_epilogue_any(const ConvParams& p, f32x4 (&acc)[4][8], int mrow0, int co0, int fr,
                                                  int fq, int HoWo) {
  if (p.act == 1) wave_epilogue<1>(p, acc, mrow0, co0, fr, fq, HoWo);
  else if (p.act == 2) wave_epilogue<2>(p, acc, mrow0, co0, fr, fq, HoWo);
  else wave_epilogue<0>(p, acc, mrow0, co0, fr, fq, HoWo);
}

// DMA_EARLY: how many of the 4 LDS-DMA pieces of K-tile t+3 a wave issues in its LOAD slot (before the wait), the
// rest go out in the shadow of its MFMAs
#ifdef EMP_CLOCK_STAMP
// DIAGNOSTIC BUILD ONLY (tools/conv256_clock.py; the product library is built without this macro and executes no stamp):
// wave 0 of every workgroup stamps the shader clock (s_memtime) and the 100 MHz reference (s_memrealtime) around its K
// loop; their quotient is the clock the chip holds INSIDE the loop (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ unsigned long long emp_clock_stamp_buf[2 * 8192];
#endif

template <int DMA_EARLY, bool B2B>
__global__ void __launch_bounds__(512, 1) conv_igemm256_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
#ifdef EMP_CLOCK_STAMP
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  int mtile = xcd * p.mt_per_xcd + j / p.nt;
  int ntile = j % p.nt;
  if (p.ngroup > 0) {
    // intermediate raster (round 6): the XCD keeps `ngroup` cout tiles resident and walks ALL its pixel
    // tiles under them, then the next group -- the weights (ngroup x K x 512 B) survive the rounds' output in L2, the
    // activations are re-read nt / ngroup times instead of the weights mt_per_xcd times
    const int per = p.mt_per_xcd * p.ngroup, g = j / per, rem = j - g * per;
    mtile = xcd * p.mt_per_xcd + rem / p.ngroup;
    ntile = g * p.ngroup + rem % p.ngroup;
  }
  if (mtile >= p.mt) return;
  const int m0 = mtile * 256, n0 = ntile * 256;

  const int tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases stay scalar (M0)
  const int grp = wave >> 2, wc = wave & 3;
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / KS;              // 32-channel slabs per tap
  // kgroup < 0: ConvParams::wgt is this tile's PACKED weight image (pack256_kernel below, written once per network by
  // pdl_net.hip): the 1 KiB piece a DMA instruction moves -- 16 cout rows x 64 B of one K-tile -- is contiguous there, in
  // K-walk order, so the instruction asks L2 for eight whole 128-byte lines instead of sixteen half lines (round 5:
  // TCP_TCC_READ_REQ 143.7 M of 64 B per ASPP launch, the texture-data path 94 % busy: profiles/r05_conv256_requests.txt)
  const bool wpacked = p.kgroup < 0;
  const int KG = wpacked ? -p.kgroup : p.kgroup;      // slabs per K-walk group (divides CB)
  const int KT1 = KT * CB;                // K-tiles of the main source
  const int KTOT = KT1 + (p.in2 ? p.Cin2 / KS : 0);
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  // ---- staging: wave w moves pixel pieces {w, w+8} and cout pieces {w, w+8} (16 rows x 64 B each) ----
  const int srow = l >> 2;                                   // row within the piece
  const int schunk = (l & 3) ^ ((-(l >> 4)) & 3);            // logical 16-B chunk this lane fetches
  const half_t* a_img[2];
  int a_iy0[2], a_ix0[2];
  const half_t* a_cur[2];
  int a_inc[2];
  const half_t* a_two[2];                 // second source (ConvParams::in2): the row's pixel there, or the zero page
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (wave + 8 * i) * 16 + srow;
    a_img[i] = p.in;
    a_iy0[i] = a_ix0[i] = -(1 << 28);
    a_cur[i] = p.zero;
    a_inc[i] = 0;
    a_two[i] = p.zero;
    if (m < p.M) {
      if (p.in2) {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_two[i] = p.in2 + (((size_t)n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_ld + schunk * 8;
      }
      if (pointwise) {
        a_cur[i] = p.in + (size_t)m * p.in_ld + schunk * 8;
        a_inc[i] = KS;
      } else {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = p.in + (size_t)n * p.H * p.W * p.in_ld + schunk * 8;
      }
    }
  }
  const half_t* b_base[2];
  const half_t* b_cur[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 8 * i) * 16 + srow;
    const int co = n0 + (row & ~31) + perm32b(row & 31);
    b_base[i] = b_cur[i] = p.wgt + (size_t)co * (KT * p.Cin + (p.in2 ? p.Cin2 : 0)) + schunk * 8;
    if (wpacked) b_cur[i] = p.wgt + (((size_t)ntile * KTOT * 16 + (wave + 8 * i)) * 512 + l * 8);
  }
  const int b_step = wpacked ? 16 * 512 : KS;      // halfs per K-tile: the packed image holds the 16 pieces of a K-tile side by side
  int st_ky = 0, st_kx = 0, st_cb = 0, st_grp = 0, st_u = 0;
  const half_t* src[4];      // sources of the next K-tile's 4 pieces (2 pixel, 2 cout)
  auto stage_prep = [&]() {   // address work of K-tile st_u (kept out of the MFMA slot)
    if (st_u == KT1 && p.in2) {      // the main source is exhausted: continue along K in the second one
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_cur[i] = a_two[i];
        a_inc[i] = (a_two[i] != p.zero) ? KS : 0;
      }
    }
    if (st_cb == 0 && st_u < KT1) {
      const int c0 = st_grp * KG * KS;
      if (!pointwise) {
        const int dy = st_ky * p.dil, dx = st_kx * p.dil;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          a_cur[i] = ok ? a_img[i] + ((size_t)iy * p.W + ix) * p.in_ld + c0 : p.zero;
          a_inc[i] = ok ? KS : 0;
        }
      }
      if (KG != CB && !wpacked) {
        const int koff = (st_ky * p.KW + st_kx) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) b_cur[i] = b_base[i] + koff;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      src[i] = a_cur[i];
      a_cur[i] += a_inc[i];
      src[2 + i] = b_cur[i];
      b_cur[i] += b_step;
    }
    if (++st_cb == KG) {
      st_cb = 0;
      if (++st_kx == p.KW) {
        st_kx = 0;
        if (++st_ky == p.KH) { st_ky = 0; ++st_grp; }
      }
    }
  };
  auto dma = [&](int i) {     // piece i of K-tile st_u into ring slot st_u & 3
    char* dst = lds + (st_u & (NSLOT - 1)) * SLOT_BYTES + (i >> 1) * W_OFF + (wave + 8 * (i & 1)) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- fragment addressing (lane-constant) ----
  const int fr = l & 15, fq = l >> 4;
  const int foff = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  const int p_off = grp * 128 * 64 + foff;            // + pt*1024
  const int w_off = W_OFF + wc * 64 * 64 + foff;      // + ct*1024

  f32x4 acc[4][8];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (B2B) {
    // the next conv's weights (64 couts x 256 channels = 32 KiB) behind the ring: 4 LDS-DMA pieces per wave, rows of
    // 512 B, 16-byte chunk c of row r at chunk c ^ (r & 15); oldest on the vector-memory counter, so every later
    // counted wait covers them
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;                    // 1 KiB = rows 2*piece, 2*piece + 1
      const int r = piece * 2 + (l >> 5), cdst = l & 31;
      const half_t* srcw = p.next_w + (size_t)r * 256 + ((cdst ^ (r & 15)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcw,
                                       (__attribute__((address_space(3))) void*)(lds + NSLOT * SLOT_BYTES + piece * 1024), 16, 0, 0);
    }
  }
  f16x8 pf[8], wf[4];
  if (KTOT < 4) {
    // short reduction (ResNet layer1's 64 -> 256 conv3 with a fused next conv): everything staged at once, no pipeline
#pragma unroll 1
    for (int u = 0; u < KTOT; ++u) {
      stage_prep();
#pragma unroll
      for (int i = 0; i < 4; ++i) dma(i);
      ++st_u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
#pragma unroll 1
    for (int t = 0; t < KTOT; ++t) {
      const char* base = lds + t * SLOT_BYTES;
#pragma unroll
      for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const f16x8*>(base + w_off + c * 1024);
#pragma unroll
      for (int q = 0; q < 8; ++q) pf[q] = *reinterpret_cast<const f16x8*>(base + p_off + q * 1024);
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[q], acc[c][q], 0, 0, 0);
    }
    lds_barrier();      // every wave is done reading the ring
  } else {
  // ---- prologue: three K-tiles in flight, the first one landed ----
#pragma unroll 1
  for (int u = 0; u < 3; ++u) {
    stage_prep();
#pragma unroll
    for (int i = 0; i < 4; ++i) dma(i);
    ++st_u;
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  lds_barrier();
  if (grp == 1) lds_barrier();      // group 1 runs one slot behind group 0

  auto load_frags = [&](int t) {
    const char* base = lds + (t & (NSLOT - 1)) * SLOT_BYTES;
#pragma unroll
    for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const f16x8*>(base + w_off + c * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) pf[q] = *reinterpret_cast<const f16x8*>(base + p_off + q * 1024);
  };
  // main loop: K-tiles whose look-ahead tile t+3 exists
  for (int t = 0; t + 3 < KTOT; ++t) {
    // ---- LOAD slot: fragments of K-tile t, addresses of K-tile t+3 ----
    load_frags(t);
    stage_prep();
#pragma unroll
    for (int i = 0; i < DMA_EARLY; ++i) dma(i);
    // my pieces of K-tile t+1 have landed (the 4 pieces of t+2 may be in flight, and the DMA_EARLY pieces of t+3
    // just issued; the rest of t+3 is issued in the MFMA slot)
    if (DMA_EARLY == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    if (DMA_EARLY == 2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    if (DMA_EARLY == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    // ---- COMPUTE slot: 32 MFMAs with the 4 DMA pieces of K-tile t+3 issued in their shadow ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[q], acc[c][q], 0, 0, 0);
      if ((q & 1) && (q >> 1) >= DMA_EARLY) {
        __builtin_amdgcn_sched_barrier(0);
        dma(q >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    ++st_u;
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }
  // tail: the last three K-tiles, nothing left to stage
#pragma unroll 1
  for (int t = KTOT - 3; t < KTOT; ++t) {
    load_frags(t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[q], acc[c][q], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }
  if (grp == 0) lds_barrier();      // equal barrier count for both groups
  }

#ifdef EMP_CLOCK_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    emp_clock_stamp_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_c0;
    emp_clock_stamp_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  }
#endif
  if constexpr (B2B) {      // n0 == 0: the launcher guarantees Cout == 256
    if (p.act == 1) wave_epilogue_b2b<1>(p, acc, lds, m0, grp, wc, wave, fr, fq, HoWo);
    else wave_epilogue_b2b<0>(p, acc, lds, m0, grp, wc, wave, fr, fq, HoWo);
  } else {
    wave_epilogue_any(p, acc, m0 + grp * 128, n0 + wc * 64, fr, fq, HoWo);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same tile with WHOLE-LINE pixel fetches (round 5).  PMC on the kernel above (profiles/r05_conv256_requests.txt): its
// LDS-DMA pieces are 16 rows x 64 B -- half a 128-byte line per row, the other half belongs to the next K-tile -- so L2 sees
// 144 M requests of 64 B per ASPP launch and the texture-data return path is 94 % busy; the same bytes asked for as whole lines
// (a timing-only experiment) took 3-10 % less.  The weights get there by a packed image (pack256_kernel); the pixels by
// K-tile PAIRS: a pixel piece is 8 rows x 128 B (64 channels), a pixel slot holds 256 rows of 128 B = the two K-tiles of a
// pair side by side, and the K-tile's half is picked by the fragment address.
//   LDS (160 KiB, all of it): cout ring  4 x 16 KiB, K-tile t in slot t & 3, rows of 64 B as above, filled 3 K-tiles ahead;
//                             pixel ring 3 x 32 KiB, pair u in slot u % 3, rows of 128 B, filled two pairs ahead.
//   128-byte rows: logical 16-byte chunk c (0..7: K-tile half = c / 4) of row r lives at chunk c ^ ((r >> 1) & 7) -- with
//   lane (fr, fq) reading row fr, chunk 4 h + fq, each of gfx950's ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...)
//   touches 8 chunk positions in even rows and the same 8 in odd rows = 64 distinct banks.
//   Schedule per wave: K-tile t issues its 2 cout pieces of K-tile t + 3 first and, for even t = 2u, the 4 pixel pieces of
//   pair u + 2 behind them, all in the shadow of the MFMAs.  In LOAD(T) the wave needs its cout pieces of K-tile T + 1 (issued
//   first in K-tile T - 2) and, for odd T, its pixel pieces of pair (T + 1) / 2 (issued in K-tile T - 3): everything younger
//   is the 4 pixel pieces of whichever of T - 2, T - 1 is even plus the 2 cout pieces of T - 1 = vmcnt(6), always.
//   Same K order, same fragments, same epilogue: bit-identical to the kernel above (tests/test_gpu_conv_packed.py).
constexpr int WW_SLOT = 256 * 64;                 // cout slot
constexpr int WP_SLOT = 256 * 128;                // pixel slot (a K-tile pair)
constexpr int WP_BASE = 4 * WW_SLOT;
constexpr int WIDE_LDS_BYTES = WP_BASE + 3 * WP_SLOT;      // 163840

__global__ void __launch_bounds__(512, 1) conv_igemm256w_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  int mtile = xcd * p.mt_per_xcd + j / p.nt;
  int ntile = j % p.nt;
  if (p.ngroup > 0) {
    // intermediate raster (round 6): the XCD keeps `ngroup` cout tiles resident and walks ALL its pixel
    // tiles under them, then the next group -- the weights (ngroup x K x 512 B) survive the rounds' output in L2, the
    // activations are re-read nt / ngroup times instead of the weights mt_per_xcd times
    const int per = p.mt_per_xcd * p.ngroup, g = j / per, rem = j - g * per;
    mtile = xcd * p.mt_per_xcd + rem / p.ngroup;
    ntile = g * p.ngroup + rem % p.ngroup;
  }
  if (mtile >= p.mt) return;
  const int m0 = mtile * 256, n0 = ntile * 256;

  const int tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / KS;
  const bool wpacked = p.kgroup < 0;
  const int KG = wpacked ? -p.kgroup : p.kgroup;      // even (launcher)
  const int KT1 = KT * CB;
  const int KTOT = KT1 + (p.in2 ? p.Cin2 / KS : 0);   // even, >= 8 (launcher)
  const int KP1 = KT1 >> 1;                           // pairs of the main source
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  // ---- cout staging: as the kernel above (pieces {w, w + 8} of 16 rows x 64 B per K-tile) ----
  const half_t* b_base[2];
  const half_t* b_cur[2];
  {
    const int srow = l >> 2, schunk = (l & 3) ^ ((-(l >> 4)) & 3);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave + 8 * i) * 16 + srow;
      const int co = n0 + (row & ~31) + perm32b(row & 31);
      b_base[i] = b_cur[i] = p.wgt + (size_t)co * (KT * p.Cin + (p.in2 ? p.Cin2 : 0)) + schunk * 8;
      if (wpacked) b_cur[i] = p.wgt + (((size_t)ntile * KTOT * 16 + (wave + 8 * i)) * 512 + l * 8);
    }
  }
  const int b_step = wpacked ? 16 * 512 : KS;
  int w_t = 0, w_ky = 0, w_kx = 0, w_cb = 0, w_grp = 0;      // K-tile the next cout pieces belong to, and its place in the walk
  auto w_prep = [&]() {
    if (!wpacked && KG != CB && w_cb == 0 && w_t < KT1) {
      const int koff = (w_ky * p.KW + w_kx) * p.Cin + w_grp * KG * KS;
#pragma unroll
      for (int i = 0; i < 2; ++i) b_cur[i] = b_base[i] + koff;
    }
    if (++w_cb == KG) {
      w_cb = 0;
      if (++w_kx == p.KW) {
        w_kx = 0;
        if (++w_ky == p.KH) { w_ky = 0; ++w_grp; }
      }
    }
  };
  auto dma_w = [&](int i) {      // cout piece i of K-tile w_t
    char* dst = lds + (w_t & 3) * WW_SLOT + (wave + 8 * i) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[i],
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    b_cur[i] += b_step;
  };

  // ---- pixel staging: pieces {w, w + 8, w + 16, w + 24} of 8 rows x 128 B per K-tile PAIR ----
  const int prow = l >> 3;                                             // row within the piece
  const int pchunk = (l & 7) ^ (((wave & 1) << 2) | (prow >> 1));      // logical chunk this lane fetches: position ^ ((row >> 1) & 7)
  int a_iy0[4], a_ix0[4], a_pix[4];
  const half_t* a_cur[4];
  int a_inc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (wave + 8 * i) * 8 + prow;
    a_iy0[i] = a_ix0[i] = -(1 << 28);
    a_pix[i] = 0;
    a_cur[i] = p.zero;
    a_inc[i] = 0;
    if (m < p.M) {
      if (pointwise) {
        a_cur[i] = p.in + (size_t)m * p.in_ld + pchunk * 8;
        a_inc[i] = 2 * KS;
      } else {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_pix[i] = n * p.H * p.W;
      }
    }
  }
  int p_u = 0, p_ky = 0, p_kx = 0, p_cb = 0, p_grp = 0, p_slot = 0;      // pair the next pixel pieces belong to
  const int KG2 = KG >> 1;
  auto p_prep = [&]() {
    if (p_u == KP1 && p.in2) {      // the main source is exhausted: the walk continues in the second one (its pixel at stride2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wave + 8 * i) * 8 + prow;
        a_cur[i] = p.zero;
        a_inc[i] = 0;
        if (m < p.M) {
          const int n = m / HoWo;
          const int r = m - n * HoWo;
          const int oy = r / p.Wo;
          const int ox = r - oy * p.Wo;
          a_cur[i] = p.in2 + (((size_t)n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_ld + pchunk * 8;
          a_inc[i] = 2 * KS;
        }
      }
    }
    if (p_cb == 0 && p_u < KP1 && !pointwise) {
      const int c0 = p_grp * KG * KS + pchunk * 8;
      const int dy = p_ky * p.dil, dx = p_kx * p.dil;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        a_cur[i] = ok ? p.in + ((size_t)(a_pix[i] + iy * p.W + ix) * p.in_ld + c0) : p.zero;
        a_inc[i] = ok ? 2 * KS : 0;
      }
    }
    if (++p_cb == KG2) {
      p_cb = 0;
      if (++p_kx == p.KW) {
        p_kx = 0;
        if (++p_ky == p.KH) { p_ky = 0; ++p_grp; }
      }
    }
  };
  auto dma_p = [&](int i) {      // pixel piece i of pair p_u into pixel slot p_slot
    char* dst = lds + WP_BASE + p_slot * WP_SLOT + (wave + 8 * i) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[i],
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    a_cur[i] += a_inc[i];
  };
  auto p_next = [&]() { ++p_u; p_slot = (p_slot == 2) ? 0 : p_slot + 1; };

  // ---- fragment addressing (lane-constant) ----
  const int fr = l & 15, fq = l >> 4;
  const int w_off = wc * 64 * 64 + fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);                    // + ct * 1024
  const int p_off = WP_BASE + grp * 128 * 128 + fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);       // + q * 2048; ^ 64 for the pair's second K-tile

  f32x4 acc[4][8];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 pf[8], wf[4];

  // ---- prologue: pairs 0, 1 and K-tiles 0, 1, 2 in flight (in the order the steady state's counted wait assumes) ----
  p_prep();
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_p(i);
  p_next();
  w_prep();
#pragma unroll
  for (int i = 0; i < 2; ++i) dma_w(i);
  ++w_t;
  w_prep();
#pragma unroll
  for (int i = 0; i < 2; ++i) dma_w(i);
  ++w_t;
  p_prep();
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_p(i);
  p_next();
  w_prep();
#pragma unroll
  for (int i = 0; i < 2; ++i) dma_w(i);
  ++w_t;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // pair 0 and K-tile 0 landed
  lds_barrier();
  if (grp == 1) lds_barrier();      // group 1 runs one slot behind group 0

  int r_slot = 0;      // pixel slot of the pair being read
  auto load_frags = [&](int t, int h) {
    const char* wb = lds + (t & 3) * WW_SLOT + w_off;
    const char* pb = lds + r_slot * WP_SLOT + (p_off ^ (h << 6));
#pragma unroll
    for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const f16x8*>(wb + c * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) pf[q] = *reinterpret_cast<const f16x8*>(pb + q * 2048);
  };
  auto mfmas = [&](int q) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[q], acc[c][q], 0, 0, 0);
  };
  // main loop: pairs (t, t + 1) whose look-ahead exists (cout K-tile t + 4, pixel pair t / 2 + 2)
  int t = 0;
  for (; t + 5 < KTOT; t += 2) {
    // ---- even K-tile: LOAD, then 32 MFMAs with 2 cout + 4 pixel pieces issued in their shadow ----
    load_frags(t, 0);
    w_prep();
    p_prep();
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mfmas(q);
      if (q & 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (q == 1) { dma_w(0); dma_w(1); }
        if (q == 3) { dma_p(0); dma_p(1); }
        if (q == 5) dma_p(2);
        if (q == 7) dma_p(3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    ++w_t;
    p_next();
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    // ---- odd K-tile: the pair's second half, 2 cout pieces ----
    load_frags(t + 1, 1);
    w_prep();
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mfmas(q);
      if (q == 1 || q == 3) {
        __builtin_amdgcn_sched_barrier(0);
        dma_w(q >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    ++w_t;
    r_slot = (r_slot == 2) ? 0 : r_slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }
  // tail: the last four K-tiles (t = KTOT - 4 here); only the cout pieces of K-tile KTOT - 1 are still to be staged
#pragma unroll 1
  for (int tt = 0; tt < 4; ++tt, ++t) {
    load_frags(t, t & 1);
    if (tt == 0) {
      w_prep();
      dma_w(0);
      dma_w(1);
      ++w_t;
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) mfmas(q);
    __builtin_amdgcn_s_setprio(0);
    if (t & 1) r_slot = (r_slot == 2) ? 0 : r_slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }
  if (grp == 0) lds_barrier();      // equal barrier count for both groups
  wave_epilogue_any(p, acc, m0 + grp * 128, n0 + wc * 64, fr, fq, HoWo);
}

// ---------------------------------------------------------------------------------------------------------------
// Half tile: one "group" of the kernel above as its own workgroup -- 4 waves, PXR pixels x COR couts with
// PXR + COR = 384 (128 x 256 or 256 x 128), the same 128 x 64 wave tile (12 fragment reads per 32 MFMAs), K-tiles of
// 32 channels through a ring of THREE 24 KiB slots filled two K-tiles ahead, ONE LDS-only barrier per K-tile, and TWO
// workgroups per CU (72 KiB LDS, <= 256 VGPRs each).  For the layers whose K is too short for the full tile (ResNet
// conv3 / projection shortcuts / stride-2 3x3s: 4-24 K-tiles): there the 256 x 256 tile spends as long in its
// prologue + epilogue (256 KiB of residual + output per tile, nothing to overlap them with on a one-workgroup CU) as
// in its main loop, and the 128 x 128 tile is LDS-read bound (64 x 64 wave tile: 128 B/clk at the MFMA peak).  Two
// independent workgroups overlap one's epilogue with the other's main loop.  K order = the other kernels' order.
template <int PXR, int COR>
__global__ void __launch_bounds__(256, 2) conv_igemm_h256_kernel(const ConvParams p) {
  constexpr int PPW = PXR / 64, CPW = COR / 64;        // 16-row pieces per wave and K-tile: pixels / couts (2+4 or 4+2)
  constexpr int HSLOT = (PXR + COR) * 64;              // 24576
  constexpr int HW_OFF = PXR * 64;
  constexpr int NWC = COR / 64;                        // waves along the couts (4 or 2)
  constexpr int NP = PPW + CPW;                        // 6 DMA pieces per wave and K-tile
  static_assert(PXR + COR == 384 && NP == 6, "half tile: 128 x 256 or 256 x 128");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int mtile = xcd * p.mt_per_xcd + j / p.nt;
  const int ntile = j % p.nt;
  if (mtile >= p.mt) return;
  const int m0 = mtile * PXR, n0 = ntile * COR;

  const int tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / NWC, wc = wave % NWC;
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / KS;
  const int KG = p.kgroup;
  const int KT1 = KT * CB;
  const int KTOT = KT1 + (p.in2 ? p.Cin2 / KS : 0);
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  const int srow = l >> 2;
  const int schunk = (l & 3) ^ ((-(l >> 4)) & 3);
  const half_t* a_img[PPW];
  int a_iy0[PPW], a_ix0[PPW];
  const half_t* a_cur[PPW];
  int a_inc[PPW];
  const half_t* a_two[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int m = m0 + (wave + 4 * i) * 16 + srow;
    a_img[i] = p.in;
    a_iy0[i] = a_ix0[i] = -(1 << 28);
    a_cur[i] = p.zero;
    a_inc[i] = 0;
    a_two[i] = p.zero;
    if (m < p.M) {
      if (p.in2) {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_two[i] = p.in2 + (((size_t)n * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_ld + schunk * 8;
      }
      if (pointwise) {
        a_cur[i] = p.in + (size_t)m * p.in_ld + schunk * 8;
        a_inc[i] = KS;
      } else {
        const int n = m / HoWo;
        const int r = m - n * HoWo;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = p.in + (size_t)n * p.H * p.W * p.in_ld + schunk * 8;
      }
    }
  }
  const half_t* b_base[CPW];
  const half_t* b_cur[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int row = (wave + 4 * i) * 16 + srow;
    const int co = n0 + (row & ~31) + perm32b(row & 31);
    b_base[i] = b_cur[i] = p.wgt + (size_t)co * (KT * p.Cin + (p.in2 ? p.Cin2 : 0)) + schunk * 8;
  }
  int st_ky = 0, st_kx = 0, st_cb = 0, st_grp = 0, st_u = 0;
  auto stage = [&]() {        // address work + the 6 LDS-DMA pieces of K-tile st_u into ring slot st_u % 3
    if (st_u == KT1 && p.in2) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        a_cur[i] = a_two[i];
        a_inc[i] = (a_two[i] != p.zero) ? KS : 0;
      }
    }
    if (st_cb == 0 && st_u < KT1) {
      const int c0 = st_grp * KG * KS;
      if (!pointwise) {
        const int dy = st_ky * p.dil, dx = st_kx * p.dil;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
          const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
          const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          a_cur[i] = ok ? a_img[i] + ((size_t)iy * p.W + ix) * p.in_ld + c0 : p.zero;
          a_inc[i] = ok ? KS : 0;
        }
      }
      if (KG != CB) {
        const int koff = (st_ky * p.KW + st_kx) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < CPW; ++i) b_cur[i] = b_base[i] + koff;
      }
    }
    char* slot = lds + (st_u % 3) * HSLOT;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[i],
                                       (__attribute__((address_space(3))) void*)(slot + (wave + 4 * i) * 1024), 16, 0, 0);
      a_cur[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[i],
                                       (__attribute__((address_space(3))) void*)(slot + HW_OFF + (wave + 4 * i) * 1024), 16, 0, 0);
      b_cur[i] += KS;
    }
    if (++st_cb == KG) {
      st_cb = 0;
      if (++st_kx == p.KW) {
        st_kx = 0;
        if (++st_ky == p.KH) { st_ky = 0; ++st_grp; }
      }
    }
    ++st_u;
  };

  const int fr = l & 15, fq = l >> 4;
  const int foff = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
  const int p_off = wp * 128 * 64 + foff;
  const int w_off = HW_OFF + wc * 64 * 64 + foff;

  f32x4 acc[4][8];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: two K-tiles in flight, the first one landed
  stage();
  if (KTOT > 1) {
    stage();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  lds_barrier();

  f16x8 pf[8], wf[4];
  for (int t = 0; t < KTOT; ++t) {
    const char* base = lds + (t % 3) * HSLOT;
#pragma unroll
    for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const f16x8*>(base + w_off + c * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) pf[q] = *reinterpret_cast<const f16x8*>(base + p_off + q * 1024);
    // K-tile t+2 goes into the slot of K-tile t-1, which every wave finished reading before the previous barrier
    if (t + 2 < KTOT) {
      stage();
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");     // my pieces of t+1 landed, t+2 in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[c], pf[q], acc[c][q], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_epilogue_any(p, acc, m0 + wp * 128, n0 + wc * 64, fr, fq, HoWo);
}

}  // namespace

bool conv_b2b_supported(const ConvParams& p) {
  return p.next_w != nullptr && p.Cout == 256 && p.next_cout == 64 &&
         p.next_ld % 8 == 0 && p.ps_cout == 0 && p.out2 == nullptr && p.bias_n == nullptr && p.act <= 1 &&
         (int64_t)cdiv(p.M, 256) >= 192 && p.Cin % KS == 0 && (!p.in2 || p.Cin2 % KS == 0);
}

bool conv_igemm256_supported(const ConvParams& p) {
  return p.Cout % 256 == 0 && p.ps_cout == 0 && p.Cin % KS == 0 && p.KH * p.KW * (p.Cin / KS) + (p.in2 ? p.Cin2 / KS : 0) >= 1;
}

// Packed weight image for conv_igemm256_kernel (ConvParams::kgroup < 0): [cout tile of 256][K-tile in walk order][piece of 16
// rows][lane][8 halfs] -- the bytes lane l of the wave that stages piece pc of K-tile t writes to LDS, in LDS order, i.e. the
// row permutation (perm32b), the chunk swizzle and the K walk (groups of kg slabs, tap-major inside a group, the second
// source's channels behind the main source's) are applied here once instead of by every workgroup's address arithmetic.
__global__ void __launch_bounds__(256) pack256_kernel(const half_t* __restrict__ w, half_t* __restrict__ out, int Cout, int KT,
                                                      int Cin, int Cin2, int KG) {
  const int CB = Cin / KS, KT1 = KT * CB, KTOT = KT1 + Cin2 / KS, Krow = KT * Cin + Cin2;
  const int64_t total = (int64_t)(Cout / 256) * KTOT * 16 * 64;      // 16-byte chunks
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int l = (int)(i & 63), pc = (int)((i >> 6) & 15);
    const int64_t tt = i >> 10;
    const int t = (int)(tt % KTOT), ntile = (int)(tt / KTOT);
    const int row = pc * 16 + (l >> 2);
    const int co = ntile * 256 + (row & ~31) + perm32b(row & 31);
    const int chunk = (l & 3) ^ ((-(l >> 4)) & 3);
    int koff;
    if (t < KT1) {
      const int per = KT * KG, g = t / per, idx = t - g * per, tap = idx / KG, cb = idx - tap * KG;
      koff = tap * Cin + (g * KG + cb) * KS;
    } else {
      koff = KT * Cin + (t - KT1) * KS;
    }
    *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(w + (size_t)co * Krow + koff + chunk * 8);
  }
}

template <int DMA_EARLY, bool B2B = false>
static int launch256(const ConvParams& p, int grid, hipStream_t stream) {
  constexpr int LDS_BYTES = NSLOT * SLOT_BYTES + (B2B ? 64 * 256 * 2 : 0);
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv_igemm256_kernel<DMA_EARLY, B2B>), LDS_BYTES)) return rc;
  hipLaunchKernelGGL((conv_igemm256_kernel<DMA_EARLY, B2B>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// kg: channel slabs (32 ch) per K-walk group, 0 = default; mode: 0 = default, 1 / 2 = 2 / 4 DMA pieces issued early
int launch_conv_igemm256(ConvParams p, hipStream_t stream, int kg, int mode, bool wpacked) {
  EMP_REQUIRE(conv_igemm256_supported(p), "conv256: unsupported shape (Cout=%d)", p.Cout);
  EMP_REQUIRE(p.next_w == nullptr || conv_b2b_supported(p), "conv256: the fused next convolution needs Cout == 256 (got %d)", p.Cout);
  EMP_REQUIRE(p.out2 == nullptr || (p.split % 256 == 0 && p.split > 0 && p.split < p.Cout && p.out3 == nullptr &&
                                    p.next_w == nullptr && p.out2_ld % 8 == 0 && p.out2_ld >= p.Cout - p.split),
              "conv256: a second destination takes whole 256-cout tiles (split=%d)", p.split);
  {
    const int CB = p.Cin / KS, KT = p.KH * p.KW;
    static const int env_kg = [] { const char* e = getenv("EMP_CONV256_KGROUP"); return e ? atoi(e) : 0; }();   // A/B runs
    if (kg == 0) kg = env_kg;
    if (kg == 0) kg = 8;                                       // 256-channel groups, see conv_igemm.hip
    if (KT == 1 || kg > CB || CB % kg != 0) kg = CB;
    p.kgroup = kg;
  }
  if (wpacked) {      // ConvParams::wgt is conv256_pack_weights' image, made for the walk group conv256_pack_kgroup returns
    EMP_REQUIRE(p.next_w == nullptr && p.kgroup == conv256_pack_kgroup(p.KH * p.KW, p.Cin), "conv256: packed weights need the default K walk");
    p.kgroup = -p.kgroup;
  }
  static const int env_mode = [] { const char* e = getenv("EMP_CONV256_MODE"); return e ? atoi(e) : 0; }();
  if (mode == 0) mode = env_mode;
  p.mt = cdiv(p.M, 256);
  p.nt = p.Cout / 256;
  p.mt_per_xcd = cdiv(p.mt, 8);
  {
    // EMP_CONV256_NGROUP=g: the intermediate raster for the 1x1 convolutions with at least 2g cout tiles (A/B; the measured
    // outcome is in profiles/r06_conv_raster.txt)
    static const int env_ng = [] { const char* e = getenv("EMP_CONV256_NGROUP"); return e ? atoi(e) : 0; }();
    p.ngroup = (env_ng > 0 && p.KH * p.KW == 1 && p.nt >= 2 * env_ng && p.nt % env_ng == 0 && !p.next_w) ? env_ng : 0;
  }
  const int grid = 8 * p.mt_per_xcd * p.nt;
  if (p.next_w) return launch256<0, true>(p, grid, stream);
  {
    // whole-line pixel fetches (conv_igemm256w_kernel): K-tile pairs need 64-channel granularity along the whole walk
    static const bool wide_on = [] { const char* e = getenv("EMP_CONV256_WIDE"); return !(e && e[0] == '0'); }();      // A/B runs
    const int kga = p.kgroup < 0 ? -p.kgroup : p.kgroup;
    const int ktot = p.KH * p.KW * (p.Cin / KS) + (p.in2 ? p.Cin2 / KS : 0);
    if (wide_on && mode == 0 && p.Cin % (2 * KS) == 0 && (!p.in2 || p.Cin2 % (2 * KS) == 0) && kga % 2 == 0 && ktot >= 8 &&
        (int64_t)p.N * p.H * p.W < (1ll << 31)) {      // (its pixel indices are 32-bit)
      if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv_igemm256w_kernel), WIDE_LDS_BYTES)) return rc;
      hipLaunchKernelGGL(conv_igemm256w_kernel, dim3(grid), dim3(512), WIDE_LDS_BYTES, stream, p);
      EMP_LAUNCH_CHECK();
      return EMP_OK;
    }
  }
  if (mode == 1) return launch256<2>(p, grid, stream);
  if (mode == 2) return launch256<4>(p, grid, stream);
  return launch256<0>(p, grid, stream);
}

// the K-walk group (32-channel slabs) launch_conv_igemm256 takes by default: the packed image is laid out along it
int conv256_pack_kgroup(int KT, int Cin) {
  const int CB = Cin / KS;
  static const int env_kg = [] { const char* e = getenv("EMP_CONV256_KGROUP"); return e ? atoi(e) : 0; }();
  int kg = env_kg ? env_kg : 8;
  if (KT == 1 || kg > CB || CB % kg != 0) kg = CB;
  return kg;
}
// w: [Cout][KT * Cin + Cin2] halfs -> out: the same number of halfs as a packed image (Cout % 256 == 0, Cin, Cin2 % 32 == 0)
int conv256_pack_weights(const half_t* w, half_t* out, int Cout, int KT, int Cin, int Cin2, hipStream_t stream) {
  EMP_REQUIRE(w && out && Cout > 0 && Cout % 256 == 0 && Cin > 0 && Cin % KS == 0 && Cin2 >= 0 && Cin2 % KS == 0 && KT >= 1,
              "conv256_pack_weights: bad shape (Cout=%d Cin=%d Cin2=%d)", Cout, Cin, Cin2);
  const int64_t chunks = (int64_t)Cout * (KT * Cin + Cin2) / 8;
  hipLaunchKernelGGL(pack256_kernel, dim3((unsigned)std::min<int64_t>((chunks + 255) / 256, 4096)), dim3(256), 0, stream, w, out, Cout,
                     KT, Cin, Cin2, conv256_pack_kgroup(KT, Cin));
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// half tile (conv_igemm_h256_kernel): 128 pixels x 256 couts when Cout % 256 == 0, else 256 pixels x 128 couts
bool conv_igemm_h256_supported(const ConvParams& p) {
  return p.Cout % 128 == 0 && p.ps_cout == 0 && p.out2 == nullptr && p.Cin % KS == 0 && (!p.in2 || p.Cin2 % KS == 0);
}

template <int PXR, int COR>
static int launch_h256(ConvParams p, hipStream_t stream) {
  constexpr int LDS_BYTES = 3 * (PXR + COR) * 64;
  if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(&conv_igemm_h256_kernel<PXR, COR>), LDS_BYTES)) return rc;
  p.mt = cdiv(p.M, PXR);
  p.nt = p.Cout / COR;
  p.mt_per_xcd = cdiv(p.mt, 8);
  const int grid = 8 * p.mt_per_xcd * p.nt;
  hipLaunchKernelGGL((conv_igemm_h256_kernel<PXR, COR>), dim3(grid), dim3(256), LDS_BYTES, stream, p);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_conv_igemm_h256(ConvParams p, hipStream_t stream, int kg) {
  EMP_REQUIRE(conv_igemm_h256_supported(p), "conv h256: unsupported shape (Cout=%d)", p.Cout);
  const int CB = p.Cin / KS, KT = p.KH * p.KW;
  if (kg == 0) kg = 8;
  if (KT == 1 || kg > CB || CB % kg != 0) kg = CB;
  p.kgroup = kg;
  if (p.Cout % 256 == 0) return launch_h256<128, 256>(p, stream);
  return launch_h256<256, 128>(p, stream);
}

}  // namespace emp

#ifdef EMP_CLOCK_STAMP
extern "C" __attribute__((visibility("default"))) int emp_diag_clock_stamps(unsigned long long* host, int n_workgroups) {
  if (!host || n_workgroups <= 0 || n_workgroups > 8192) return -1;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(emp::emp_clock_stamp_buf), sizeof(unsigned long long) * 2 * (size_t)n_workgroups);
}
#endif
