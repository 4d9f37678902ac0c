// fp16x3 convolution on the dominant kernel's machinery (round 6): 256 pixels x 256 couts per workgroup,
// eight waves in two staggered groups, BOTH operands by LDS-DMA from pre-split fp16 planes -- no register staging, no split in
// the kernel -- and three MFMAs per fragment pair into the same 128 accumulator registers.
//
// The reference computes this path in fp32 (empanada/inference/engines.py:248-255; the convolutions that dominate:
// models/decoders/aspp.py:51-103, models/encoders/resnet.py:109-129).  conv16x3.hip (round 5) computes every product as
//     w_lo . x_hi  +  w_hi . x_hi  +  w_hi . x_lo        (x = hi + lo, hi = fp16(x), lo = fp16(x - hi): 22 significant bits)
// on the fp16 matrix pipe from fp32 maps that every workgroup loads to registers, splits and stores to LDS: 96 fp16 flops per
// L2 byte on a 128 x 128 tile, 0.28 of the fp16 pipe whole-step.  Here the SPLIT IS DONE BY THE PRODUCER: a map of the
// stride-16 region (layer3, layer4, ASPP) is kept in HBM as "hl32" rows -- per pixel and per block of 32 channels 64 B of hi
// halfs followed by 64 B of lo halfs, the same 4 B per element as the fp32 map -- so that the 128 bytes one pixel contributes to
// a K step are ONE whole cache line and one row of the LDS tile; weights come from a packed image (x3p_pack_kernel) whose 1 KiB
// LDS-DMA pieces are contiguous.  Per K step of 32 channels a workgroup moves 64 KiB into LDS for 12.6 MFLOP: 192 fp16 flops per
// L2 byte (conv_igemm256w_kernel: 128), and 24 ds_read_b128 per 96 MFMAs and wave (the fp16 kernel: 12 per 32).
//
// LDS (all 160 KiB):   Wl ring 2 x 16 KiB | Wh ring 2 x 16 KiB | X ring 3 x 32 KiB
//   W slots: 256 cout rows of 64 B, 16-byte chunk c of row r at chunk c ^ ((-(r >> 2)) & 3)   (conv_igemm256.hip's cout slot)
//   X slots: 256 pixel rows of 128 B = [hi chunks 0-3 | lo chunks 4-7], chunk c of row r at c ^ ((r >> 1) & 7)
//            (conv_igemm256w_kernel's pair slot with "second K-tile of the pair" replaced by "lo": same conflict-free reads)
// Schedule.  Time is cut into SLOTS by workgroup barriers; group 0 (waves 0-3: pixels 0-127) runs LOAD(t) in slot 2t and
// COMPUTE(t) in slot 2t + 1, group 1 (waves 4-7, the SIMD partners) one slot later, so a SIMD always has one wave in its 96
// MFMAs and one in its fragment reads / DMA issue / waits.  LOAD(t) reads Wl(t) and the hi half of X(t) (12 ds_read_b128);
// COMPUTE(t) runs pass 1 (w_lo . x_hi) with the four Wh(t) reads in its shadow, pass 2 (w_hi . x_hi) with the eight lo reads of
// X(t) in its shadow, pass 3 (w_hi . x_lo): at most 80 fragment registers live, two barriers per 96 MFMAs (the fp16 kernel: two
// per 32).  DMA by slot, every wave its own pieces (X: 4 of 32, Wl / Wh: 2 of 16 each):
//     even slot 2t    : Wl(t + 1)                       (its ring slot was read last in slot 2t - 1)
//     odd slot 2t + 1 : Wh(t + 1), then X(t + 2)        (read last in slot 2t)
// Before the barrier that closes ANY slot a wave waits for all but its 6 youngest DMA operations -- one constant counted
// vmcnt(6): Wl(t + 1) has landed a full slot before it is read (slot 2t + 2), Wh(t + 1) likewise (2t + 3), X(t + 2) two slots
// (2t + 4).  With the tap-major K walk (the C ABI's emp_conv2d_hl32_f16x3; X3P::kg = Cin / 32) the K order and the three products per
// step are conv16x3.hip's: bit-identical accumulators given the same operands (tests/test_gpu_x3p.py).  The network walks K in
// groups of 128 channels (x3p_kgroup below): another summation order of the same terms.
#include <type_traits>

#include "common.h"

namespace emp {
namespace {

constexpr int KS = 32;
constexpr int WSLOT = 256 * 64;                  // one of Wl / Wh for one K step
constexpr int XSLOT = 256 * 128;                 // [hi | lo] pixel rows of one K step
constexpr int WL_BASE = 0, WH_BASE = 2 * WSLOT, X_BASE = 4 * WSLOT;
constexpr int X3P_LDS = X_BASE + 3 * XSLOT;      // 163840

struct X3P {
  const half_t* in; int in_ld;          // hl32 map: rows 2 * in_ld halfs apart; channel c at (c / 32) * 64 + c % 32, its lo half 32 further
  const half_t* wimg;                   // packed image: [cout tile][K step][Wl 16 pieces | Wh 16 pieces][1 KiB]
  const float* bias; const float* bias_n;
  const void* res; int res_ld, res_fmt; // residual: fp32 rows (0) or hl32 rows (1)
  void* out; int out_ld;                // fp32 rows of out_ld floats or hl32 rows of 2 * out_ld halfs (template)
  void* out2; int out2_ld, split;       // optional: couts [split, Cout) go to out2 (its channel 0 = cout `split`), split % 256 == 0 --
                                        // two convolutions of one map (the two decoders' ASPP branches) as one launch
  const half_t* zero;
  int N, H, W, Cin, Cout, KH, KW, stride, pad, dil, Ho, Wo;
  int M, mt, nt, mt_per_xcd;
  int kg;      // K steps (32 channels) per K-walk group and tap: the walk is [group of kg * 32 channels][tap][step]; kg == Cin / 32: tap-major
  int ksteps, ksplit;  // KSPLIT instantiation: ksplit = S in {2, 4, 8} workgroups per tile, split s walks K steps [s * ksteps, ...) and stores
               // raw partial sums to out + s * M * Cout floats (>= 4 steps in every split)
};

__device__ __forceinline__ int perm32b(int x) {
  const int t = x >> 4, i = x & 15;
  return ((i >> 2) << 3) + (t << 2) + (i & 3);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int ACT>
__device__ __forceinline__ float act_p(float x) {
  if (ACT == 1) return fmaxf(x, 0.f);
  if (ACT == 2) return x / (1.f + __expf(-x));
  return x;
}

// lane (fq, fr) owns couts co0 + P * 32 + fq * 8 + [0, 8) of pixel mrow0 + q * 16 + fr (the weight rows are permuted in the
// image, perm32b): 32 contiguous bytes of an fp32 row, or 16 B of hi and 16 B of lo inside one 32-channel block of an hl32 row
template <int ACT, int OUTF, int RESF>      // RESF: 0 no residual, 1 fp32 rows, 2 hl32 rows
__device__ __forceinline__ void x3p_epilogue(const X3P& p, void* out_base, f32x4 (&acc)[4][8], int mrow0, int co0, int fr, int fq, int HoWo) {
  // Two passes (couts P * 32 + fq * 8 + [0, 8) each).  The residual of FOUR pixel fragments at a time is requested up front --
  // 8 loads in flight per lane, in the K loop's free fragment registers -- instead of one load -> wait -> store round trip per
  // fragment: with one workgroup per CU nothing else hides that latency, and the short-K launches (conv3 of a bottleneck: 8-16 K
  // steps) spent more time in 16 serial round trips than in their main loop (finding 66).  Rows past M load row M - 1 and are not stored.
  const int mlast = p.M - 1;
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int co = co0 + P * 32 + fq * 8;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
      bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
    const bool second = p.out2 && co0 >= p.split;      // (whole 256-cout tiles: wave-uniform)
    const int oc = second ? co - p.split : co;
    const int old_ = second ? p.out2_ld : p.out_ld;
    void* const obase = second ? p.out2 : out_base;
    constexpr int QG = 4;      // fragments per group
#pragma unroll
    for (int qh = 0; qh < 8 / QG; ++qh) {      // (four fragments at a time: 32 registers of residual beside the 128 accumulators)
      constexpr int NRH = RESF == 2 ? QG : 1, NRF = RESF == 1 ? QG : 1;
      f16x8 rh[NRH], rl[NRH];
      float4 r0[NRF], r1[NRF];
      if constexpr (RESF == 2) {
#pragma unroll
        for (int u = 0; u < QG; ++u) {
          const int m = min(mrow0 + (qh * QG + u) * 16 + fr, mlast);
          const half_t* rp = reinterpret_cast<const half_t*>(p.res) + (size_t)m * (2 * p.res_ld) + (co >> 5) * 64 + (co & 31);
          rh[u] = *reinterpret_cast<const f16x8*>(rp);
          rl[u] = *reinterpret_cast<const f16x8*>(rp + 32);
        }
      } else if constexpr (RESF == 1) {
#pragma unroll
        for (int u = 0; u < QG; ++u) {
          const int m = min(mrow0 + (qh * QG + u) * 16 + fr, mlast);
          const float* rp = reinterpret_cast<const float*>(p.res) + (size_t)m * p.res_ld + co;
          r0[u] = *reinterpret_cast<const float4*>(rp);
          r1[u] = *reinterpret_cast<const float4*>(rp + 4);
        }
      }
#pragma unroll
      for (int u = 0; u < QG; ++u) {
        const int q = qh * QG + u;
        const int m = mrow0 + q * 16 + fr;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[2 * P][q][r] + bv[r];
          v[4 + r] = acc[2 * P + 1][q][r] + bv[4 + r];
        }
        if (p.bias_n) {
          const float* bn = p.bias_n + (size_t)(min(m, mlast) / HoWo) * p.Cout + co;
          const float4 b0 = *reinterpret_cast<const float4*>(bn);
          const float4 b1 = *reinterpret_cast<const float4*>(bn + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if constexpr (RESF == 2) {      // hi + lo is exact in fp32 (22 bits)
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rh[u][r] + (float)rl[u][r];
        } else if constexpr (RESF == 1) {
          v[0] += r0[u].x; v[1] += r0[u].y; v[2] += r0[u].z; v[3] += r0[u].w;
          v[4] += r1[u].x; v[5] += r1[u].y; v[6] += r1[u].z; v[7] += r1[u].w;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = act_p<ACT>(v[r]);
        if (m >= p.M) continue;
        if (OUTF) {
          half_t* op = reinterpret_cast<half_t*>(obase) + (size_t)m * (2 * old_) + (oc >> 5) * 64 + (oc & 31);
          f16x8 h, lo;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            h[r] = (half_t)v[r];
            lo[r] = (half_t)(v[r] - (float)h[r]);
          }
          *reinterpret_cast<f16x8*>(op) = h;
          *reinterpret_cast<f16x8*>(op + 32) = lo;
        } else {
          float* op = reinterpret_cast<float*>(obase) + (size_t)m * old_ + oc;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
  }
}

template <int ACT, int OUTF, int RESF, bool KSPLIT = false>
__global__ void __launch_bounds__(512, 1) conv16x3p_kernel(const X3P p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  int mtile = xcd * p.mt_per_xcd + j / p.nt;      // XCD-aware raster: the cout tiles of a pixel tile side by side on one XCD
  int ntile = j % p.nt;
  int ksplit_idx = 0;
  if constexpr (KSPLIT) {
    // S in {2, 4, 8} splits: a split's workgroups share its K range of the weight image and its channel range of the map, so a
    // split lives on 8 / S XCDs -- every XCD's L2 fetches 1 / S of the weights (with the plain raster each of the 8 L2s fetched all
    // of them: the merged ASPP branch of one tile moved 300 MB instead of 38 and ran at a third of its MFMA rate)
    const int S = p.ksplit, per = 8 / S;
    ksplit_idx = xcd % S;
    const int tile = j * per + xcd / S;
    mtile = tile / p.nt;
    ntile = tile - mtile * p.nt;
  }
  if (mtile >= p.mt) return;
  const int m0 = mtile * 256, n0 = ntile * 256;

  const int tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int HoWo = p.Ho * p.Wo;
  const int KT = p.KH * p.KW;
  const int CB = p.Cin / KS;              // K steps per tap
  const int KTOT_ALL = KT * CB;           // >= 4 (launcher)
  const int t0 = KSPLIT ? ksplit_idx * p.ksteps : 0;                                    // KSPLIT: this workgroup's first K step ...
  const int KTOT = KSPLIT ? min(p.ksteps, KTOT_ALL - t0) : KTOT_ALL;                    // ... and how many it walks (>= 4, launcher)
  const bool pointwise = (KT == 1) && p.stride == 1 && p.pad == 0;

  // ---- weight pieces: buffer form (descriptor + scalar piece offset + one lane-offset register) ----
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.wimg), 0, 0x7fffffff, 0x00020000);
  const int w_voff = l * 16;
  const int w_tile0 = ntile * KTOT_ALL + t0;
  auto dma_w = [&](int t, int hi) {      // this wave's two pieces (wave, wave + 8) of Wl(t) / Wh(t) into ring slot t & 1
    const int soff = ((w_tile0 + t) * 32 + hi * 16 + wave) * 1024;
    char* dst = lds + (hi ? WH_BASE : WL_BASE) + (t & 1) * WSLOT + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, w_voff, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst + 8 * 1024), 16, w_voff, soff + 8 * 1024, 0, 0);
  };

  // ---- pixel pieces {w, w + 8, w + 16, w + 24} of 8 rows x 128 B per K step: whole lines, [hi | lo] of 32 channels ----
  const int prow = l >> 3;
  const int pchunk = (l & 7) ^ (((wave & 1) << 2) | (prow >> 1));      // logical chunk this lane fetches: position ^ ((row >> 1) & 7)
  const int row_halfs = 2 * p.in_ld;
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
        a_cur[i] = p.in + (size_t)m * row_halfs + pchunk * 8 + t0 * (2 * KS);
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
  int x_ky = 0, x_kx = 0, x_cb = 0, x_grp = 0, x_slot = 0;
  const int KG = p.kg;
  if constexpr (KSPLIT) {      // step t0 of the walk [group][tap][step]: its group, tap and step inside the tap's group
    const int per = KT * KG;
    x_grp = t0 / per;
    const int idx = t0 - x_grp * per;
    const int tap = idx / KG;
    x_cb = idx - tap * KG;
    x_ky = tap / p.KW;
    x_kx = tap - x_ky * p.KW;
    if (x_cb != 0 && !pointwise) {      // the split starts inside a tap's group: x_prep computes addresses at x_cb == 0 only
      const int dy = x_ky * p.dil, dx = x_kx * p.dil;
      const int c0 = (x_grp * KG + x_cb) * (2 * KS) + pchunk * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        a_cur[i] = ok ? p.in + ((size_t)(a_pix[i] + iy * p.W + ix) * row_halfs + c0) : p.zero;
        a_inc[i] = ok ? 2 * KS : 0;
      }
    }
  }
  auto x_prep = [&]() {      // address work of the next K step's pixel pieces (in a LOAD phase, out of the MFMAs' way)
    if (x_cb == 0 && !pointwise) {
      const int dy = x_ky * p.dil, dx = x_kx * p.dil;
      const int c0 = x_grp * KG * (2 * KS) + pchunk * 8;      // halfs into the hl32 row: the group's first 32-channel block
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        a_cur[i] = ok ? p.in + ((size_t)(a_pix[i] + iy * p.W + ix) * row_halfs + c0) : p.zero;
        a_inc[i] = ok ? 2 * KS : 0;
      }
    }
    if (++x_cb == KG) {
      x_cb = 0;
      if (++x_kx == p.KW) {
        x_kx = 0;
        if (++x_ky == p.KH) { x_ky = 0; ++x_grp; }
      }
    }
  };
  auto dma_x = [&](int i) {
    char* dst = lds + X_BASE + x_slot * XSLOT + (wave + 8 * i) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[i],
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    a_cur[i] += a_inc[i];
  };
  auto x_next = [&]() { x_slot = (x_slot == 2) ? 0 : x_slot + 1; };

  // ---- fragment addressing (lane-constant) ----
  const int fr = l & 15, fq = l >> 4;
  const int w_off = wc * 64 * 64 + fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);                  // + c * 1024, in a Wl / Wh slot
  const int p_off = X_BASE + grp * 128 * 128 + fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);      // + q * 2048; ^ 64: the lo half

  f32x4 acc[4][8];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 wl[4], wh[4], xh[8], xl[8];

  // ---- prologue, in the order the steady state's counted wait assumes: X(0), Wl(0), Wh(0), X(1) ----
  x_prep();
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_x(i);
  x_next();
  dma_w(0, 0);
  dma_w(0, 1);
  x_prep();
#pragma unroll
  for (int i = 0; i < 4; ++i) dma_x(i);
  x_next();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // X(0), Wl(0) landed
  lds_barrier();                                        // slot 0 opens

  int r_slot = 0;      // X ring slot of the K step being read
  auto load_first = [&](int t) {      // LOAD(t): Wl(t) and the hi half of X(t)
    const char* wb = lds + WL_BASE + (t & 1) * WSLOT + w_off;
    const char* pb = lds + r_slot * XSLOT + p_off;
#pragma unroll
    for (int c = 0; c < 4; ++c) wl[c] = *reinterpret_cast<const f16x8*>(wb + c * 1024);
#pragma unroll
    for (int q = 0; q < 8; ++q) xh[q] = *reinterpret_cast<const f16x8*>(pb + q * 2048);
  };
  // COMPUTE(t): three passes over the wave's 4 x 8 fragment pairs; DMA is issued in pass 1's shadow
  //   ev: 0 nothing, 1 this wave's Wl pieces of K step tw (an even slot), 2 its Wh pieces of K step tw and its X pieces (an odd slot),
  //       3 the Wh pieces alone (the last odd slot with anything to stage)
  auto compute = [&](int t, int ev, int tw) {
    const char* whb = lds + WH_BASE + (t & 1) * WSLOT + w_off;
    const char* plb = lds + r_slot * XSLOT + (p_off ^ 64);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {      // pass 1: w_lo . x_hi
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], xh[q], acc[c][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wh[c] = *reinterpret_cast<const f16x8*>(whb + c * 1024);
      }
      if (q == 1 && ev == 1) dma_w(tw, 0);
      if (q == 1 && (ev == 2 || ev == 3)) dma_w(tw, 1);
      if (ev == 2) {
        if (q == 3) { dma_x(0); dma_x(1); }
        if (q == 5) dma_x(2);
        if (q == 7) dma_x(3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {      // pass 2: w_hi . x_hi
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], xh[q], acc[c][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q == 0) {      // (the registers of w_lo are free)
#pragma unroll
        for (int u = 0; u < 4; ++u) xl[u] = *reinterpret_cast<const f16x8*>(plb + u * 2048);
      }
      if (q == 4) {      // (and those of x_hi[0..3])
#pragma unroll
        for (int u = 4; u < 8; ++u) xl[u] = *reinterpret_cast<const f16x8*>(plb + u * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {      // pass 3: w_hi . x_lo
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], xl[q], acc[c][q], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    r_slot = (r_slot == 2) ? 0 : r_slot + 1;
  };

  if (grp == 0) {
    // group 0: LOAD(t) in the even slot 2t (issues Wl(t + 1)), COMPUTE(t) in the odd slot 2t + 1 (issues Wh(t + 1), X(t + 2))
    int t = 0;
    for (; t + 2 < KTOT; ++t) {
      load_first(t);
      x_prep();
      dma_w(t + 1, 0);
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
      compute(t, 2, t + 1);
      x_next();
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
    }
    // t = KTOT - 2: Wl(t + 1) and Wh(t + 1) are the last pieces there are
    load_first(t);
    dma_w(t + 1, 0);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    compute(t, 3, t + 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    ++t;
    load_first(t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    compute(t, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    lds_barrier();      // equal barrier count for both groups
  } else {
    // group 1, one slot behind: slot 0 is only its DMA share; LOAD(t) in the odd slot 2t + 1 (issues Wh(t + 1), X(t + 2)),
    // COMPUTE(t) in the even slot 2t + 2 (issues Wl(t + 2))
    dma_w(1, 0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    lds_barrier();
    int t = 0;
    for (; t + 2 < KTOT; ++t) {
      load_first(t);
      x_prep();
      dma_w(t + 1, 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma_x(i);
      x_next();
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
      compute(t, 1, t + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      lds_barrier();
    }
    // t = KTOT - 2 (odd slot: Wh(t + 1) only), then nothing left to stage
    load_first(t);
    dma_w(t + 1, 1);
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    compute(t, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    ++t;
    load_first(t);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    compute(t, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);      // (no epilogue address arithmetic hoisted into the last K steps: it spilled there)
  void* out_base = p.out;
  if constexpr (KSPLIT) out_base = reinterpret_cast<float*>(p.out) + (size_t)ksplit_idx * p.M * p.Cout;
  x3p_epilogue<ACT, OUTF, RESF>(p, out_base, acc, m0 + grp * 128, n0 + wc * 64, fr, fq, HoWo);
}

// [cout tile of 256][K step][part: 0 = lo, 1 = hi][piece of 16 rows][lane][8 halfs]: the bytes lane l of the wave that stages piece
// pc writes to LDS, in LDS order -- conv_igemm256.hip's pack256_kernel with the split applied (rows permuted by perm32b, chunks
// swizzled).  w: [Cout][K] fp32, K = KH * KW * Cin walked tap-major.
__global__ void __launch_bounds__(256) x3p_pack_kernel(const float* __restrict__ w, half_t* __restrict__ out, int Cout, int K, int KT, int Cin,
                                                       int KG) {
  const int KTOT = K / KS;
  const int64_t total = (int64_t)(Cout / 256) * KTOT * 2 * 16 * 64;      // 16-byte chunks
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int l = (int)(i & 63), pc = (int)((i >> 6) & 15), part = (int)((i >> 10) & 1);
    const int64_t tt = i >> 11;
    const int t = (int)(tt % KTOT), ntile = (int)(tt / KTOT);
    const int row = pc * 16 + (l >> 2);
    const int co = ntile * 256 + (row & ~31) + perm32b(row & 31);
    const int chunk = (l & 3) ^ ((-(l >> 4)) & 3);
    // K step t of the walk [group][tap][step] -> its place in the row (tap-major there): tap * Cin + (g * KG + cb) * 32
    const int per = KT * KG, g = t / per, idx = t - g * per, tap = idx / KG, cb = idx - tap * KG;
    const float* src = w + (size_t)co * K + tap * Cin + (g * KG + cb) * KS + chunk * 8;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = src[e];
      const half_t h = (half_t)x;
      v[e] = part ? h : (half_t)(x - (float)h);
    }
    *reinterpret_cast<f16x8*>(out + i * 8) = v;
  }
}

// fp32 rows -> hl32 rows and back (the boundary of the plane region; C % 32 == 0)
__global__ void __launch_bounds__(256) hl32_from_f32_kernel(const float* __restrict__ in, half_t* __restrict__ out, int64_t rows, int C,
                                                            int in_ld, int out_ld) {
  const int cpr = C / 8;      // 8-channel chunks per row
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * cpr; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;
    const float4 a = *reinterpret_cast<const float4*>(in + m * in_ld + c), b = *reinterpret_cast<const float4*>(in + m * in_ld + c + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 h, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = (half_t)v[e];
      lo[e] = (half_t)(v[e] - (float)h[e]);
    }
    half_t* op = out + m * (2 * (int64_t)out_ld) + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<f16x8*>(op) = h;
    *reinterpret_cast<f16x8*>(op + 32) = lo;
  }
}
__global__ void __launch_bounds__(256) hl32_to_f32_kernel(const half_t* __restrict__ in, float* __restrict__ out, int64_t rows, int C,
                                                          int in_ld, int out_ld) {
  const int cpr = C / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * cpr; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cpr;
    const int c = (int)(i - m * cpr) * 8;
    const half_t* ip = in + m * (2 * (int64_t)in_ld) + (c >> 5) * 64 + (c & 31);
    const f16x8 h = *reinterpret_cast<const f16x8*>(ip), lo = *reinterpret_cast<const f16x8*>(ip + 32);
    float* op = out + m * out_ld + c;
    *reinterpret_cast<float4*>(op) = make_float4((float)h[0] + (float)lo[0], (float)h[1] + (float)lo[1], (float)h[2] + (float)lo[2], (float)h[3] + (float)lo[3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4((float)h[4] + (float)lo[4], (float)h[5] + (float)lo[5], (float)h[6] + (float)lo[6], (float)h[7] + (float)lo[7]);
  }
}

// per-image mean over the pixels of an hl32 map (aspp.py:45-48's AdaptiveAvgPool2d(1) on the plane-region p5): the fp32 mode's
// avgpool order -- 16 waves x 4 partials per channel pair, then a tree in LDS -- on hi + lo
__global__ void __launch_bounds__(1024) avgpool_hl32_kernel(const half_t* __restrict__ in, int HW, int C, int in_ld, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int n = blockIdx.y, c0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = c0 + lane;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const half_t* base = in + (size_t)n * HW * (2 * in_ld) + (c >> 5) * 64 + (c & 31);
    int k = 0;
    for (int px = wv; px < HW; px += 16, ++k) {
      const half_t* q = base + (size_t)px * (2 * in_ld);
      s[k & 3] += (float)q[0] + (float)q[32];
    }
  }
  part[wv][lane] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  for (int st = 8; st >= 1; st >>= 1) {
    if (wv < st) part[wv][lane] += part[wv + st][lane];
    __syncthreads();
  }
  if (wv == 0 && c < C) out[(size_t)n * C + c] = part[0][lane] / (float)HW;
}

// split-K finish: out = act(sum over s (ascending) of part[s] + bias + bias_n + res), eight couts per thread; c carries the
// ORIGINAL launch's epilogue operands (p.out = the fp32 / hl32 destination), part = the KSPLIT launch's scratch [S][M][Cout]
template <int ACT>
__global__ void __launch_bounds__(256) x3p_finish_kernel(const X3P p, const float* __restrict__ part, int S, int out_fmt, int64_t total) {
  const int C8 = p.Cout >> 3, HoWo = p.Ho * p.Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / C8;
    const int co = (int)(i - m * C8) * 8;
    float v[8];
    {
      const float* q = part + (size_t)m * p.Cout + co;
      const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    for (int sidx = 1; sidx < S; ++sidx) {
      const float* q = part + ((size_t)sidx * p.M + m) * p.Cout + co;
      const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (p.bias) {
      const float4 a = *reinterpret_cast<const float4*>(p.bias + co), b = *reinterpret_cast<const float4*>(p.bias + co + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (p.bias_n) {
      const float* q = p.bias_n + (size_t)(m / HoWo) * p.Cout + co;
      const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (p.res) {
      if (p.res_fmt) {
        const half_t* rp = reinterpret_cast<const half_t*>(p.res) + (size_t)m * (2 * p.res_ld) + (co >> 5) * 64 + (co & 31);
        const f16x8 rh = *reinterpret_cast<const f16x8*>(rp), rl = *reinterpret_cast<const f16x8*>(rp + 32);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rh[r] + (float)rl[r];
      } else {
        const float* q = reinterpret_cast<const float*>(p.res) + (size_t)m * p.res_ld + co;
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = act_p<ACT>(v[r]);
    const bool second = p.out2 && co >= p.split;
    const int oc = second ? co - p.split : co;
    const int old_ = second ? p.out2_ld : p.out_ld;
    void* const obase = second ? p.out2 : p.out;
    if (out_fmt) {
      half_t* op = reinterpret_cast<half_t*>(obase) + (size_t)m * (2 * old_) + (oc >> 5) * 64 + (oc & 31);
      f16x8 h, lo;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        h[r] = (half_t)v[r];
        lo[r] = (half_t)(v[r] - (float)h[r]);
      }
      *reinterpret_cast<f16x8*>(op) = h;
      *reinterpret_cast<f16x8*>(op + 32) = lo;
    } else {
      float* op = reinterpret_cast<float*>(obase) + (size_t)m * old_ + oc;
      *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

}  // namespace

bool conv16x3p_supported(const Conv32& p) {
  const int KTOT = p.KH * p.KW * (p.Cin / KS);
  return p.x3 && p.groups <= 1 && p.Cout % 256 == 0 && p.Cin % KS == 0 && KTOT >= 4 && p.ps_cout == 0 && !p.in2 && !p.head_w &&
         (int64_t)p.N * p.H * p.W < (1ll << 31) && (int64_t)p.Cout * p.KH * p.KW * p.Cin * 4 < (1ll << 31);
}

int64_t x3p_image_halfs(int Cout, int K) { return (Cout % 256 == 0 && K % KS == 0) ? (int64_t)2 * Cout * K : 0; }

// the K-walk group (K steps of 32 channels per tap) of a layer in the network: groups of 128 channels -- the nine taps of a 3x3
// re-read a group's slice of the map (1/16 of layer4's p5) while it is still in L2 / the Infinity Cache instead of coming back to the
// whole 537 MB map nine times.  Whole fp16x3 step at batch 16, same box: tap-major 554 | 256-channel groups 552 | 128 563-566 |
// 64 564-566 | 32 560 tiles/s.  EMP_X3P_KGROUP=n (A/B; n >= Cin / 32: tap-major)
int x3p_kgroup(int KT, int Cin) {
  const int CB = Cin / KS;
  static const int env_kg = [] { const char* e = getenv("EMP_X3P_KGROUP"); return e ? atoi(e) : 0; }();
  int kg = env_kg > 0 ? env_kg : 4;
  if (KT == 1 || kg > CB || CB % kg != 0) kg = CB;
  return kg;
}

// KT, Cin: the taps and channels of the main source (K == KT * Cin); kg: the walk group the kernel will be launched with (0: x3p_kgroup)
int launch_x3p_pack(const float* w, half_t* out, int Cout, int K, hipStream_t s, int KT, int Cin, int kg) {
  EMP_REQUIRE(w && out && Cout > 0 && Cout % 256 == 0 && K > 0 && K % KS == 0, "x3p_pack: Cout %% 256 and K %% 32 must be 0 (got %d, %d)", Cout, K);
  if (KT <= 0) { KT = 1; Cin = K; }
  EMP_REQUIRE(KT * Cin == K && Cin % KS == 0, "x3p_pack: K = %d is not %d taps of %d channels", K, KT, Cin);
  if (kg <= 0) kg = x3p_kgroup(KT, Cin);
  EMP_REQUIRE((Cin / KS) % kg == 0, "x3p_pack: the walk group must divide the channel blocks");
  const int64_t chunks = (int64_t)Cout * K / 4;
  hipLaunchKernelGGL(x3p_pack_kernel, dim3((unsigned)std::min<int64_t>((chunks + 255) / 256, 8192)), dim3(256), 0, s, w, out, Cout, K, KT, Cin, kg);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

int launch_hl32_from_f32(const float* in, half_t* out, int64_t rows, int C, int in_ld, int out_ld, hipStream_t s) {
  EMP_REQUIRE(in && out && rows > 0 && C > 0 && C % 32 == 0 && in_ld % 4 == 0 && out_ld % 32 == 0 && in_ld >= C && out_ld >= C &&
                  ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "hl32_from_f32: C, out_ld %% 32 and 16-byte alignment required");
  const int64_t n = rows * (C / 8);
  hipLaunchKernelGGL(hl32_from_f32_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, in, out, rows, C, in_ld, out_ld);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}
int launch_hl32_to_f32(const half_t* in, float* out, int64_t rows, int C, int in_ld, int out_ld, hipStream_t s) {
  EMP_REQUIRE(in && out && rows > 0 && C > 0 && C % 32 == 0 && in_ld % 32 == 0 && out_ld % 4 == 0 && in_ld >= C && out_ld >= C &&
                  ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "hl32_to_f32: C, in_ld %% 32 and 16-byte alignment required");
  const int64_t n = rows * (C / 8);
  hipLaunchKernelGGL(hl32_to_f32_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65535)), dim3(256), 0, s, in, out, rows, C, in_ld, out_ld);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}
int launch_avgpool_hl32(const half_t* in, int N, int HW, int C, int in_ld, float* out, hipStream_t s) {
  EMP_REQUIRE(in && out && N > 0 && HW > 0 && C > 0 && C % 32 == 0 && in_ld % 32 == 0, "avgpool_hl32: bad arguments");
  hipLaunchKernelGGL(avgpool_hl32_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)N), dim3(1024), 0, s, in, HW, C, in_ld, out);
  EMP_LAUNCH_CHECK();
  return EMP_OK;
}

// Conv32 contract (ref32.hip) with p.in an hl32 map and p.wimgp the packed image; p.out_fmt / p.res_fmt: 0 fp32 rows, 1 hl32 rows
int launch_conv16x3p(const Conv32& c, hipStream_t s) {
  EMP_REQUIRE(conv16x3p_supported(c), "conv16x3p: unsupported shape (Cout=%d Cin=%d K steps=%d)", c.Cout, c.Cin, c.KH * c.KW * (c.Cin / KS));
  EMP_REQUIRE(c.wimgp && c.in && c.out && c.in_fmt == 1, "conv16x3p: needs an hl32 input map and a packed weight image");
  EMP_REQUIRE(c.in_ld % 32 == 0 && ((uintptr_t)c.in % 128) == 0 && c.in_ld >= c.Cin, "conv16x3p: hl32 input rows must be whole 128-byte blocks");
  EMP_REQUIRE(c.out_fmt ? (c.out_ld % 32 == 0 && ((uintptr_t)c.out % 128) == 0) : (c.out_ld % 4 == 0 && ((uintptr_t)c.out % 16) == 0),
              "conv16x3p: misaligned output (out_ld=%d)", c.out_ld);
  EMP_REQUIRE(!c.res || (c.res_fmt ? (c.res_ld % 32 == 0 && ((uintptr_t)c.res % 128) == 0) : (c.res_ld % 4 == 0 && ((uintptr_t)c.res % 16) == 0)),
              "conv16x3p: misaligned residual");
  EMP_REQUIRE((!c.bias || ((uintptr_t)c.bias % 16) == 0) && (!c.bias_n || ((uintptr_t)c.bias_n % 16) == 0), "conv16x3p: misaligned bias");
  X3P p{};
  p.in = reinterpret_cast<const half_t*>(c.in); p.in_ld = c.in_ld;
  p.wimg = c.wimgp;
  p.bias = c.bias; p.bias_n = c.bias_n;
  p.res = c.res; p.res_ld = c.res_ld; p.res_fmt = c.res_fmt;
  p.out = c.out; p.out_ld = c.out_ld;
  if (c.out2) {
    EMP_REQUIRE(c.split2 > 0 && c.split2 % 256 == 0 && c.split2 < c.Cout && !c.bias_n && !c.res &&
                    (c.out_fmt ? (c.out2_ld % 32 == 0 && ((uintptr_t)c.out2 % 128) == 0) : (c.out2_ld % 4 == 0 && ((uintptr_t)c.out2 % 16) == 0)),
                "conv16x3p: a second destination takes whole 256-cout tiles (split=%d)", c.split2);
    p.out2 = c.out2; p.out2_ld = c.out2_ld; p.split = c.split2;
  }
  p.zero = reinterpret_cast<const half_t*>(zero_page());
  EMP_REQUIRE(p.zero != nullptr, "conv16x3p: no zero page");
  p.N = c.N; p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.Cout = c.Cout; p.KH = c.KH; p.KW = c.KW; p.stride = c.stride; p.pad = c.pad; p.dil = c.dil;
  p.Ho = c.Ho; p.Wo = c.Wo;
  const int64_t M = (int64_t)c.N * c.Ho * c.Wo;
  EMP_REQUIRE(M < (1ll << 31), "conv16x3p: too many output pixels");
  p.M = (int)M;
  p.mt = cdiv(p.M, 256);
  p.nt = c.Cout / 256;
  p.mt_per_xcd = cdiv(p.mt, 8);
  p.kg = c.x3p_kg > 0 ? c.x3p_kg : c.Cin / KS;      // (the image was packed along this walk)
  EMP_REQUIRE((c.Cin / KS) % p.kg == 0, "conv16x3p: the walk group must divide the channel blocks");
  const int grid = 8 * p.mt_per_xcd * p.nt;
  const int act = c.act;
  // split-K (Conv32::kpart): a long-K launch of a few pixel tiles -- the merged 3x3 ASPP branches of ONE 1024^2 tile are 32 workgroups
  // over 576 K steps -- as S workgroups per tile while S * tiles fit the chip (one workgroup per CU), every split >= 8 steps; raw partial sums to the scratch, x3p_finish_kernel does the epilogue
  if (c.kpart) {
    const int KTOT = c.KH * c.KW * (c.Cin / KS);
    const int unit = 1;      // (a split may start anywhere in the walk)
    int S = std::min(std::min(8, 256 / std::max(p.mt * p.nt, 1)), KTOT / 8);
    S = S >= 8 ? 8 : (S >= 4 ? 4 : (S >= 2 ? 2 : 0));      // (the kernel's raster: a split per 8 / S XCDs)
    while (S >= 2) {
      const int per = cdiv(cdiv(KTOT, S), unit) * unit;
      const int last = KTOT - (S - 1) * per;
      if (per < 8 || last < 4) { S >>= 1; continue; }      // (every split walks >= 4 steps, the last one what is left)
      break;
    }
    if (S >= 2) {
      const int per = cdiv(cdiv(KTOT, S), unit) * unit;
      if ((int64_t)S * M * c.Cout * 4 <= c.kpart_bytes && ((uintptr_t)c.kpart % 16) == 0) {
        X3P q = p;
        q.bias = nullptr; q.bias_n = nullptr; q.res = nullptr; q.out2 = nullptr; q.split = 0;
        q.out = c.kpart; q.out_ld = c.Cout;
        q.ksteps = per; q.ksplit = S;
        auto kern = &conv16x3p_kernel<0, 0, 0, true>;
        if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), X3P_LDS)) return rc;
        hipLaunchKernelGGL(kern, dim3(8 * cdiv(p.mt * p.nt * S, 8)), dim3(512), X3P_LDS, s, q);
        EMP_LAUNCH_CHECK();
        const int64_t total = M * (c.Cout / 8);
        const dim3 fg((unsigned)std::min<int64_t>((total + 255) / 256, 8192));
        if (act == 1) hipLaunchKernelGGL(x3p_finish_kernel<1>, fg, dim3(256), 0, s, p, c.kpart, S, c.out_fmt, total);
        else if (act == 2) hipLaunchKernelGGL(x3p_finish_kernel<2>, fg, dim3(256), 0, s, p, c.kpart, S, c.out_fmt, total);
        else hipLaunchKernelGGL(x3p_finish_kernel<0>, fg, dim3(256), 0, s, p, c.kpart, S, c.out_fmt, total);
        EMP_LAUNCH_CHECK();
        return EMP_OK;
      }
    }
  }
  auto go = [&](auto kern) -> int {
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(kern), X3P_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), X3P_LDS, s, p);
    EMP_LAUNCH_CHECK();
    return EMP_OK;
  };
  const int resf = !c.res ? 0 : (c.res_fmt ? 2 : 1);
  auto pick = [&](auto A, auto O) -> int {
    constexpr int a = decltype(A)::value, o = decltype(O)::value;
    if (resf == 2) return go(&conv16x3p_kernel<a, o, 2>);
    if (resf == 1) return go(&conv16x3p_kernel<a, o, 1>);
    return go(&conv16x3p_kernel<a, o, 0>);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (c.out_fmt) {
    if (act == 1) return pick(I1{}, I1{});
    if (act == 2) return pick(I2{}, I1{});
    return pick(I0{}, I1{});
  }
  if (act == 1) return pick(I1{}, I0{});
  if (act == 2) return pick(I2{}, I0{});
  return pick(I0{}, I0{});
}

}  // namespace emp
