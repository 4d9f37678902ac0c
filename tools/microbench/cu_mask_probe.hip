// CU-masked streams on MI355X (hipExtStreamCreateWithCUMask): where the bits of a mask land, and what an MFMA-bound loop and
// an HBM-bound stream cost on a subset of the chip, alone and side by side (the abstract form of its
// tables (a) / (b) / (c): no library kernel involved; tools/cu_partition.py runs the library's own launches on such streams).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
// Masks are built so that EVERY XCD keeps at least one CU of every stream: bit i of the mask is CU (i / 8) of XCD (i % 8)
// (checked below against HW_REG_XCC_ID / HW_REG_HW_ID), and a queue whose mask leaves an XCD without a CU is not probed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) where_kernel(unsigned* out, int spin) {
  __shared__ char pad[60 * 1024];      // two workgroups per CU at most
  pad[threadIdx.x] = 0;
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);      // 100 MHz ticks
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw + pad[1]; }
}

// MFMA-bound: registers only, eight accumulator chains per wave (tools/microbench/mfma_rate.hip)
__global__ void __launch_bounds__(256) mfma_kernel(const f16x8* __restrict__ src, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x + 64 * i + 256) & 1023]; }
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[i & 3]), "v"(b[(i + (i >> 2)) & 3]));
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// HBM-bound: grid-stride copy, 16 B per lane, four loads in flight
__global__ void __launch_bounds__(256) copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i + 768 < n; i += stride) {
    uint4 v0 = src[i], v1 = src[i + 256], v2 = src[i + 512], v3 = src[i + 768];
    dst[i] = v0; dst[i + 256] = v1; dst[i + 512] = v2; dst[i + 768] = v3;
  }
}

// fp32 VALU-bound: 8 independent fma chains per lane (the separable blocks' depthwise half is of this kind)
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0000001f, c = 1e-7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static const int NCU = 256, NXCD = 8, CU_PER_XCD = 32;

// CUs [lo, hi) of EVERY XCD
static std::vector<uint32_t> mask_range(int lo, int hi) {
  std::vector<uint32_t> m(NCU / 32, 0u);
  for (int i = 0; i < NCU; ++i)
    if (i / NXCD >= lo && i / NXCD < hi) m[i / 32] |= 1u << (i % 32);
  return m;
}

static hipStream_t masked_stream(int lo, int hi) {
  hipStream_t s;
  std::vector<uint32_t> m = mask_range(lo, hi);
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
  return s;
}

static float timed(hipStream_t s, void (*launch)(hipStream_t, void*), void* arg, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(s, arg);
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) launch(s, arg);
  CK(hipEventRecord(e1, s));
  CK(hipStreamSynchronize(s));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms / reps;
}

struct Bufs { f16x8* src; float* out; uint4 *ca, *cb; size_t copy_n; int mfma_iters; int mfma_blocks; int copy_blocks; int valu_iters; };
static void l_mfma(hipStream_t s, void* p) { Bufs* b = (Bufs*)p; mfma_kernel<<<b->mfma_blocks, 256, 0, s>>>(b->src, b->out, b->mfma_iters); }
static void l_copy(hipStream_t s, void* p) { Bufs* b = (Bufs*)p; copy_kernel<<<b->copy_blocks, 256, 0, s>>>(b->ca, b->cb, b->copy_n); }
static void l_valu(hipStream_t s, void* p) { Bufs* b = (Bufs*)p; valu_kernel<<<b->mfma_blocks, 256, 0, s>>>(b->out, b->valu_iters); }

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  if (prop.multiProcessorCount != NCU) { printf("expected %d CUs\n", NCU); return 1; }

  // ---- 1. where do the bits land? ----
  const int NB = 4096;
  unsigned* where;
  CK(hipMalloc(&where, NB * 2 * sizeof(unsigned)));
  std::vector<unsigned> h(NB * 2);
  const int cuts[][2] = {{0, 32}, {0, 8}, {8, 32}, {0, 16}, {16, 32}, {0, 1}, {31, 32}};
  for (auto& c : cuts) {
    hipStream_t s = masked_stream(c[0], c[1]);
    where_kernel<<<NB, 256, 0, s>>>(where, 2000);      // 20 us per workgroup
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), where, NB * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < NB; ++b) {
      const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
      per_xcc[xcc].insert((hw >> 8) & 0xff);      // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
    }
    int total = 0;
    printf("mask CUs [%2d, %2d) of every XCD (%3d bits): ", c[0], c[1], (c[1] - c[0]) * NXCD);
    for (auto& kv : per_xcc) { printf("xcc%u:%zu ", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf(" -> %d distinct CUs seen\n", total);
    CK(hipStreamDestroy(s));
  }

  // ---- 2. MFMA loop / VALU loop / HBM copy on a subset of the chip ----
  Bufs b;
  CK(hipMalloc(&b.src, 1024 * sizeof(f16x8)));
  std::vector<_Float16> hs(1024 * 8);
  srand(1);
  for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  CK(hipMemcpy(b.src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
  b.mfma_blocks = 256 * 8;        // 8 workgroups per CU of the whole chip (2 waves per SIMD resident, x4 rounds)
  CK(hipMalloc(&b.out, (size_t)b.mfma_blocks * 256 * 4));
  b.mfma_iters = 20000;           // 8 x 16x16x32 MFMAs per iteration and wave
  b.valu_iters = 40000;
  b.copy_n = (size_t)(2u << 30) / 16;      // 2 GiB read + 2 GiB written per launch
  b.copy_blocks = 256 * 16;
  CK(hipMalloc(&b.ca, b.copy_n * 16)); CK(hipMalloc(&b.cb, b.copy_n * 16));
  CK(hipMemset(b.ca, 1, b.copy_n * 16));
  const double mfma_flop = (double)b.mfma_blocks * 4 /*waves*/ * b.mfma_iters * 8 * (2.0 * 16 * 16 * 32);
  const double valu_flop = (double)b.mfma_blocks * 256 * b.valu_iters * 8 * 2.0;
  const double copy_bytes = (double)b.copy_n * 16 * 2;

  printf("\n(a) alone on CUs [lo, 32) of every XCD: MFMA loop (random operands) / fp32 VALU loop\n");
  printf("%8s %12s %14s %12s %14s\n", "CUs", "MFMA TF/s", "TF/s per CU", "VALU TF/s", "GF/s per CU");
  for (int lo : {0, 4, 8, 12, 16}) {
    hipStream_t s = masked_stream(lo, 32);
    const int n = (32 - lo) * NXCD;
    const float ms = timed(s, l_mfma, &b, 3), mv = timed(s, l_valu, &b, 3);
    printf("%8d %12.1f %14.2f %12.1f %14.1f\n", n, mfma_flop / ms / 1e9, mfma_flop / ms / 1e9 / n, valu_flop / mv / 1e9, valu_flop / mv / 1e6 / n);
    CK(hipStreamDestroy(s));
  }
  printf("\n(b) alone on CUs [0, hi) of every XCD: copy (2 GiB read + 2 GiB written)\n");
  printf("%8s %12s\n", "CUs", "GB/s r+w");
  for (int hi : {2, 4, 8, 12, 16, 32}) {
    hipStream_t s = masked_stream(0, hi);
    const float ms = timed(s, l_copy, &b, 3);
    printf("%8d %12.1f\n", hi * NXCD, copy_bytes / ms / 1e6);
    CK(hipStreamDestroy(s));
  }
  printf("\n(c) side by side: MFMA loop on CUs [c, 32), copy on CUs [0, c) of every XCD, both streams kept busy for the same wall time\n");
  printf("%8s %8s %12s %12s %16s\n", "MFMA CUs", "copy CUs", "MFMA TF/s", "copy GB/s", "(unmasked pair)");
  for (int c : {0, 2, 4, 8, 12, 16}) {
    hipStream_t sa = c ? masked_stream(c, 32) : masked_stream(0, 32);
    hipStream_t sb = c ? masked_stream(0, c) : masked_stream(0, 32);
    // warm
    l_mfma(sa, &b); l_copy(sb, &b);
    CK(hipDeviceSynchronize());
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    const int ra = 6, rb = 12;
    CK(hipEventRecord(a0, sa)); CK(hipEventRecord(b0, sb));
    for (int r = 0; r < rb; ++r) { if (r < ra) l_mfma(sa, &b); l_copy(sb, &b); }
    CK(hipEventRecord(a1, sa)); CK(hipEventRecord(b1, sb));
    CK(hipDeviceSynchronize());
    float ma, mb;
    CK(hipEventElapsedTime(&ma, a0, a1)); CK(hipEventElapsedTime(&mb, b0, b1));
    printf("%8d %8d %12.1f %12.1f %16s   (MFMA stream busy %.1f ms, copy stream busy %.1f ms: the rates overlap for the shorter of the two)\n",
           c ? (32 - c) * NXCD : 256, c ? c * NXCD : 256, mfma_flop * ra / ma / 1e9, copy_bytes * rb / mb / 1e6, c ? "" : "both unmasked", ma, mb);
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  }
  return 0;
}
