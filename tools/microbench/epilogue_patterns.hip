// Store patterns of a conv epilogue writing a [M][256] fp16 map (512-byte pixel rows, 1 GiB): per wave instruction
// 16 rows x 64 B (the direct MFMA-layout epilogue), 8 rows x 128 B (LDS-transposed, 128x128 kernel), 2 rows x 512 B,
// with / without reading a residual of the same shape the same way.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ep tools/microbench/epilogue_patterns.hip && /tmp/ep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// a wave owns a block of 128 rows x RUN bytes... generalized: each wave instruction covers ROWS rows x (64/ROWS) lanes x 16 B
template <int ROWS, bool RES>
__global__ void __launch_bounds__(256) epi(const char* __restrict__ res, char* __restrict__ out, int M) {
  constexpr int LPR = 64 / ROWS;           // lanes per row
  constexpr int RUN = LPR * 16;            // contiguous bytes per row and instruction
  constexpr int SEG = 512 / RUN;           // instructions (column segments) per row
  const int l = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  // wave -> 128 rows x 128 B column block (as conv256: 64 couts); within it instructions walk rows then segments
  const int nwaves_per_rowblock = 4;       // 4 x 128 B = 512 B row
  const int rb = wave / nwaves_per_rowblock, cb = wave % nwaves_per_rowblock;
  if (rb * 128 >= M) return;
  if (RUN <= 128) {
    // column block of 128 B: instructions cover ROWS rows x RUN bytes
    for (int r0 = 0; r0 < 128; r0 += ROWS)
      for (int s = 0; s < 128 / RUN; ++s) {
        const size_t off = ((size_t)rb * 128 + r0 + l / LPR) * 512 + cb * 128 + s * RUN + (l % LPR) * 16;
        u32x4 v = u32x4{1u, 2u, 3u, (unsigned)off};
        if (RES) v += *reinterpret_cast<const u32x4*>(res + off);
        *reinterpret_cast<u32x4*>(out + off) = v;
      }
  } else {
    // whole rows: the wave owns 32 full rows (same bytes per wave: 128 rows x 128 B = 32 rows x 512 B)
    for (int r0 = 0; r0 < 32; r0 += ROWS)
      for (int s = 0; s < SEG; ++s) {
        const size_t off = ((size_t)rb * 128 + cb * 32 + r0 + l / LPR) * 512 + s * RUN + (l % LPR) * 16;
        u32x4 v = u32x4{1u, 2u, 3u, (unsigned)off};
        if (RES) v += *reinterpret_cast<const u32x4*>(res + off);
        *reinterpret_cast<u32x4*>(out + off) = v;
      }
  }
}
template <int ROWS, bool RES> void run(const char* res, char* out, int M) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int waves = M / 128 * 4, grid = waves / 4;
  hipLaunchKernelGGL((epi<ROWS, RES>), dim3(grid), dim3(256), 0, 0, res, out, M);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((epi<ROWS, RES>), dim3(grid), dim3(256), 0, 0, res, out, M);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%2d rows x %4d B per instruction, residual %d: %.3f ms  write %.2f TB/s\n", ROWS, 64 / ROWS * 16, (int)RES, ms,
         (double)M * 512 / ms / 1e9);
}
int main() {
  const int M = 1 << 21;
  char *res, *out; hipMalloc(&res, (size_t)M * 512); hipMalloc(&out, (size_t)M * 512); hipMemset(res, 1, (size_t)M * 512);
  run<16, false>(res, out, M); run<8, false>(res, out, M); run<2, false>(res, out, M);
  run<16, true>(res, out, M); run<8, true>(res, out, M); run<2, true>(res, out, M);
  return 0;
}
