// Copy (read + write) rate against the kernel shape: grid-stride loop vs one block per 4/16 KiB, contiguous vs
// 64-byte-strided stores per instruction.  hipcc --offload-arch=gfx950 -O3 -o /tmp/cp tools/microbench/copy_patterns.hip && /tmp/cp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int U, int LAYOUT>
__global__ void __launch_bounds__(256) copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n, size_t nsrc) {
  const size_t per_iter = (size_t)gridDim.x * 256 * U;
  for (size_t base = 0; base < n; base += per_iter) {
    u32x4 v[U];
    size_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (LAYOUT == 0) idx[u] = base + ((size_t)blockIdx.x * 256 + threadIdx.x) * U + u;
      else idx[u] = base + ((size_t)blockIdx.x * U + u) * 256 + threadIdx.x;
      if (idx[u] < n) v[u] = src[idx[u] % nsrc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (idx[u] < n) dst[idx[u]] = v[u];
  }
}
template <int U, int LAYOUT> void run(const u32x4* src, u32x4* dst, size_t n, size_t nsrc, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((copyk<U, LAYOUT>), dim3(grid), dim3(256), 0, 0, src, dst, n, nsrc);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((copyk<U, LAYOUT>), dim3(grid), dim3(256), 0, 0, src, dst, n, nsrc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("U=%d layout=%d grid=%7d src %4zu MB: %.3f ms  write %.2f TB/s (+ read %.2f)\n", U, LAYOUT, grid, nsrc * 16 >> 20, ms,
         n * 16 / ms / 1e9, n * 16 / ms / 1e9);
}
int main() {
  const size_t n = (size_t)1 << 26;   // 1 GiB
  u32x4 *src, *dst; hipMalloc(&src, n * 16); hipMalloc(&dst, n * 16); hipMemset(src, 1, n * 16); hipMemset(dst, 0, n * 16);
  for (size_t nsrc : {n, n / 4, n / 64}) {
    run<1, 1>(src, dst, n, nsrc, 256 * 16);
    run<4, 1>(src, dst, n, nsrc, 256 * 16);
    run<4, 0>(src, dst, n, nsrc, 256 * 16);
    run<1, 1>(src, dst, n, nsrc, (int)(n / 256));
    run<4, 1>(src, dst, n, nsrc, (int)(n / 1024));
    run<4, 0>(src, dst, n, nsrc, (int)(n / 1024));
    run<8, 1>(src, dst, n, nsrc, (int)(n / 2048));
  }
  return 0;
}
