// Pure-write rate against the store pattern (how many 16-byte stores a thread has in flight, how they are laid out,
// block and grid size): hipcc --offload-arch=gfx950 -O3 -o /tmp/fill tools/microbench/fill_patterns.hip && /tmp/fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// U stores per thread and iteration; LAYOUT 0: thread's U chunks are consecutive (64 B * U per thread),
// LAYOUT 1: chunk u of all threads of a block forms one contiguous run (coalesced per instruction)
template <int U, int LAYOUT, int NT>
__global__ void __launch_bounds__(1024) fill(u32x4* __restrict__ dst, size_t n) {
  const size_t per_iter = (size_t)gridDim.x * blockDim.x * U;
  for (size_t base = 0; base < n; base += per_iter) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t i;
      if (LAYOUT == 0) i = base + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * U + u;
      else i = base + ((size_t)blockIdx.x * U + u) * blockDim.x + threadIdx.x;
      if (i < n) {
        const u32x4 v = u32x4{1u, 2u, 3u, (unsigned)i};
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
      }
    }
  }
}
template <int U, int LAYOUT, int NT> void run(u32x4* dst, size_t n, int block, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((fill<U, LAYOUT, NT>), dim3(grid), dim3(block), 0, 0, dst, n);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((fill<U, LAYOUT, NT>), dim3(grid), dim3(block), 0, 0, dst, n);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("U=%d layout=%d nt=%d block=%4d grid=%6d : %.3f ms  %.2f TB/s\n", U, LAYOUT, NT, block, grid, ms, n * 16 / ms / 1e9);
}
int main() {
  const size_t n = (size_t)1 << 26;   // 1 GiB
  u32x4* dst; hipMalloc(&dst, n * 16);
  hipMemset(dst, 0, n * 16); hipDeviceSynchronize();
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipMemsetAsync(dst, 1, n * 16, 0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5; printf("hipMemsetAsync 1 GiB: %.3f ms  %.2f TB/s\n", ms, n * 16 / ms / 1e9);
  }
  for (int block : {256, 1024})
    for (int grid : {256 * 4, 256 * 16, 256 * 64}) {
      run<1, 1, 0>(dst, n, block, grid);
      run<4, 1, 0>(dst, n, block, grid);
      run<4, 0, 0>(dst, n, block, grid);
      run<8, 1, 0>(dst, n, block, grid);
      run<4, 1, 1>(dst, n, block, grid);
    }
  // one block per 16 KiB, no grid-stride loop (every workgroup writes once and exits)
  run<4, 1, 0>(dst, n, 256, (int)(n / (256 * 4)));
  run<4, 1, 1>(dst, n, 256, (int)(n / (256 * 4)));
  run<1, 1, 0>(dst, n, 256, (int)(n / 256));
  return 0;
}
