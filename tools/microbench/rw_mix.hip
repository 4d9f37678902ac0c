// Write rate of a kernel that also reads (DESIGN.md finding 10): dst[i] = src[i % n_src] with plain / non-temporal /
// cache-policy-tagged stores.   hipcc --offload-arch=gfx950 -O3 -o /tmp/rw_mix tools/microbench/rw_mix.hip && /tmp/rw_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n_dst, size_t n_src) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dst; i += (size_t)gridDim.x * 256) {
    const size_t j = n_src ? i % n_src : 0;
    u32x4 v;
    if (n_src == 0) v = u32x4{1u, 2u, 3u, (unsigned)i};
    else if (MODE == 2 || MODE == 5) v = __builtin_nontemporal_load(src + j);
    else v = src[j];
    u32x4* d = dst + i;
    if (MODE == 0) *d = v;
    if (MODE == 1 || MODE == 2) __builtin_nontemporal_store(v, d);
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(d), "v"(v) : "memory");
    if (MODE == 4 || MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(d), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(d), "v"(v) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(d), "v"(v) : "memory");
  }
}
template <int MODE> void run(const char* name, u32x4* src, u32x4* dst, size_t n_dst, size_t n_src) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 16;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n_dst, n_src);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n_dst, n_src);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-28s src %5zu MB: %.3f ms  write %.2f TB/s  read %.2f TB/s\n", name, n_src * 16 >> 20, ms, n_dst * 16 / ms / 1e9,
         (n_src ? n_dst * 16 : 0) / ms / 1e9);
}
int main() {
  const size_t n_dst = (size_t)1 << 26;   // 1 GiB
  u32x4 *src, *dst; hipMalloc(&src, n_dst * 16); hipMalloc(&dst, n_dst * 16); hipMemset(src, 1, n_dst * 16);
  for (size_t n_src : {(size_t)0, n_dst / 64, n_dst / 4, n_dst}) {
    run<0>("plain", src, dst, n_dst, n_src);
    run<1>("nt store", src, dst, n_dst, n_src);
    run<2>("nt store + nt load", src, dst, n_dst, n_src);
    run<3>("store sc0 sc1", src, dst, n_dst, n_src);
    run<4>("store sc1 nt", src, dst, n_dst, n_src);
    run<5>("store sc1 nt + nt load", src, dst, n_dst, n_src);
    run<6>("store sc0 sc1 nt", src, dst, n_dst, n_src);
    run<7>("store sc0", src, dst, n_dst, n_src);
  }
  return 0;
}
