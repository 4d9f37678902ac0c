#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  float a[16]; f32x2 p[16];
  f16x2 h0 = {(_Float16)seed, (_Float16)(seed * 0.5f)}, h1 = {(_Float16)(seed + 1.f), (_Float16)0.25f};
  f32x2 q0 = {seed, seed * 0.5f}, q1 = {seed + 1.f, 0.25f};
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.f}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 512; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(seed), "v"(q1[0]));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q0), "v"(q1));
      if (MODE == 2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(h0), "v"(h1));
      if (MODE == 3) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(h0), "v"(h1));
      if (MODE == 4) asm volatile("v_perm_b32 %0, %1, %2, %3" : "+v"(a[i]) : "v"(seed), "v"(q1[0]), "v"(0x05040100));
      if (MODE == 5) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(a[i]) : "v"(h0));
      if (MODE == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[i]) : "v"(h0), "v"(h1));
      if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(p[i]) : "v"(q0), "v"(q1));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int waves) {
  float* out; unsigned long long* cyc, h;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize(); }
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-18s waves/WG %2d: %.2f cycles per instruction per wave (%.2f per SIMD-instr)\n", name, waves, (double)h / (512 * 16), (double)h / (512 * 16) / ((waves + 3) / 4));
}
template <int MODE> void wall(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 8);
  const int grid = 256 * 2048 / threads;      // 2048 threads (32 waves) per CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 0, 0, out, cyc, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double instr = (double)grid * (threads / 64) * 512 * 16;
  printf("%-18s full chip, %4d threads/WG: %.3f ms, %.2f G wave-instr/s = %.3f wave-instr per SIMD per ns\n", name, threads, ms, instr / ms / 1e6, instr / ms / 1e6 / 1024);
}
int main() {
  wall<0>("v_fma_f32", 256); wall<1>("v_pk_fma_f32", 256); wall<2>("v_dot2_f32_f16", 256); wall<5>("v_cvt_f32_f16", 256); wall<4>("v_perm_b32", 256);
  wall<1>("v_pk_fma_f32", 1024);

  for (int w : {1, 4, 8, 12, 16}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_dot2_f32_f16", w); run<3>("v_dot2c_f32_f16", w);
    run<4>("v_perm_b32", w); run<5>("v_cvt_f32_f16", w); run<6>("v_fma_mix_f32", w); run<7>("v_pk_mul_f32", w);
  }
  return 0;
}
