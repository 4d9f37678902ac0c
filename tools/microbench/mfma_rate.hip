// Sustained fp16 MFMA rate of the whole chip under its power limit, registers only: v_mfma_f32_16x16x32_f16 (what every
// kernel of the library issues) against v_mfma_f32_32x32x16_f16 (twice the FLOPs per operand register read), with random
// and with zero operands (data-dependent switching power), ~0.4 s per case so the clock settles.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const f16x8* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  // four independent accumulator chains per wave for each shape; operands stay in registers
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 1023]; b[i] = src[(threadIdx.x + 64 * i + 256) & 1023]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (MODE == 0) {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)      // in place (the builtin form made the compiler rotate the accumulators through v_accvgpr moves)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[i & 3]), "v"(b[(i + (i >> 2)) & 3]));
    }
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
  } else {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[i]), "v"(b[(i + 1) & 3]));
    }
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const f16x8* src, int wgs_per_cu) {
  float* out; unsigned long long *clk, h[2];
  hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 16);
  const int grid = 256 * wgs_per_cu, iters = 4000000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters, clk);      // warm: clocks settle
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters, clk);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // FLOPs per wave and iteration: MODE 0: 8 x 16*16*32*2, MODE 1: 4 x 32*32*16*2 (the same 131072)
  const double flops = (double)grid * 4 * iters * 131072.0;
  printf("%-34s %d WG/CU: %7.1f ms  %7.1f TFLOP/s   shader clock %.2f GHz (s_memtime / s_memrealtime @100 MHz)\n", name, wgs_per_cu, ms,
         flops / ms / 1e9, (double)h[0] / ((double)h[1] / 100e6) / 1e9);
  hipFree(out); hipFree(clk);
}

int main() {
  f16x8* src; f16x8 h[1024];
  hipMalloc(&src, sizeof(h));
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (int i = 0; i < 1024; ++i)
      for (int j = 0; j < 8; ++j) h[i][j] = zero ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    printf("operands: %s\n", zero ? "zeros" : "random in [-1, 1]");
    for (int w : {1, 2}) {
      run<0>("v_mfma_f32_16x16x32_f16", src, w);
      run<1>("v_mfma_f32_32x32x16_f16", src, w);
    }
  }
  return 0;
}
