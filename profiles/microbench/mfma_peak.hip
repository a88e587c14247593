// Sustained fp32 MFMA rate on the whole chip: v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  float *out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flops = (double)grid * 4 /*waves*/ * iters * 4 * NACC * (2.0 * 32 * 32 * 2);
  printf("NACC %d blocks/CU %d iters %d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(1, 4000); run<4>(2, 4000); run<4>(3, 4000); run<4>(4, 2000); run<1>(2, 8000); run<2>(2, 8000);
  run<4>(3, 40000);
  run<4>(64, 32); run<4>(64, 64); run<4>(64, 128); run<4>(16, 128); run<4>(256, 32);
  return 0;
}
