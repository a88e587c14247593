// Same-wave MFMA issue rate: one wave per SIMD (1 block of 256 threads per CU) against two, for v_mfma_f32_32x32x2_f32
// (16 passes) and v_mfma_f32_16x16x4_f32 (8 passes), 16 independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k32(float *out, int iters, float a0, float b0) {
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) k16(float *out, int iters, float a0, float b0) {
  f32x4 acc[64];
  for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 64; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
template <typename K>
void run(const char *name, K kern, int blocks_per_cu, int iters, double flop_per_iter_wave) {
  float *out; (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  kern<<<grid, 256>>>(out, iters, 1.f, 2.f);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(out, iters, 1.f, 2.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flops = (double)grid * 4 * iters * flop_per_iter_wave;
  printf("%s blocks/CU %d: %.3f ms  %.1f TFLOP/s (%.3f)\n", name, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  (void)hipFree(out);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run("32x32x2 ", k32, 1, 1000, 64 * 4096.0);
    run("32x32x2 ", k32, 2, 1000, 64 * 4096.0);
    run("16x16x4 ", k16, 1, 1000, 128 * 2048.0);
    run("16x16x4 ", k16, 2, 1000, 128 * 2048.0);
  }
  return 0;
}
