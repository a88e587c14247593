// fp32 MFMA rate with operand values that toggle (random per lane / per step) vs constant operands: power-limited clocks?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return (float)(x & 0xffffff) / 8388608.f - 1.f; }
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a[8], b[8];
  for (int u = 0; u < 8; ++u) {
    if (MODE == 0) { a[u] = 1.f; b[u] = 2.f; }                       // constant everywhere
    else if (MODE == 1) { a[u] = rnd(threadIdx.x * 8 + u); b[u] = rnd(threadIdx.x * 8 + u + 77777); }   // random per lane and step
    else { a[u] = rnd(u) ; b[u] = rnd(u + 99); }                      // random per step, same in all lanes
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[(u + 1) & 7], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
template <int MODE>
void run(int iters) {
  float *out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4;
  k<MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flops = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
  printf("mode %d iters %d: %.3f ms  %.1f TFLOP/s\n", MODE, iters, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<0>(1000); run<1>(1000); run<2>(1000); run<0>(1000); run<1>(1000); run<1>(10000); run<0>(10000);
  return 0;
}
