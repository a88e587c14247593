// HBM bandwidth calibration for the streaming kernels: what do plain copy / read kernels reach at the sizes of C3/C4?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T, int U>
__global__ void __launch_bounds__(256) copy_k(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i < n; i += stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) out[i + u * 256] = v[u];
  }
}
template <typename T, int U>
__global__ void __launch_bounds__(256) read_k(const T* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  float acc = 0.f;
  for (; i < n; i += stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = in[i + u * 256]; else v[u] = T{};
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1];
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <typename T, int U>
__global__ void __launch_bounds__(256) write_k(T* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  T v; v[0] = 1.f; v[1] = 2.f;
  for (; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) out[i + u * 256] = v;
  }
}
template <typename F> float timeit(F f, int it = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < it; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main(int argc, char** argv) {
  for (size_t mb : {84, 168, 336, 1024}) {
    size_t bytes = mb << 20;
    float *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    printf("---- %zu MiB per buffer\n", mb);
    for (int blocks : {1024, 2048, 4096, 8192, 0}) {
      size_t n4 = bytes / 16, n2 = bytes / 8;
      int b4u1 = blocks ? blocks : (int)((n4 + 255) / 256), b4u4 = blocks ? blocks : (int)((n4 + 1023) / 1024);
      int b2u4 = blocks ? blocks : (int)((n2 + 1023) / 1024);
      float t;
      t = timeit([&] { copy_k<f4, 1><<<b4u1, 256>>>((f4*)in, (f4*)out, n4); });
      printf("copy f4 U1 blocks %6d: %7.1f us %6.2f TB/s\n", b4u1, t * 1e3, 2.0 * bytes / t / 1e9);
      t = timeit([&] { copy_k<f4, 4><<<b4u4, 256>>>((f4*)in, (f4*)out, n4); });
      printf("copy f4 U4 blocks %6d: %7.1f us %6.2f TB/s\n", b4u4, t * 1e3, 2.0 * bytes / t / 1e9);
      t = timeit([&] { copy_k<f2, 4><<<b2u4, 256>>>((f2*)in, (f2*)out, n2); });
      printf("copy f2 U4 blocks %6d: %7.1f us %6.2f TB/s\n", b2u4, t * 1e3, 2.0 * bytes / t / 1e9);
      t = timeit([&] { read_k<f4, 4><<<b4u4, 256>>>((f4*)in, out, n4); });
      printf("read f4 U4 blocks %6d: %7.1f us %6.2f TB/s\n", b4u4, t * 1e3, 1.0 * bytes / t / 1e9);
      t = timeit([&] { read_k<f4, 8><<<blocks ? blocks : (int)((n4 + 2047) / 2048), 256>>>((f4*)in, out, n4); });
      printf("read f4 U8 blocks %6d: %7.1f us %6.2f TB/s\n", blocks, t * 1e3, 1.0 * bytes / t / 1e9);
      t = timeit([&] { write_k<f4, 4><<<b4u4, 256>>>((f4*)out, n4); });
      printf("write f4 U4 blocks %6d: %7.1f us %6.2f TB/s\n", b4u4, t * 1e3, 1.0 * bytes / t / 1e9);
    }
    CK(hipFree(in)); CK(hipFree(out));
  }
  return 0;
}
