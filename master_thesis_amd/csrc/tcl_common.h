// Shared helpers for the libtcl_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/tcl.h"

namespace tcl {

int set_error(int code, const char *fmt, ...);  // records the thread-local message, returns code
int check_launch(const char *what);             // hipGetLastError -> TCL_ERR_LAUNCH

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hardware facts used for launch sizing (MI355X: 8 XCDs x 32 CUs)
constexpr int kNumXCD = 8;
constexpr int kNumCU = 256;
constexpr int kMaxStreamBlocks = kNumCU * 8;  // grid cap for grid-stride bandwidth kernels

constexpr float kNormEps = 1e-9f;

#define TCL_REQUIRE(cond, code, ...)                         \
  do {                                                       \
    if (!(cond)) return ::tcl::set_error((code), __VA_ARGS__); \
  } while (0)

// Placed after the N global loads of a step, before their first use: asks the machine scheduler to issue all N
// vector-memory reads of the region before anything else.  Without it hipcc pairs every load with its use
// (`load; s_waitcnt vmcnt(0); use`) to save registers, which serialises the memory round trips.
#define TCL_LOADS_FIRST(N) __builtin_amdgcn_sched_group_barrier(0x020, (N), 0)

// Streaming (non-temporal) stores for outputs the producing kernel never reads back: the lines bypass the L2's
// write-allocate path instead of evicting operands (measured: ref_importance 114 -> 76 us, 3.8 -> 5.7 TB/s).
// (A/B of round 1: align_set fwd 4.9 -> 5.7 TB/s, context matching fwd + ref_importance 0.302 -> 0.242 ms.)  Non-temporal LOADS of once-read operands (grids,
// incoming gradients) were measured too: 7 % slower in the warp forward, noise elsewhere, so loads stay plain.
typedef float tcl_f2 __attribute__((ext_vector_type(2)));
typedef float tcl_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_stream(float *p, float v) {
  __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ void st_stream2(float *p, float a, float b) {
  const tcl_f2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<tcl_f2 *>(p));
}
__device__ __forceinline__ void st_stream4(float *p, float a, float b, float c, float d) {
  const tcl_f4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<tcl_f4 *>(p));
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact n / d for n < 2^31 by a host-made multiplier (2 integer instructions; a run-time divisor otherwise costs hipcc's
// ~20-instruction float-reciprocal sequence per division, per lane, even for block-uniform operands: no scalar division).
struct FastDiv {
  uint32_t m, sh, one;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return one ? n : (__umulhi(n, m) >> sh); }
};
static inline FastDiv make_fastdiv(uint32_t d) {
  if (d <= 1) return FastDiv{0, 0, 1};
  const int s = 32 - __builtin_clz(d - 1);   // ceil(log2 d)
  return FastDiv{(uint32_t)((((uint64_t)1 << (31 + s)) + d - 1) / d), (uint32_t)(s - 1), 0};
}

// Bijective XCD-aware remap of a 1-D block id: blocks are dealt round-robin over the 8 XCDs, so give each
// XCD a contiguous chunk of the logical tile order (neighbouring tiles then share one L2).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk / kNumXCD, r = nblk % kNumXCD;
  const unsigned xcd = bid % kNumXCD, idx = bid / kNumXCD;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}


// torch.linspace(-1, 1, n)[i] in fp32 (evaluated symmetrically from both ends), optionally scaled by (n-1)/n:
// the base coordinate of F.affine_grid for align_corners = True / False.
__device__ __forceinline__ float base_coord(int i, int n, bool align_corners) {
  if (n <= 1) return 0.f;
  const float step = __fdiv_rn(2.f, (float)(n - 1));
  float r = (i < n / 2) ? __fadd_rn(-1.f, __fmul_rn(step, (float)i)) : __fsub_rn(1.f, __fmul_rn(step, (float)(n - 1 - i)));
  if (!align_corners) r = __fdiv_rn(__fmul_rn(r, (float)(n - 1)), (float)n);
  return r;
}

// Source taps of F.interpolate(mode='bilinear', align_corners=False): src = scale*(dst+0.5)-0.5 clamped at 0.
struct LinTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ LinTap linear_tap(int dst, int in_size, float scale) {
  float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
  if (src < 0.f) src = 0.f;
  LinTap t;
  t.i0 = (int)src;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
  t.l1 = __fsub_rn(src, (float)t.i0);
  t.l0 = __fsub_rn(1.f, t.l1);
  return t;
}
// value = (p00*lx0 + p01*lx1)*ly0 + (p10*lx0 + p11*lx1)*ly1, each operation rounded separately
__device__ __forceinline__ float bilerp(float p00, float p01, float p10, float p11, const LinTap &tx, const LinTap &ty) {
  const float top = __fadd_rn(__fmul_rn(p00, tx.l0), __fmul_rn(p01, tx.l1));
  const float bot = __fadd_rn(__fmul_rn(p10, tx.l0), __fmul_rn(p11, tx.l1));
  return __fadd_rn(__fmul_rn(top, ty.l0), __fmul_rn(bot, ty.l1));
}

}  // namespace tcl
