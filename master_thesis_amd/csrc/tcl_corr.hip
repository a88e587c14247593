// All-pairs masked feature correlation (cost volume), forward and backward, plus the per-pixel
// correlation and the argmax -> flow reduction, for gfx950.
//
// Replaces compute_masked_4d_correlation (utils/correlation.py:28-54), compute_masked_correlation
// (utils/correlation.py:4-25) and corr_to_flow (utils/flow.py:8-26); see include/tcl.h.
//
// Structure of A1 (per batch item b and reference frame f, P = H*W pixels, C channels):
//   1. normalise:  a^[c][p] = x v / (||x v||_2 + 1e-9)   (one streaming pass; writes the GEMM operands
//                  k-major, i.e. in the NCHW order they arrive in, and the norms for the backward pass)
//   2. contraction out[p][q] = sum_c a^[c][p] r^[c][q]    LDS-tiled MFMA GEMM, K = C
//   backward:      d_a^[c][p] = sum_{f,q} r^[c][q] G[p][q],  d_r^[c][q] = sum_p a^[c][p] G[p][q]
//                  (two more MFMA GEMMs with K = P), then the normalisation's Jacobian per pixel.
//
// TCL_PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD).  The four k-slices of
// an 8-deep k group are issued as four MFMAs whose lane halves take k = k0+e and k0+4+e (e = 0..3): the
// same permutation on both operands, so a k-contiguous operand is fetched from LDS with one ds_read_b128
// per lane (rows padded to 36 floats: conflict-free) and an m-contiguous one with plain ds_read_b32.
#include "tcl_common.h"
#include <map>
#include <utility>
#include <mutex>
#include <type_traits>

namespace tcl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------ normalise
// x element (b,c,f,p) at ((b*C + c)*F + f)*P + p; v element (b,f,p); hat element ((b*F+f)*C + c)*P + p.
// block = 64 lanes x 4 channel groups; a lane owns PX (4 when P % 4 == 0, else 1) adjacent pixels and walks its
// channels NU at a time: the NU loads of a step are unconditional (channel index clamped) and in flight together.
constexpr int NU = 8;

template <int PX> struct PxV { float v[PX]; };
template <int PX> __device__ __forceinline__ PxV<PX> ldv(const float *p) {
  PxV<PX> r;
  if (PX == 4) { const f32x4 t = *reinterpret_cast<const f32x4 *>(p); r.v[0] = t[0]; r.v[1 % PX] = t[1]; r.v[2 % PX] = t[2]; r.v[3 % PX] = t[3]; }
  else if (PX == 2) { const tcl_f2 t = *reinterpret_cast<const tcl_f2 *>(p); r.v[0] = t[0]; r.v[1 % PX] = t[1]; }
  else r.v[0] = p[0];
  return r;
}
template <int PX> __device__ __forceinline__ void stv(float *p, const PxV<PX> &a) {
  if (PX == 4) { f32x4 t = {a.v[0], a.v[1 % PX], a.v[2 % PX], a.v[3 % PX]}; *reinterpret_cast<f32x4 *>(p) = t; }
  else if (PX == 2) { tcl_f2 t = {a.v[0], a.v[1 % PX]}; *reinterpret_cast<tcl_f2 *>(p) = t; }
  else p[0] = a.v[0];
}

// Where the normalised operand goes (any subset, block-uniform): the fp32 image (exact-fp32 contraction, every backward
// product), its bf16 rounding `hi` and the residual `lo` = bf16(x - hi) (TCL_PREC_BF16 / BF16X3: the contraction then reads
// ready-made bf16 operands: no conversion instructions in its k-loop, half the operand bytes).  Same element order in all three.
struct HatOut { float *f; __bf16 *hi, *lo; };
template <int PX> __device__ __forceinline__ void st_hat(const HatOut &o, int64_t idx, const PxV<PX> &a) {
  if (o.f) stv<PX>(o.f + idx, a);
  if (o.hi) {
    __bf16 h[PX], l[PX];
#pragma unroll
    for (int e = 0; e < PX; ++e) { h[e] = (__bf16)a.v[e]; l[e] = (__bf16)(a.v[e] - (float)h[e]); }
    if (PX == 4) {
      const bf16x4 hv = {h[0], h[1 % PX], h[2 % PX], h[3 % PX]}, lv = {l[0], l[1 % PX], l[2 % PX], l[3 % PX]};
      *reinterpret_cast<bf16x4 *>(o.hi + idx) = hv;
      if (o.lo) *reinterpret_cast<bf16x4 *>(o.lo + idx) = lv;
    } else if (PX == 2) {
      const bf16x2 hv = {h[0], h[1 % PX]}, lv = {l[0], l[1 % PX]};
      *reinterpret_cast<bf16x2 *>(o.hi + idx) = hv;
      if (o.lo) *reinterpret_cast<bf16x2 *>(o.lo + idx) = lv;
    } else {
      o.hi[idx] = h[0];
      if (o.lo) o.lo[idx] = l[0];
    }
  }
}

template <int PX, int NCG>
__global__ void __launch_bounds__(64 * NCG) corr_normalise_kernel(const float *__restrict__ x, const float *__restrict__ v,
                                                             HatOut hat, float *__restrict__ nrm, int C, int F, int P) {
  const int bf = blockIdx.y, b = bf / F, f = bf - b * F;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p0 = (blockIdx.x * 64 + lane) * PX;
  const bool ok = p0 < P;
  const int p = ok ? p0 : 0;
  PxV<PX> m;
#pragma unroll
  for (int e = 0; e < PX; ++e) m.v[e] = 1.f;
  if (v) m = ldv<PX>(v + (int64_t)bf * P + p);
  const float *xp = x + ((int64_t)b * C * F + f) * P + p;
  const int64_t cs = (int64_t)F * P;
  float ss[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) ss[e] = 0.f;
  for (int c0 = cg; c0 < C; c0 += NCG * NU) {
    PxV<PX> t[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) t[u] = ldv<PX>(xp + min(c0 + NCG * u, C - 1) * cs);
    TCL_LOADS_FIRST(NU);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float live = c0 + NCG * u < C ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < PX; ++e) { const float q = (t[u].v[e] * m.v[e]) * live; ss[e] += q * q; }
    }
  }
  __shared__ float red[NCG][64 * PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) red[cg][lane * PX + e] = ss[e];
  __syncthreads();
  float den[PX];
  PxV<PX> nn;
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    const int l = lane * PX + e;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};   // fixed summation order: four interleaved partial sums, then pairwise
#pragma unroll
    for (int g = 0; g < NCG; ++g) t4[g & 3] += red[g][l];
    nn.v[e] = sqrtf((t4[0] + t4[1]) + (t4[2] + t4[3]));
    den[e] = nn.v[e] + kNormEps;
  }
  if (!ok) return;
  const int64_t hp = (int64_t)bf * C * P + p;
  for (int c0 = cg; c0 < C; c0 += NCG * NU) {
    PxV<PX> t[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) t[u] = ldv<PX>(xp + min(c0 + NCG * u, C - 1) * cs);
    TCL_LOADS_FIRST(NU);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int e = 0; e < PX; ++e) t[u].v[e] = (t[u].v[e] * m.v[e]) / den[e];
      if (c0 + NCG * u < C) st_hat<PX>(hat, hp + (int64_t)(c0 + NCG * u) * P, t[u]);
    }
  }
  if (cg == 0) stv<PX>(nrm + (int64_t)bf * P + p, nn);
}

// C <= NCG * KU: the thread's KU channels stay in registers between the norm and the scaling, so x is read once
// (the two-loop kernel fetched 1.5x the algorithmic bytes at C2: its second walk missed L2).  Same operation order.
template <int PX, int NCG, int KU>
__global__ void __launch_bounds__(64 * NCG) corr_normalise_keep_kernel(const float *__restrict__ x, const float *__restrict__ v,
                                                                  HatOut hat, float *__restrict__ nrm, int C, int F, int P) {
  const int bf = blockIdx.y, b = bf / F, f = bf - b * F;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p0 = (blockIdx.x * 64 + lane) * PX;
  const bool ok = p0 < P;
  const int p = ok ? p0 : 0;
  PxV<PX> m;
#pragma unroll
  for (int e = 0; e < PX; ++e) m.v[e] = 1.f;
  if (v) m = ldv<PX>(v + (int64_t)bf * P + p);
  const float *xp = x + ((int64_t)b * C * F + f) * P + p;
  const int64_t cs = (int64_t)F * P;
  PxV<PX> t[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) t[u] = ldv<PX>(xp + min(cg + NCG * u, C - 1) * cs);
  TCL_LOADS_FIRST(KU);
  float ss[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) ss[e] = 0.f;
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const float live = cg + NCG * u < C ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < PX; ++e) {
      t[u].v[e] = t[u].v[e] * m.v[e];
      const float q = t[u].v[e] * live;
      ss[e] += q * q;
    }
  }
  __shared__ float red[NCG][64 * PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) red[cg][lane * PX + e] = ss[e];
  __syncthreads();
  float den[PX];
  PxV<PX> nn;
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    const int l = lane * PX + e;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NCG; ++g) t4[g & 3] += red[g][l];
    nn.v[e] = sqrtf((t4[0] + t4[1]) + (t4[2] + t4[3]));
    den[e] = nn.v[e] + kNormEps;
  }
  if (!ok) return;
  const int64_t hp = (int64_t)bf * C * P + p;
#pragma unroll
  for (int u = 0; u < KU; ++u) {
#pragma unroll
    for (int e = 0; e < PX; ++e) t[u].v[e] = t[u].v[e] / den[e];
    if (cg + NCG * u < C) st_hat<PX>(hat, hp + (int64_t)(cg + NCG * u) * P, t[u]);
  }
  if (cg == 0) stv<PX>(nrm + (int64_t)bf * P + p, nn);
}

template <int PX, int NCG, int KU>
__global__ void __launch_bounds__(64 * NCG) corr_normalise_bwd_keep_kernel(const float *__restrict__ x, const float *__restrict__ v,
                                                                      const float *__restrict__ dhat, const float *__restrict__ nrm,
                                                                      float *__restrict__ gx, int C, int F, int P) {
  const int bf = blockIdx.y, b = bf / F, f = bf - b * F;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p0 = (blockIdx.x * 64 + lane) * PX;
  const bool ok = p0 < P;
  const int p = ok ? p0 : 0;
  PxV<PX> m;
#pragma unroll
  for (int e = 0; e < PX; ++e) m.v[e] = 1.f;
  if (v) m = ldv<PX>(v + (int64_t)bf * P + p);
  const float *xp = x + ((int64_t)b * C * F + f) * P + p;
  const float *dp = dhat + (int64_t)bf * C * P + p;
  const int64_t cs = (int64_t)F * P;
  PxV<PX> t[KU], d[KU];
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int c = min(cg + NCG * u, C - 1);
    t[u] = ldv<PX>(xp + c * cs);
    d[u] = ldv<PX>(dp + (int64_t)c * P);
  }
  const PxV<PX> n = ldv<PX>(nrm + (int64_t)bf * P + p);
  TCL_LOADS_FIRST(2 * KU + 1);
  float dot[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) dot[e] = 0.f;
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const float live = cg + NCG * u < C ? 1.f : 0.f;
#pragma unroll
    for (int e = 0; e < PX; ++e) {
      t[u].v[e] = t[u].v[e] * m.v[e];
      dot[e] += (d[u].v[e] * t[u].v[e]) * live;
    }
  }
  __shared__ float red[NCG][64 * PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) red[cg][lane * PX + e] = dot[e];
  __syncthreads();
  float ne[PX], k[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    const int l = lane * PX + e;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NCG; ++g) t4[g & 3] += red[g][l];
    const float dt = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    ne[e] = n.v[e] + kNormEps;
    k[e] = n.v[e] > 0.f ? dt / (ne[e] * ne[e] * n.v[e]) : 0.f;
  }
  if (!ok) return;
  float *gp = gx + ((int64_t)b * C * F + f) * P + p;
#pragma unroll
  for (int u = 0; u < KU; ++u) {
#pragma unroll
    for (int e = 0; e < PX; ++e) d[u].v[e] = (d[u].v[e] / ne[e] - t[u].v[e] * k[e]) * m.v[e];
    if (cg + NCG * u < C) stv<PX>(gp + (cg + NCG * u) * cs, d[u]);
  }
}

// Jacobian of the normalisation: g_x = v * (d_hat/(n+eps) - t * <d_hat,t> / ((n+eps)^2 n)),  0-norm -> first term only.
template <int PX, int NCG>
__global__ void __launch_bounds__(64 * NCG) corr_normalise_bwd_kernel(const float *__restrict__ x, const float *__restrict__ v,
                                                                 const float *__restrict__ dhat, const float *__restrict__ nrm,
                                                                 float *__restrict__ gx, int C, int F, int P) {
  const int bf = blockIdx.y, b = bf / F, f = bf - b * F;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int p0 = (blockIdx.x * 64 + lane) * PX;
  const bool ok = p0 < P;
  const int p = ok ? p0 : 0;
  PxV<PX> m;
#pragma unroll
  for (int e = 0; e < PX; ++e) m.v[e] = 1.f;
  if (v) m = ldv<PX>(v + (int64_t)bf * P + p);
  const float *xp = x + ((int64_t)b * C * F + f) * P + p;
  const float *dp = dhat + (int64_t)bf * C * P + p;
  const int64_t cs = (int64_t)F * P;
  float dot[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) dot[e] = 0.f;
  constexpr int NH = NU / 2;
  for (int c0 = cg; c0 < C; c0 += NCG * NH) {
    PxV<PX> t[NH], d[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int c = min(c0 + NCG * u, C - 1);
      t[u] = ldv<PX>(xp + c * cs);
      d[u] = ldv<PX>(dp + (int64_t)c * P);
    }
    TCL_LOADS_FIRST(NU);
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const float live = c0 + NCG * u < C ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < PX; ++e) dot[e] += (d[u].v[e] * (t[u].v[e] * m.v[e])) * live;
    }
  }
  __shared__ float red[NCG][64 * PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) red[cg][lane * PX + e] = dot[e];
  __syncthreads();
  const PxV<PX> n = ldv<PX>(nrm + (int64_t)bf * P + p);
  float ne[PX], k[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    const int l = lane * PX + e;
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NCG; ++g) t4[g & 3] += red[g][l];
    const float dt = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    ne[e] = n.v[e] + kNormEps;
    k[e] = n.v[e] > 0.f ? dt / (ne[e] * ne[e] * n.v[e]) : 0.f;
  }
  if (!ok) return;
  float *gp = gx + ((int64_t)b * C * F + f) * P + p;
  for (int c0 = cg; c0 < C; c0 += NCG * NH) {
    PxV<PX> t[NH], d[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
      const int c = min(c0 + NCG * u, C - 1);
      t[u] = ldv<PX>(xp + c * cs);
      d[u] = ldv<PX>(dp + (int64_t)c * P);
    }
    TCL_LOADS_FIRST(NU);
#pragma unroll
    for (int u = 0; u < NH; ++u) {
#pragma unroll
      for (int e = 0; e < PX; ++e) d[u].v[e] = (d[u].v[e] / ne[e] - (t[u].v[e] * m.v[e]) * k[e]) * m.v[e];
      if (c0 + NCG * u < C) stv<PX>(gp + (c0 + NCG * u) * cs, d[u]);
    }
  }
}

// ------------------------------------------------------------------------------------------ fp32 MFMA GEMM
struct GemmArgs {
  const float *A, *B;
  float *D;
  int M, N, K;            // K = depth of ONE segment
  int nseg;               // segments accumulated into the same D (the reference frames, for d_a^)
  int64_t a_ld, b_ld;     // stride between rows of the contiguous axis
  int64_t a_bs, b_bs, d_bs;  // batch strides
  int64_t a_ss, b_ss;     // segment strides
  int64_t d_ld;
  int a_bdiv;             // A batch index = z / a_bdiv
  int nbatch;
  int tiles_m, tiles_n;
  int d_vec;              // D rows are 16-byte aligned: the store epilogue writes float4
};

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;  // padded row of a k-contiguous LDS tile (floats): ds_read_b128 conflict-free

// Stages one operand tile (ROWS x BK, rows = m or n) from global memory into registers.
//  KC:  element (row,k) at base + row*ld + k   -> LDS [ROWS][KC_LD]
//  !KC: element (row,k) at base + k*ld + row   -> LDS [BK][ROWS]
template <int ROWS, int NT, bool KC, bool VEC>
struct TileLoader {
  static constexpr int NV = ROWS * BK / 4 / NT;  // float4 per thread
  static_assert(ROWS * BK / 4 % NT == 0, "tile must divide over the block");
  f32x4 r[NV];

  __device__ __forceinline__ void load(const float *base, int64_t ld, int row0, int k0, int rows, int depth, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      int row, k;
      if (KC) { row = idx / (BK / 4); k = (idx % (BK / 4)) * 4; }
      else    { k = idx / (ROWS / 4); row = (idx % (ROWS / 4)) * 4; }
      const int gr = row0 + row, gk = k0 + k;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (KC) {
        if (gr < rows) {
          const float *p = base + (int64_t)gr * ld + gk;
          if (VEC) { if (gk < depth) val = *reinterpret_cast<const f32x4 *>(p); }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gk + e < depth) val[e] = p[e];
          }
        }
      } else {
        if (gk < depth) {
          const float *p = base + (int64_t)gk * ld + gr;
          if (VEC) { if (gr < rows) val = *reinterpret_cast<const f32x4 *>(p); }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gr + e < rows) val[e] = p[e];
          }
        }
      }
      r[i] = val;
    }
  }

  // Full tiles (M, N multiples of the block tile, K of BK, 16-byte aligned rows): no bounds, no per-load 64-bit index
  // arithmetic.  float4 i of a thread sits a block-uniform `step(ld)` floats behind float4 i - 1, the next k-tile `kstep(ld)`
  // floats further.
  __device__ __forceinline__ static int64_t step(int64_t ld) { return KC ? (NT / (BK / 4)) * ld : (NT / (ROWS / 4)) * ld; }
  __device__ __forceinline__ static int64_t kstep(int64_t ld) { return KC ? BK : BK * ld; }
  // The same loads as `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`: a descriptor on the operand's (row0, k-tile 0)
  // element, loop-invariant per-thread 32-bit byte offsets, and the k-tile as a scalar byte offset: no vector address
  // arithmetic per load (hipcc keeps `uniform pointer + zext(offset)` as a 64-bit vector add per load otherwise).
  __device__ __forceinline__ static void thread_offsets(uint32_t (&off)[NV], int64_t ld, int tid) {
    // 32-bit arithmetic, one multiply: float4 i lies a block-uniform step(ld) floats behind float4 i - 1
    const uint32_t ld4 = (uint32_t)ld * 4u, st4 = (uint32_t)step(ld) * 4u;
    off[0] = KC ? (uint32_t)(tid / (BK / 4)) * ld4 + (uint32_t)(tid % (BK / 4)) * 16u : (uint32_t)(tid / (ROWS / 4)) * ld4 + (uint32_t)(tid % (ROWS / 4)) * 16u;
#pragma unroll
    for (int i = 1; i < NV; ++i) off[i] = off[i - 1] + st4;
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rsrc, const uint32_t (&off)[NV], int soff) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[i], soff, 0);
      r[i] = __builtin_bit_cast(f32x4, v);
    }
  }

  // PW > 0 (k-major tiles only): rows are stored permuted so that the 32-row MFMA blocks of ONE wave lie PW*32 floats
  // apart: row = w*(ROWS/PW) + i*32 + l  ->  i*(PW*32) + w*32 + l.  With PW = 2 that stride is 64 dwords and a k row is 128,
  // so every fragment pair of a k-tile is one ds_read2st64_b32 off a single base register.
  template <int PW = 0>
  __device__ __forceinline__ void store(float *lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      if (KC) {
        const int row = idx / (BK / 4), k = (idx % (BK / 4)) * 4;
        *reinterpret_cast<f32x4 *>(lds + row * KC_LD + k) = r[i];
      } else {
        const int k = idx / (ROWS / 4);
        int row = (idx % (ROWS / 4)) * 4;
        if (PW > 0) row = ((row % (ROWS / PW)) / 32) * (PW * 32) + (row / (ROWS / PW)) * 32 + (row & 31);
        *reinterpret_cast<f32x4 *>(lds + k * ROWS + row) = r[i];
      }
    }
  }
};

template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 read_frag(const float *lds, int row, int kbase) {
  // -> the four k-slices (kbase + e, e = 0..3) of one row, for this lane
  if (KC) return *reinterpret_cast<const f32x4 *>(lds + row * KC_LD + kbase);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = lds[(kbase + e) * ROWS + row];
  return v;
}

// One BKT-deep k-tile of a wave's TM x TN MFMA blocks, both operands k-major in LDS ([k][rows], rows permuted as
// TileLoader::store<PW> describes).  A fragment slice (one k per lane half) is TM + TN scalar LDS reads.  Software-pipelined
// by hand: the reads of slice s+1 are issued BEFORE the TM*TN MFMAs of slice s, so the LDS latency hides under 256
// cycles of MFMA.  (hipcc on its own sinks every read group next to its use: `ds_read; s_waitcnt lgkmcnt(0);
// 4 x mfma`, which leaves the matrix pipe idle for the LDS round trip 16 times per k-tile.)  The
// sched_group_barriers pin that order: the DS reads of the next slice, then the MFMAs of this one.
constexpr int kEpiAux = 2;     // cache policy of the epilogues' buffer stores: 2 = non-temporal (plain stores measured in DESIGN.md section 4)
constexpr int kGemmPrio = 2;   // wave priority during the MFMA phase of gemm_f32_kernel
// FIRST: the accumulators are undefined on entry; the first slice's MFMAs take an inline-constant zero as C (saves the 64
// v_accvgpr_write of a zero fill: VALU instructions, which take issue slots from the MFMAs of the SIMD's other waves).
template <int BM, int BN, int WM, int WN, int BKT, bool FIRST = false>
__device__ __forceinline__ void mma_ktile_kmajor(const float *As, const float *Bs, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], int wid, int lr, int lh) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NSL = BKT / 2;   // slices per k-tile
    float a_cur[TM], b_cur[TN], a_nxt[TM], b_nxt[TN];
    const float *ap = As + (lh * 4) * BM + (wid / WN) * 32 + lr, *bp = Bs + (lh * 4) * BN + (wid % WN) * 32 + lr;
#pragma unroll
    for (int i = 0; i < TM; ++i) a_cur[i] = ap[i * (WM * 32)];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_cur[j] = bp[j * (WN * 32)];
    __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) / 2, 0);       // DS reads of slice 0
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
      if (sl + 1 < NSL) {
        const int kk = ((sl + 1) >> 2) * 8 + ((sl + 1) & 3);   // k row of the next slice (this lane half adds lh * 4)
#pragma unroll
        for (int i = 0; i < TM; ++i) a_nxt[i] = ap[kk * BM + i * (WM * 32)];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_nxt[j] = bp[kk * BN + j * (WN * 32)];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[j], a_cur[i], FIRST && sl == 0 ? zero : acc[i][j], 0, 0, 0);
        }
      if (sl + 1 < NSL) __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN) / 2, 0);   // DS reads of slice sl+1
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                           // MFMAs of slice sl
      if (sl + 1 < NSL) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_cur[j] = b_nxt[j];
      }
    }
}

enum { EPI_STORE = 0, EPI_ARGMAX = 1 };

// Epilogue of the fp32 GEMM kernels (the k-loop has ended on a barrier: `smem`, SMEM floats, is free again).
template <int BM, int BN, int WM, int WN, int EPI, bool FULL, int SMEM>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float *smem, const GemmArgs &g, unsigned z, int m0, int n0,
                                              unsigned tn) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
  const int lr = lane & 31, lh = lane >> 5;
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The tiles were accumulated
  // transposed (B fragment as the MFMA's A operand): a lane holds ONE row m of D and, in register quad q, the four
  // consecutive columns n = 8 q + 4 (lane >> 5) + e.
  if (EPI == EPI_ARGMAX) {
    // A row's maximum is therefore an in-register scan over the lane's TN * 16 columns (ascending n, strict '>' keeps the
    // first maximum), one exchange with the other lane half and one LDS step across the WN waves of the row.
    // smem is free again (the loop ended on a barrier): [BM][WN] (value, column) pairs
    float2 *best = reinterpret_cast<float2 *>(smem);
    static_assert(BM * WN * 2 <= SMEM, "argmax scratch must fit the staging buffer");
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
      if (FULL) {
        // whole tiles: three vector instructions per element (compare, two selects; the running index is the element's position
        // in the lane's ascending column order, an inline constant) - this scan is most of a bf16 tile's epilogue, and vector
        // instructions take matrix-pipe issue slots from the other waves of the SIMD
        int li = 0;
        bv = acc[i][0][0];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (j == 0 && r == 0) continue;
            const float v = acc[i][j][r];
            const bool gt = v > bv;
            bv = gt ? v : bv;
            li = gt ? j * 16 + r : li;
          }
        bi = n0 + wn0 + (li >> 4) * 32 + ((li >> 2) & 3) * 8 + 4 * lh + (li & 3);
      } else
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn0 + j * 32 + 8 * q + 4 * lh + e;
            const float v = acc[i][j][4 * q + e];
            if (n < g.N && (v > bv || bi == 0x7fffffff)) { bv = v; bi = n; }
          }
      const float ov = __shfl_xor(bv, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      if (lh == 0) best[(wm0 + i * 32 + lr) * WN + (wid % WN)] = make_float2(bv, __int_as_float(bi));
    }
    __syncthreads();
    for (int ml = tid; ml < BM; ml += NT) {
      const int m = m0 + ml;
      if (m >= g.M) continue;
      float2 b = best[ml * WN];
#pragma unroll
      for (int k = 1; k < WN; ++k) {
        const float2 o = best[ml * WN + k];
        if (o.x > b.x || (o.x == b.x && __float_as_int(o.y) < __float_as_int(b.y))) b = o;
      }
      reinterpret_cast<float2 *>(g.D)[((int64_t)z * g.M + m) * g.tiles_n + tn] = b;
    }
    return;
  }
  // The tiles were accumulated transposed (B fragment as the MFMA's A operand): a lane holds ONE row m of D and, in
  // each register quad, four consecutive columns n -> one 16-byte store per quad instead of four 4-byte stores.
  float *Db = g.D + (int64_t)z * g.d_bs;
  constexpr int ELD = WN * 32 + 4;   // staging row (floats), padded
  if (FULL && EPI == EPI_STORE && WM * 32 * ELD <= SMEM && g.d_vec) {
    // Full tiles: the accumulators go through LDS once so that every store instruction of a wave writes eight complete,
    // line-aligned 128-byte row segments (the direct path writes 32-byte pieces of 32 different rows per instruction), and
    // those leave as non-temporal stores (whole lines: nothing to merge in L2, and the operand panels stay there).
    // One pass per (i, j): the (i, j) 32 x 32 blocks of all WM x WN waves = WM*32 rows x (WN x 32) columns, rows padded by 4
    // floats.  C2 forward: 1.205 -> 1.188 ms back to back (non-temporal stores of the direct path's 16-byte pieces had
    // doubled the time of the bf16 variant).
    // The stores are buffer stores: a descriptor on the tile's first element, four loop-invariant per-thread byte offsets and
    // the pass (i, j) as a scalar offset, i.e. no vector address arithmetic (VALU instructions take issue slots from the MFMAs
    // of the other waves on the SIMD; the launcher checks that a tile's byte offsets fit 31 bits).
    const int wm = wid / WN, wn = wid % WN;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(Db + (int64_t)m0 * g.d_ld + n0, 0, 0x7fffffff, 0x00020000);
    const int d_ld4 = g.d_ld * 4;
    int st_off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // WM*32 x WN*8 float4 over NT = WM*WN*64 threads
      const int idx = tid + NT * k, row = idx / (WN * 8), c4 = idx % (WN * 8), seg = c4 >> 3, within = (c4 & 7) * 4;
      st_off[k] = ((row >> 5) * (BM / WM) + (row & 31)) * d_ld4 + (seg * (BN / WN) + within) * 4;
    }
    const int rd_off = (tid / (WN * 8)) * ELD + ((tid % (WN * 8)) >> 3) * 32 + (tid % 8) * 4;   // this thread's float4 of readback k = 0
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (i + j > 0) __syncthreads();   // the previous pass has been read back (the k-loop itself ended on a barrier)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 val = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4 *>(smem + (wm * 32 + lr) * ELD + wn * 32 + 8 * q + 4 * lh) = val;
        }
        __syncthreads();
        const int pass_off = i * 32 * d_ld4 + j * 32 * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 val = *reinterpret_cast<const f32x4 *>(smem + rd_off + k * (NT / (WN * 8)) * ELD);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rd, st_off[k], pass_off, kEpiAux);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = m0 + wm0 + i * 32 + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn0 + j * 32 + 8 * q + 4 * lh;
        float *dst = Db + (int64_t)m * g.d_ld + n;
        if (m < g.M) {
          if (g.d_vec && n + 3 < g.N) {
            const f32x4 val = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4 *>(dst) = val;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < g.N) dst[e] = acc[i][j][4 * q + e];
          }
        }
      }
    }
}

// D[m][n] (+)= sum_k A(m,k) B(n,k); block tile BM x BN, WM x WN waves, each wave (BM/WM) x (BN/WN).
// EPI_STORE writes D; EPI_ARGMAX instead reduces every row of the tile to (max value, first column reaching it) and
// writes that pair to g.D as float2 at [z][m][n-tile] (the volume itself never reaches HBM).

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, int EPI, bool FULL = false>
__global__ void __launch_bounds__(WM *WN * 64, (EPI == EPI_STORE && FULL ? ((BM / WM) * (BN / WN) > 64 * 64 ? 2 : 4) : 1)) gemm_f32_kernel(GemmArgs g) {   // full-tile store kernels: 4 waves per SIMD (<= 128 registers)
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_LDS = A_KC ? BM * KC_LD : BK * BM;
  constexpr int B_LDS = B_KC ? BN * KC_LD : BK * BN;
  __shared__ __attribute__((aligned(16))) float smem[A_LDS + B_LDS];
  float *As = smem, *Bs = smem + A_LDS;

  // block -> (batch, m-tile, n-tile): XCD-contiguous chunks, grouped so that neighbours share operand panels
  const unsigned per_batch = g.tiles_m * g.tiles_n;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lin / per_batch, local = lin - z * per_batch;
  constexpr unsigned GROUP = 8;
  const unsigned gw = GROUP * g.tiles_n;
  const unsigned gid = local / gw, first_m = gid * GROUP;
  const unsigned gsz = min((unsigned)g.tiles_m - first_m, GROUP);
  const unsigned tm = first_m + (local % gw) % gsz, tn = (local % gw) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
  const int lr = lane & 31, lh = lane >> 5;

  const float *Ab = g.A + (int64_t)(z / g.a_bdiv) * g.a_bs;
  const float *Bb = g.B + (int64_t)z * g.b_bs;

  f32x16 acc[TM][TN];   // defined by the first k-tile (see mma_ktile_kmajor<FIRST>)

  TileLoader<BM, NT, A_KC, VEC> la;
  TileLoader<BN, NT, B_KC, VEC> lb;
  const int ktiles = (g.K + BK - 1) / BK;
  const int total = ktiles * g.nseg;

  // FULL: operand cursors = a buffer descriptor on the block's (row0, k-tile 0) element + loop-invariant per-thread byte
  // offsets + a scalar byte offset that moves from k-tile to k-tile (segment change: rewind + a_ss / b_ss): no vector address
  // arithmetic per load.  (The launcher checks that these offsets fit 31 bits.)
  constexpr bool BUF = FULL;
  const int64_t a_k = la.kstep(g.a_ld), b_k = lb.kstep(g.b_ld);
  uint32_t a_off[la.NV], b_off[lb.NV];
  __amdgpu_buffer_rsrc_t ra, rb;
  int sa = 0, sb = 0;
  if (BUF) {
    la.thread_offsets(a_off, g.a_ld, tid);
    lb.thread_offsets(b_off, g.b_ld, tid);
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Ab + (A_KC ? (int64_t)m0 * g.a_ld : m0)), 0, 0x7fffffff, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bb + (B_KC ? (int64_t)n0 * g.b_ld : n0)), 0, 0x7fffffff, 0x00020000);
  }
  int kt_next = 1;
  if (BUF) { la.load_buf(ra, a_off, 0); lb.load_buf(rb, b_off, 0); }
  else {
    la.load(Ab, g.a_ld, m0, 0, g.M, g.K, tid);
    lb.load(Bb, g.b_ld, n0, 0, g.N, g.K, tid);
  }
  constexpr bool PIPE = !A_KC && !B_KC;   // hand-pipelined fragment reads, permuted LDS rows
  auto ktile = [&](auto firstc, int t) {
    constexpr bool FIRST = decltype(firstc)::value;
    la.template store<PIPE ? WM : 0>(As, tid);
    lb.template store<PIPE ? WN : 0>(Bs, tid);
    __syncthreads();
    if (t + 1 < total) {  // prefetch the next tile while this one is multiplied
      if (BUF) {
        if (kt_next == ktiles) { kt_next = 0; sa += (int)(g.a_ss - (ktiles - 1) * a_k) * 4; sb += (int)(g.b_ss - (ktiles - 1) * b_k) * 4; }
        else { sa += (int)a_k * 4; sb += (int)b_k * 4; }
        ++kt_next;
        la.load_buf(ra, a_off, sa);
        lb.load_buf(rb, b_off, sb);
      } else {
        const int seg = (t + 1) / ktiles, kt = (t + 1) - seg * ktiles;
        la.load(Ab + seg * g.a_ss, g.a_ld, m0, kt * BK, g.M, g.K, tid);
        lb.load(Bb + seg * g.b_ss, g.b_ld, n0, kt * BK, g.N, g.K, tid);
      }
    }
    // Raised wave priority while a wave multiplies (back to 0 for its staging): 1.3 % more cycles, but on the boxes that
    // throttle this kernel (2.22 GHz instead of 2.40) the clock only drops to 2.31-2.34 GHz: 1.06 -> 1.005 ms
    // there, 0.943 -> 0.955 ms on an unthrottled box.
    __builtin_amdgcn_s_setprio(kGemmPrio);
    if (PIPE) {
      mma_ktile_kmajor<BM, BN, WM, WN, BK, FIRST>(As, Bs, acc, wid, lr, lh);
    } else
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_KC>(As, wm0 + i * 32 + lr, kg * 8 + lh * 4);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = read_frag<BN, B_KC>(Bs, wn0 + j * 32 + lr, kg * 8 + lh * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
          {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][e], af[i][e], FIRST && kg == 0 && e == 0 ? zero : acc[i][j], 0, 0, 0);   // transposed tile, see the epilogues
          }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  };
  ktile(std::true_type{}, 0);   // peeled: its MFMAs start from an inline zero
  for (int t = 1; t < total; ++t) ktile(std::false_type{}, t);

  gemm_epilogue<BM, BN, WM, WN, EPI, FULL, A_LDS + B_LDS>(acc, smem, g, z, m0, n0, tn);
}

// ---- wide contraction: 256 x 256 block tile, 4 waves, each wave 128 x 128 = 4 x 4 MFMA blocks (256 accumulators, in
// AGPRs), ONE wave per SIMD and one block per CU.  Against the 64 x 64 wave tiles above this halves the LDS fragment traffic
// per MFMA; with nothing else resident on the SIMD, every latency is hidden inside the wave instead:
//   * two LDS stages, ONE barrier per 32-deep k-tile, placed before the tile's last 8-deep group: by then every wave has its
//     last fragments of this tile in registers and has written its share of the next tile, so after the barrier the last
//     group's MFMAs run while the first fragments of the NEXT tile are read;
//   * fragments of group g+1 are read during the 64 MFMAs of group g (two register sets);
//   * the next tile's global loads (issued a whole k-tile earlier) are written to the other LDS stage during groups 0 and 1,
//     and the loads of the tile after that are issued right behind them (one staging register set).
// Used for the two backward products (K = P is long).  A persistent variant for the forward contraction (one block per CU walking
// over output tiles) was measured in round 2 and not adopted; it is no longer part of this file.
// Full tiles only (M, N % 256 == 0, K % 64 == 0, 16-byte aligned rows).
template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) gemm_f32_wide_kernel(GemmArgs g) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 2, NT = 256, TM = 4, TN = 4;
  constexpr int A_LDS = A_KC ? BM * KC_LD : BK * BM, B_LDS = B_KC ? BN * KC_LD : BK * BN, STAGE = A_LDS + B_LDS;
  extern __shared__ __attribute__((aligned(16))) float smem_wide[];
  float *smem = smem_wide;

  const unsigned per_batch = g.tiles_m * g.tiles_n;
  const unsigned first = xcd_remap(blockIdx.x, gridDim.x);   // this block's output tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];   // defined by the first k-tile (mma_group<FIRST>: C = inline zero)

  TileLoader<BM, NT, A_KC, true> la;
  TileLoader<BN, NT, B_KC, true> lb;
  const int ktiles = g.K / BK, per_tile = ktiles * g.nseg, total = per_tile;
  // operand cursors: block-uniform origins of the k-tile to load next (scalar registers) + loop-invariant per-thread offsets
  uint32_t a_off[la.NV], b_off[lb.NV];
  la.thread_offsets(a_off, g.a_ld, tid);
  lb.thread_offsets(b_off, g.b_ld, tid);
  const int a_k = (int)(la.kstep(g.a_ld) * 4), b_k = (int)(lb.kstep(g.b_ld) * 4);   // bytes from one k-tile to the next
  const int a_seg = (int)(g.a_ss * 4) - (ktiles - 1) * a_k, b_seg = (int)(g.b_ss * 4) - (ktiles - 1) * b_k;   // ... into the next segment
  __amdgpu_buffer_rsrc_t ra, rb;
  int sa = 0, sb = 0;   // scalar byte offsets of the k-tile to load next
  auto tile_cursors = [&](unsigned lin) {
    const unsigned z = lin / per_batch, local = lin - z * per_batch;
    const int64_t m0 = (int64_t)(local % g.tiles_m) * BM, n0 = (int64_t)(local / g.tiles_m) * BN;
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.A + (int64_t)(z / g.a_bdiv) * g.a_bs + (A_KC ? m0 * g.a_ld : m0)), 0, 0x7fffffff, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.B + (int64_t)z * g.b_bs + (B_KC ? n0 * g.b_ld : n0)), 0, 0x7fffffff, 0x00020000);
    sa = 0; sb = 0;
  };
  tile_cursors(first);
  int kt = 0, issued = 0;   // the k-tile the cursors point at: within its segment / overall
  // moves the cursors to the next k-tile; past the last one they stay (the extra loads re-read it, nobody uses them)
  auto advance = [&]() {
    // branch-free: a branch on the common path would split the MFMA stream of the k-loop into basic blocks
    const bool more = issued + 1 < total, wrap = kt + 1 == ktiles;
    issued += more ? 1 : 0;
    kt = more ? (wrap ? 0 : kt + 1) : kt;
    sa += more ? (wrap ? a_seg : a_k) : 0;
    sb += more ? (wrap ? b_seg : b_k) : 0;
  };

  // k-major operands are read INTERLEAVED: MFMA block i (j) of this wave owns the rows (columns) base + 4 * l + i, l = the
  // MFMA's 32 row / column index, so ONE ds_read_b128 at a k brings that k's value for all four blocks (instead of four scalar
  // reads per block); the epilogues below undo the interleave.  k-contiguous operands keep block i = rows base + 32 i + l.
  constexpr bool AQ = !A_KC, BQ = !B_KC;
  f32x4 af[2][TM], bf[2][TN];
  auto read_group = [&](auto setc, const float *st, int kg) {
    constexpr int set = decltype(setc)::value;
    if (AQ) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(st + (kg * 8 + lh * 4 + e) * BM + wm0 + lr * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[set][i][e] = v[i];
      }
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) af[set][i] = read_frag<BM, A_KC>(st, wm0 + i * 32 + lr, kg * 8 + lh * 4);
    if (BQ) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(st + A_LDS + (kg * 8 + lh * 4 + e) * BN + wn0 + lr * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[set][j][e] = v[j];
      }
    } else
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[set][j] = read_frag<BN, B_KC>(st + A_LDS, wn0 + j * 32 + lr, kg * 8 + lh * 4);
  };
  auto mma_group = [&](auto setc, auto firstc) {
    constexpr int set = decltype(setc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j][e], af[set][i][e], FIRST && e == 0 ? zero : acc[i][j], 0, 0, 0);   // transposed tile
        }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // instruction order of one group = 64 MFMAs + NR fragment reads (+ 8 LDS writes + 8 global loads in groups 0 and 1)
  constexpr int NR = 8;   // 4 ds_read_b128 per operand
  auto deal_staged = [&]() {   // LDS reads cannot move across the LDS writes (may alias, as far as hipcc knows): reads first
#pragma unroll
    for (int i = 0; i < NR; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, NR > 8 ? 1 : 2, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 64 - 8 * (NR > 8 ? 3 : 4) - 2 * NR, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto deal_reads = [&]() {
#pragma unroll
    for (int i = 0; i < NR; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, NR > 12 ? 3 : 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 64 - (NR > 12 ? 3 : 4) * NR, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  la.load_buf(ra, a_off, sa); lb.load_buf(rb, b_off, sb);
  advance();
  la.template store<0>(smem, tid); lb.template store<0>(smem + A_LDS, tid);
  la.load_buf(ra, a_off, sa); lb.load_buf(rb, b_off, sb);   // k-tile 1 (or 0 again)
  advance();
  __syncthreads();
  read_group(S0{}, smem, 0);
  __builtin_amdgcn_sched_barrier(0);
  // one k-tile out of LDS stage PAR (compile-time: every LDS address of the loop is then a loop-invariant base + an immediate)
  using NO = std::false_type;
  auto ktile = [&](auto par, auto firstc) {
    constexpr int PAR = decltype(par)::value;
    const float *cur = smem + PAR * STAGE;
    float *nxt = smem + (1 - PAR) * STAGE;
    // group 0: fragments of group 1; this thread's share of the next k-tile's A image; loads of the k-tile after it.
    // hipcc on its own lumps the fragment reads and the staging instructions between two MFMA runs (8 ds_read_b128 in a row
    // take longer to issue than one MFMA runs).  The sched_group_barriers deal them out one at a time between pairs of MFMAs;
    // the sched_barriers keep every group's instructions inside the group.
    read_group(S1{}, cur, 1);
    la.template store<0>(nxt, tid);
    la.load_buf(ra, a_off, sa);
    mma_group(S0{}, firstc);
    deal_staged();
    // group 1: the B image likewise
    read_group(S0{}, cur, 2);
    lb.template store<0>(nxt + A_LDS, tid);
    lb.load_buf(rb, b_off, sb);
    mma_group(S1{}, NO{});
    deal_staged();
    advance();
    // group 2: the k-tile's last fragments
    read_group(S1{}, cur, 3);
    mma_group(S0{}, NO{});
    deal_reads();
    __syncthreads();   // every wave: done reading `cur`, done writing `nxt`
    read_group(S0{}, nxt, 0);
    mma_group(S1{}, NO{});
    deal_reads();
  };
  // The accumulators' way out when an operand is interleaved: one pass per i (the (i, *) blocks of all four waves = 64 rows x 256
  // columns) through a free LDS stage (exactly 64 KB: no room for padding, so the float4 columns are XOR-swizzled with the
  // row: 2-way conflicts at worst), leaving as whole 1 KB rows: buffer stores, loop-invariant per-thread offsets, the pass as a
  // scalar offset.  Register r of the four blocks (i, 0..3) = four consecutive columns wn0 + 4 n_l + j, n_l = (r & 3) +
  // 8 (r >> 2) + 4 lh (BQ); with AQ block i's row l is row wm0 + 4 l + i of the tile, else wm0 + 32 i + l.
  auto store_tile_interleaved = [&](float *stg, unsigned z, int m0, int n0) {
    static_assert(WM * 32 * BN <= STAGE, "epilogue staging must fit one operand stage");
    const int wm = wid / WN, wn = wid % WN;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(g.D + (int64_t)z * g.d_bs + (int64_t)m0 * g.d_ld + n0, 0, 0x7fffffff, 0x00020000);
    const int d_ld4 = g.d_ld * 4;
    constexpr int RPT = WM * 32 * (BN / 4) / NT;       // float4 per thread and pass (16): staged rows rrow + 4 k
    const int rrow = tid / (BN / 4), rc4 = tid % (BN / 4);
    const int row_bytes = (AQ ? 4 : 1) * d_ld4;       // distance of two consecutive staged rows of one wave row in D
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i > 0) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f32x4 val = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        const int nl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<f32x4 *>(stg + (wm * 32 + lr) * BN + ((wn * (BN / WN) + nl * 4) ^ (lr << 2))) = val;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        // staged row rrow + 4 k: wave row (k >> 3), l = 4 (k & 7) + rrow
        const int l = 4 * (k & 7) + rrow;
        const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + ((k >> 3) * 32 + l) * BN + ((rc4 * 4) ^ (l << 2)));
        const int voff = rrow * row_bytes + rc4 * 16;
        const int soff = (k >> 3) * (BM / WM) * d_ld4 + 4 * (k & 7) * row_bytes + (AQ ? i : i * 32) * d_ld4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rd, voff, soff, kEpiAux);
      }
    }
  };
  // the k-tiles of the output tile (an even number: the launcher checks), two per trip, the first from a zero C
  ktile(S0{}, std::true_type{});
  ktile(S1{}, NO{});
  for (int kk = 2; kk < per_tile; kk += 2) {
    ktile(S0{}, NO{});
    ktile(S1{}, NO{});
  }
  {
    const unsigned z = first / per_batch, local = first - z * per_batch;
    const int m0 = (int)(local % g.tiles_m) * BM, n0 = (int)(local / g.tiles_m) * BN;
    __syncthreads();
    if (AQ || BQ) store_tile_interleaved(smem, z, m0, n0);
    else gemm_epilogue<BM, BN, WM, WN, EPI_STORE, true, 2 * STAGE>(acc, smem, g, z, m0, n0, local / g.tiles_m);
  }
}

// Once per device and instantiation: raise the kernel's dynamic-LDS limit to what it needs (139-147 KB of the CU's 160 KB) and
// read the CU count.  Returns the CU count, or 0 if the device refuses (the caller then takes the 64 x 64-wave-tile kernels).
template <bool A_KC, bool B_KC>
static int wide_kernel_ready() {
  constexpr size_t lds = 2 * ((A_KC ? 256 * KC_LD : BK * 256) + (B_KC ? 256 * KC_LD : BK * 256)) * sizeof(float);
  static std::mutex mu;
  static int cus[64];   // 0 = not asked yet, -1 = refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  std::lock_guard<std::mutex> lock(mu);
  if (cus[dev] == 0) {
    int n = 0;
    const bool ok = hipFuncSetAttribute((const void *)gemm_f32_wide_kernel<A_KC, B_KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0;
    if (!ok) (void)hipGetLastError();   // a refusal is not an error of the call that asked
    cus[dev] = ok ? n : -1;
  }
  return cus[dev] > 0 ? cus[dev] : 0;
}

template <bool A_KC, bool B_KC>
static int launch_gemm_wide(GemmArgs g, hipStream_t s, const char *what) {
  g.tiles_m = g.M / 256;
  g.tiles_n = g.N / 256;
  g.d_vec = aligned(g.D, 16) && g.d_ld % 4 == 0 && g.d_bs % 4 == 0;
  const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.nbatch;
  constexpr size_t lds = 2 * ((A_KC ? 256 * KC_LD : BK * 256) + (B_KC ? 256 * KC_LD : BK * 256)) * sizeof(float);
  hipLaunchKernelGGL((gemm_f32_wide_kernel<A_KC, B_KC>), dim3((unsigned)blocks), dim3(256), lds, s, g);
  return check_launch(what);
}
// the wide kernel wants whole 256 x 256 x 64 tiles, 16-byte aligned rows and enough tiles to give most CUs one
static bool gemm_wide_ok(const GemmArgs &g, bool a_kc, bool b_kc) {
  const bool vec = aligned(g.A, 16) && aligned(g.B, 16) && aligned(g.D, 16) && g.a_ld % 4 == 0 && g.b_ld % 4 == 0 && g.d_ld % 4 == 0 &&
                   g.a_bs % 4 == 0 && g.b_bs % 4 == 0 && g.d_bs % 4 == 0 && g.a_ss % 4 == 0 && g.b_ss % 4 == 0;
  // K % 64: an even number of k-tiles per output tile (the k-loop is unrolled over the two LDS stages); the byte offsets of
  // one output tile's operand panels (rows x ld for a k-contiguous operand, K x ld for a k-major one, all segments) are kept
  // in 31 bits (buffer loads: 32-bit voffset + soffset; beyond num_records a raw buffer load silently returns 0)
  return vec && g.M % 256 == 0 && g.N % 256 == 0 && g.K % (2 * BK) == 0 && g.a_ld < (1 << 21) && g.b_ld < (1 << 21) &&
         (int64_t)g.nseg * g.a_ss * 4 + (int64_t)(a_kc ? 256 : g.K) * g.a_ld * 4 < (1LL << 31) &&
         (int64_t)g.nseg * g.b_ss * 4 + (int64_t)(b_kc ? 256 : g.K) * g.b_ld * 4 < (1LL << 31) &&
         (int64_t)256 * g.d_ld * 4 < (1LL << 31) && (int64_t)(g.M / 256) * (g.N / 256) * g.nbatch >= 128;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI = EPI_STORE>
static int launch_gemm(GemmArgs g, hipStream_t s, const char *what) {
  g.tiles_m = (int)cdiv(g.M, BM);
  g.tiles_n = (int)cdiv(g.N, BN);
  g.d_vec = aligned(g.D, 16) && g.d_ld % 4 == 0 && g.d_bs % 4 == 0;
  const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.nbatch;
  TCL_REQUIRE(blocks > 0 && blocks < (1LL << 31), TCL_ERR_SHAPE, "%s: grid of %ld blocks", what, (long)blocks);
  // float4 staging needs 16-byte aligned rows along each operand's contiguous axis
  const bool vec = aligned(g.A, 16) && aligned(g.B, 16) && g.a_ld % 4 == 0 && g.b_ld % 4 == 0 && g.a_bs % 4 == 0 &&
                   g.b_bs % 4 == 0 && g.a_ss % 4 == 0 && g.b_ss % 4 == 0 && (A_KC ? g.K % 4 == 0 : g.M % 4 == 0) &&
                   (B_KC ? g.K % 4 == 0 : g.N % 4 == 0);
  // full tiles; and the byte offsets of one block's operand panels (rows x ld, all segments) fit the 31 bits of its buffer loads
  const bool full = vec && g.M % BM == 0 && g.N % BN == 0 && g.K % BK == 0 &&
                    (int64_t)g.nseg * g.a_ss * 4 + (int64_t)(A_KC ? BM : g.K) * g.a_ld * 4 < (1LL << 31) &&
                    (int64_t)g.nseg * g.b_ss * 4 + (int64_t)(B_KC ? BN : g.K) * g.b_ld * 4 < (1LL << 31) && (int64_t)BM * g.d_ld * 4 < (1LL << 31) && g.d_vec;
  if (full)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_KC, B_KC, true, EPI, true>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, s, g);
  else if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_KC, B_KC, true, EPI>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, A_KC, B_KC, false, EPI>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, s, g);
  return check_launch(what);
}

// ------------------------------------------------------------------------------------------ bf16 MFMA GEMM
// Same contract as gemm_f32_kernel (fp32 operands in global memory, fp32 accumulate and output), but the products
// run on v_mfma_f32_32x32x16_bf16.  Operands are converted while they are staged into LDS:
//   NS = 1 (TCL_PREC_BF16)  : x -> bf16(x)                                   one MFMA per fragment pair
//   NS = 2 (TCL_PREC_BF16X3): x -> hi = bf16(x), lo = bf16(x - hi)           hi*hi + hi*lo + lo*hi (fp32-class)
// LDS images (bf16), one per operand and split level:
//   KC operand (k contiguous in global):  [row][k], 80-byte rows  -> fragment = one ds_read_b128 (conflict-free)
//   MC operand (row contiguous in global): [k][row], row stride = ROWS + 32 elements -> fragment = two
//     ds_read_b64_tr_b16 (the hardware transposing read; conflict-free at this stride), so that neither the
//     NCHW feature planes nor the cotangent volume ever need a transposed copy in HBM.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BKH = 32;            // k depth of one staged tile (bf16 elements): two 32x32x16 MFMA k-steps
constexpr int KC_LDH = BKH + 8;    // [row][k] image row (elements): 80 bytes
template <int ROWS> struct McLd { static constexpr int value = ROWS + 32; };  // [k][row] image row (elements)

template <int ROWS, bool KC> struct ImgSize { static constexpr int value = KC ? ROWS * KC_LDH : BKH * McLd<ROWS>::value; };

template <int ROWS, int NT, bool KC, bool VEC, int NS>
struct TileLoaderH {
  static constexpr int NV = ROWS * BKH / 4 / NT;  // float4 per thread
  static_assert(ROWS * BKH / 4 % NT == 0, "tile must divide over the block");
  f32x4 r[NV];

  __device__ __forceinline__ void load(const float *base, int64_t ld, int row0, int k0, int rows, int depth, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      int row, k;
      if (KC) { row = idx / (BKH / 4); k = (idx % (BKH / 4)) * 4; }
      else    { k = idx / (ROWS / 4); row = (idx % (ROWS / 4)) * 4; }
      const int gr = row0 + row, gk = k0 + k;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (KC) {
        if (gr < rows) {
          const float *p = base + (int64_t)gr * ld + gk;
          if (VEC) { if (gk < depth) val = *reinterpret_cast<const f32x4 *>(p); }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gk + e < depth) val[e] = p[e];
          }
        }
      } else {
        if (gk < depth) {
          const float *p = base + (int64_t)gk * ld + gr;
          if (VEC) { if (gr < rows) val = *reinterpret_cast<const f32x4 *>(p); }
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gr + e < rows) val[e] = p[e];
          }
        }
      }
      r[i] = val;
    }
  }

  // Full tiles (see TileLoader): block-uniform strides, no bounds checks
  __device__ __forceinline__ static int64_t step(int64_t ld) { return KC ? (NT / (BKH / 4)) * ld : (NT / (ROWS / 4)) * ld; }
  __device__ __forceinline__ static int64_t kstep(int64_t ld) { return KC ? BKH : BKH * ld; }
  // The same loads as `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`: a descriptor on the operand's (row0, k-tile 0)
  // element, loop-invariant per-thread 32-bit byte offsets, and the k-tile as a scalar byte offset: no vector address
  // arithmetic per load (hipcc keeps `uniform pointer + zext(offset)` as a 64-bit vector add per load otherwise).
  __device__ __forceinline__ static void thread_offsets(uint32_t (&off)[NV], int64_t ld, int tid) {
    // 32-bit arithmetic, one multiply: float4 i lies a block-uniform step(ld) floats behind float4 i - 1
    const uint32_t ld4 = (uint32_t)ld * 4u, st4 = (uint32_t)step(ld) * 4u;
    off[0] = KC ? (uint32_t)(tid / (BK / 4)) * ld4 + (uint32_t)(tid % (BK / 4)) * 16u : (uint32_t)(tid / (ROWS / 4)) * ld4 + (uint32_t)(tid % (ROWS / 4)) * 16u;
#pragma unroll
    for (int i = 1; i < NV; ++i) off[i] = off[i - 1] + st4;
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rsrc, const uint32_t (&off)[NV], int soff) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[i], soff, 0);
      r[i] = __builtin_bit_cast(f32x4, v);
    }
  }

  // img_hi / img_lo: the operand's LDS images (lo unused when NS == 1)
  __device__ __forceinline__ void store(__bf16 *img_hi, __bf16 *img_lo, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      int off;
      if (KC) { const int row = idx / (BKH / 4), k = (idx % (BKH / 4)) * 4; off = row * KC_LDH + k; }
      else    { const int k = idx / (ROWS / 4), row = (idx % (ROWS / 4)) * 4; off = k * McLd<ROWS>::value + row; }
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (__bf16)r[i][e];
      *reinterpret_cast<bf16x4 *>(img_hi + off) = h;
      if (NS == 2) {
        bf16x4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = (__bf16)(r[i][e] - (float)h[e]);
        *reinterpret_cast<bf16x4 *>(img_lo + off) = l;
      }
    }
  }
};

// A k-contiguous operand that already IS a bf16 image in global memory (the hi / lo images corr_normalise* leaves in the forward
// workspace): 16-byte pieces (8 k's of one row) straight into the [row][k] LDS image, no conversion in the loop.
struct Bf16Ops { const __bf16 *a_hi, *a_lo, *b_hi, *b_lo; };

template <int ROWS, int NT>
struct TileLoaderKC16 {
  static constexpr int NV = ROWS * BKH / 8 / NT;
  static_assert(ROWS * BKH / 8 % NT == 0 && NV >= 1, "tile must divide over the block");
  u32x4 r[NV];
  __device__ __forceinline__ static void thread_offsets(uint32_t (&off)[NV], int64_t ld, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      off[i] = ((uint32_t)(idx / (BKH / 8)) * (uint32_t)ld + (uint32_t)(idx % (BKH / 8)) * 8u) * 2u;
    }
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rsrc, const uint32_t (&off)[NV], int soff) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[i], soff, 0);
  }
  __device__ __forceinline__ void load(const __bf16 *base, int64_t ld, int row0, int k0, int rows, int depth, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT, row = row0 + idx / (BKH / 8), k = k0 + (idx % (BKH / 8)) * 8;
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (row < rows && k + j < depth) ? reinterpret_cast<const unsigned short *>(base)[(int64_t)row * ld + k + j] : (unsigned short)0;
      const u32x4 v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                       (unsigned)e[6] | ((unsigned)e[7] << 16)};
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(__bf16 *img, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      *reinterpret_cast<u32x4 *>(img + (idx / (BKH / 8)) * KC_LDH + (idx % (BKH / 8)) * 8) = r[i];
    }
  }
};

// fragment of the 32x32x16 MFMA for the 32 rows starting at row0, k-step ks (0/1) of the staged tile:
// lane (r = lane & 31, h = lane >> 5) gets element (row0 + r, k = 16 ks + 8 h + j), j = 0..7.
template <int ROWS, bool KC>
__device__ __forceinline__ bf16x8 read_frag_h(const __bf16 *img, int row0, int ks, int lane) {
  if (KC) return *reinterpret_cast<const bf16x8 *>(img + (row0 + (lane & 31)) * KC_LDH + ks * 16 + (lane >> 5) * 8);
  // transposing read: per 16-lane group a 4 (k) x 16 (row) block, lane 4q+p supplies (k row q, rows 4p..4p+3) and
  // receives row (lane & 15) of the block for the 4 k's.  Groups 0/1 = rows 0-15 / 16-31 at h = 0, groups 2/3 at h = 1.
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const __bf16 *a = img + (ks * 16 + (g >> 1) * 8 + q) * McLd<ROWS>::value + row0 + (g & 1) * 16 + 4 * p;
  const s16x4 k03 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)a);
  const s16x4 k47 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(a + 4 * McLd<ROWS>::value));
  // whole-vector reinterpretation (an element-wise bit_cast loop is mis-lowered by hipcc 7.2: it duplicates element 0)
  const s16x8 both = __builtin_shufflevector(k03, k47, 0, 1, 2, 3, 4, 5, 6, 7);
  const bf16x8 f = __builtin_bit_cast(bf16x8, both);
  return f;
}

// A16: the A operand comes from the bf16 hi (/ lo) images `o.a_hi / o.a_lo` (same element order as g.A would have: the backward
// products take r^ / a^ this way, bit-identical to converting the fp32 image in the loop and two thirds of the loop's conversion
// work gone); B (the cotangent volume) is fp32 and converted while it is staged.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, bool VEC, int NS, bool FULL = false, bool A16 = false>
__global__ void __launch_bounds__(WM *WN * 64) gemm_bf16_kernel(GemmArgs g, Bf16Ops o) {
  static_assert(!A16 || A_KC, "bf16 operand images are k-contiguous");
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_IMG = ImgSize<BM, A_KC>::value, B_IMG = ImgSize<BN, B_KC>::value;
  __shared__ __attribute__((aligned(16))) __bf16 smem[NS * (A_IMG + B_IMG)];
  __bf16 *Ah = smem, *Al = smem + (NS - 1) * A_IMG, *Bh = smem + NS * A_IMG, *Bl = Bh + (NS - 1) * B_IMG;

  const unsigned per_batch = g.tiles_m * g.tiles_n;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lin / per_batch, local = lin - z * per_batch;
  constexpr unsigned GROUP = 8;
  const unsigned gw = GROUP * g.tiles_n;
  const unsigned gid = local / gw, first_m = gid * GROUP;
  const unsigned gsz = min((unsigned)g.tiles_m - first_m, GROUP);
  const unsigned tm = first_m + (local % gw) % gsz, tn = (local % gw) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);
  const int lr = lane & 31, lh = lane >> 5;

  const float *Ab = A16 ? nullptr : g.A + (int64_t)(z / g.a_bdiv) * g.a_bs;
  const float *Bb = g.B + (int64_t)z * g.b_bs;
  const __bf16 *A16h = A16 ? o.a_hi + (int64_t)(z / g.a_bdiv) * g.a_bs : nullptr;
  const __bf16 *A16l = (A16 && NS == 2) ? o.a_lo + (int64_t)(z / g.a_bdiv) * g.a_bs : nullptr;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoaderH<BM, NT, A_KC, VEC, NS> la;
  TileLoaderH<BN, NT, B_KC, VEC, NS> lb;
  TileLoaderKC16<BM, NT> la16[NS];
  const int ktiles = (g.K + BKH - 1) / BKH;
  const int total = ktiles * g.nseg;

  // FULL: buffer-load cursors as in gemm_f32_kernel (descriptor + loop-invariant offsets + scalar k offset; the launcher
  // checks the 31-bit bound)
  const int64_t a_k = la.kstep(g.a_ld), b_k = lb.kstep(g.b_ld);
  uint32_t a_off[la.NV], b_off[lb.NV], a16_off[la16[0].NV];
  __amdgpu_buffer_rsrc_t ra, rb, ra16[NS];
  int sa = 0, sb = 0;   // sa counts ELEMENTS x 4 (bytes of the fp32 image; the bf16 images take half)
  if (FULL) {
    if (A16) {
      la16[0].thread_offsets(a16_off, g.a_ld, tid);
      ra16[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(A16h + (int64_t)m0 * g.a_ld), 0, 0x7fffffff, 0x00020000);
      if (NS == 2) ra16[NS - 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(A16l + (int64_t)m0 * g.a_ld), 0, 0x7fffffff, 0x00020000);
    } else {
      la.thread_offsets(a_off, g.a_ld, tid);
      ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Ab + (A_KC ? (int64_t)m0 * g.a_ld : m0)), 0, 0x7fffffff, 0x00020000);
    }
    lb.thread_offsets(b_off, g.b_ld, tid);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bb + (B_KC ? (int64_t)n0 * g.b_ld : n0)), 0, 0x7fffffff, 0x00020000);
  }
  auto load_a = [&](int t_abs) {   // k-tile t_abs (over the segments) of A into registers
    if (A16) {
      if (FULL) {
#pragma unroll
        for (int h = 0; h < NS; ++h) la16[h].load_buf(ra16[h], a16_off, sa / 2);
      } else {
        const int seg = t_abs / ktiles, kt = t_abs - seg * ktiles;
        la16[0].load(A16h + seg * g.a_ss, g.a_ld, m0, kt * BKH, g.M, g.K, tid);
        if (NS == 2) la16[NS - 1].load(A16l + seg * g.a_ss, g.a_ld, m0, kt * BKH, g.M, g.K, tid);
      }
    } else if (FULL) la.load_buf(ra, a_off, sa);
    else {
      const int seg = t_abs / ktiles, kt = t_abs - seg * ktiles;
      la.load(Ab + seg * g.a_ss, g.a_ld, m0, kt * BKH, g.M, g.K, tid);
    }
  };
  int kt_next = 1;
  load_a(0);
  if (FULL) lb.load_buf(rb, b_off, 0);
  else lb.load(Bb, g.b_ld, n0, 0, g.N, g.K, tid);
  for (int t = 0; t < total; ++t) {
    if (A16) {
      la16[0].store(Ah, tid);
      if (NS == 2) la16[NS - 1].store(Al, tid);
    } else la.store(Ah, Al, tid);
    lb.store(Bh, Bl, tid);
    __syncthreads();
    if (t + 1 < total) {
      if (FULL) {
        if (kt_next == ktiles) { kt_next = 0; sa += (int)(g.a_ss - (ktiles - 1) * a_k) * 4; sb += (int)(g.b_ss - (ktiles - 1) * b_k) * 4; }
        else { sa += (int)a_k * 4; sb += (int)b_k * 4; }
        ++kt_next;
        load_a(t + 1);
        lb.load_buf(rb, b_off, sb);
      } else {
        const int seg = (t + 1) / ktiles, kt = (t + 1) - seg * ktiles;
        load_a(t + 1);
        lb.load(Bb + seg * g.b_ss, g.b_ld, n0, kt * BKH, g.N, g.K, tid);
      }
    }
#pragma unroll
    for (int ks = 0; ks < BKH / 16; ++ks) {
      bf16x8 ah[TM], bh[TN], al[TM], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = read_frag_h<BM, A_KC>(Ah, wm0 + i * 32, ks, lane);
        if (NS == 2) al[i] = read_frag_h<BM, A_KC>(Al, wm0 + i * 32, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = read_frag_h<BN, B_KC>(Bh, wn0 + j * 32, ks, lane);
        if (NS == 2) bl[j] = read_frag_h<BN, B_KC>(Bl, wn0 + j * 32, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NS == 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);   // transposed tiles, see the epilogue
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // The tiles were accumulated transposed (B fragment as the MFMA's A operand): a lane holds ONE row m of D and, in
  // each register quad, four consecutive columns n -> one 16-byte store per quad instead of four 4-byte stores.
  float *Db = g.D + (int64_t)z * g.d_bs;
  if (TM == 2 && TN == 2 && WM == 2 && WN == 2 && g.d_vec && m0 + BM <= g.M && n0 + BN <= g.N) {   // block-uniform
    // full tile: LDS-transposed epilogue with whole-line non-temporal stores, as in gemm_f32_kernel (this variant is bound by
    // the volume store)
    constexpr int ELD = 68;
    static_assert(64 * ELD * 4 <= NS * (A_IMG + B_IMG) * 2, "epilogue staging must fit the operand images");
    float *stage = reinterpret_cast<float *>(smem);
    const int wm = wid / WN, wn = wid % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (i + j > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 val = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4 *>(stage + (wm * 32 + lr) * ELD + wn * 32 + 8 * q + 4 * lh) = val;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = tid + NT * k, row = idx >> 4, c4 = idx & 15, seg = c4 >> 3, within = (c4 & 7) * 4;
          const f32x4 val = *reinterpret_cast<const f32x4 *>(stage + row * ELD + seg * 32 + within);
          const int m = m0 + (row >> 5) * (BM / WM) + i * 32 + (row & 31), n = n0 + seg * (BN / WN) + j * 32 + within;
          __builtin_nontemporal_store(val, reinterpret_cast<f32x4 *>(Db + (int64_t)m * g.d_ld + n));
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = m0 + wm0 + i * 32 + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn0 + j * 32 + 8 * q + 4 * lh;
        float *dst = Db + (int64_t)m * g.d_ld + n;
        if (m < g.M) {
          if (g.d_vec && n + 3 < g.N) {
            const f32x4 val = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            *reinterpret_cast<f32x4 *>(dst) = val;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < g.N) dst[e] = acc[i][j][4 * q + e];
          }
        }
      }
    }
}

// ---- forward contraction on ready-made bf16 operands (TCL_PREC_BF16 / BF16X3).  Both operands k-major ([c][pixel], the
// NCHW order the normalisation writes them in); `hi` (and `lo`) images come from corr_normalise*: the k-loop holds no
// conversion, only 16-byte loads -> 16-byte LDS writes -> transposing fragment reads -> MFMAs.  Two LDS stages for NS = 1
// (ONE barrier per k-tile), one stage for NS = 2 (hi + lo images of both operands already fill 40 KB).  The epilogues are
// gemm_epilogue's: whole-line non-temporal stores through LDS, or the fused row argmax (EPI_ARGMAX: the volume never exists).
template <int ROWS, int NT, int BKT>
struct TileLoaderB {
  static constexpr int NV = ROWS * BKT / 8 / NT;   // 16-byte pieces (8 bf16) per thread
  static_assert(ROWS * BKT / 8 % NT == 0 && NV >= 1, "tile must divide over the block");
  u32x4 r[NV];
  __device__ __forceinline__ static void thread_offsets(uint32_t (&off)[NV], int64_t ld, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      off[i] = ((uint32_t)(idx / (ROWS / 8)) * (uint32_t)ld + (uint32_t)(idx % (ROWS / 8)) * 8u) * 2u;
    }
  }
  __device__ __forceinline__ void load_buf(__amdgpu_buffer_rsrc_t rsrc, const uint32_t (&off)[NV], int soff) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off[i], soff, 0);
  }
  // ragged shapes: element-wise, bounds-checked (rows >= `rows` and k >= `depth` read as zero)
  __device__ __forceinline__ void load(const __bf16 *base, int64_t ld, int row0, int k0, int rows, int depth, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT, k = k0 + idx / (ROWS / 8), row = row0 + (idx % (ROWS / 8)) * 8;
      unsigned short e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (k < depth && row + j < rows) ? reinterpret_cast<const unsigned short *>(base)[(int64_t)k * ld + row + j] : (unsigned short)0;
      const u32x4 v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                       (unsigned)e[6] | ((unsigned)e[7] << 16)};
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(__bf16 *img, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * NT;
      *reinterpret_cast<u32x4 *>(img + (idx / (ROWS / 8)) * McLd<ROWS>::value + (idx % (ROWS / 8)) * 8) = r[i];
    }
  }
};

// Pipeline (the wide fp32 kernel's, with two 16-deep MFMA groups per 32-deep k-tile): two LDS stages and ONE barrier per k-tile,
// placed between the tile's two groups.  Before it a wave reads its group-1 fragments, writes its share of the NEXT tile to the
// other stage, issues the loads of the tile after that, and multiplies group 0 (whose fragments it read one phase earlier);
// after it, the first fragments of the next tile are read while group 1 is multiplied.  Every LDS read has a whole group of
// MFMAs (4 or 12) to land under; the matrix pipe of a SIMD never waits on its own wave's LDS round trip.
// D = k-tiles in flight in registers beyond the staged one (the loop is unrolled D times, ktiles % D == 0): with one tile a wave
// had 4 x 16 bytes per lane on their way from L2 for ~0.3 us of MFMA work per k-tile, less than the loaded L2 latency.
template <int BM, int BN, int WM, int WN, int NS, int EPI, bool FULL, int D>
__global__ void __launch_bounds__(WM *WN * 64, (BM / WM) * (BN / WN) > 128 * 64 ? 1 : ((NS == 1 && (BM / WM) * (BN / WN) <= 64 * 64) ? 3 : 2)) gemm_bf16op_kernel(GemmArgs g, Bf16Ops o) {
  constexpr int NT = WM * WN * 64, BKT = 32;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_IMG = BKT * McLd<BM>::value, B_IMG = BKT * McLd<BN>::value, STAGE = NS * (A_IMG + B_IMG);
  constexpr int SMEM_F = 2 * STAGE / 2;   // the buffer in floats (epilogue staging)
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_bf16op[];
  __bf16 *smem = smem_bf16op;

  const unsigned per_batch = g.tiles_m * g.tiles_n;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lin / per_batch, local = lin - z * per_batch;
  constexpr unsigned GROUP = 8;
  const unsigned gw = GROUP * g.tiles_n;
  const unsigned gid = local / gw, first_m = gid * GROUP;
  const unsigned gsz = min((unsigned)g.tiles_m - first_m, GROUP);
  const unsigned tm = first_m + (local % gw) % gsz, tn = (local % gw) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);

  const int64_t a_base = (int64_t)(z / g.a_bdiv) * g.a_bs, b_base = (int64_t)z * g.b_bs;
  f32x16 acc[TM][TN];   // defined by the first MFMA group (C = inline zero: no zero fill, 64 vector instructions less per wave)

  TileLoaderB<BM, NT, BKT> la[D][NS];
  TileLoaderB<BN, NT, BKT> lb[D][NS];
  const int ktiles = (g.K + BKT - 1) / BKT;
  uint32_t a_off[la[0][0].NV], b_off[lb[0][0].NV];
  __amdgpu_buffer_rsrc_t ra[NS], rb[NS];
  if (FULL) {
    la[0][0].thread_offsets(a_off, g.a_ld, tid);
    lb[0][0].thread_offsets(b_off, g.b_ld, tid);
    ra[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.a_hi + a_base + m0), 0, 0x7fffffff, 0x00020000);
    rb[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.b_hi + b_base + n0), 0, 0x7fffffff, 0x00020000);
    if (NS == 2) {
      ra[NS - 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.a_lo + a_base + m0), 0, 0x7fffffff, 0x00020000);
      rb[NS - 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.b_lo + b_base + n0), 0, 0x7fffffff, 0x00020000);
    }
  }
  const int a_k = (int)(BKT * g.a_ld * 2), b_k = (int)(BKT * g.b_ld * 2);   // bytes from one k-tile to the next
  auto fetch = [&](auto setc, int t) {   // k-tile t -> register set SET
    constexpr int SET = decltype(setc)::value;
    if (FULL) {
#pragma unroll
      for (int h = 0; h < NS; ++h) { la[SET][h].load_buf(ra[h], a_off, t * a_k); lb[SET][h].load_buf(rb[h], b_off, t * b_k); }
    } else {
      la[SET][0].load(o.a_hi + a_base, g.a_ld, m0, t * BKT, g.M, g.K, tid);
      lb[SET][0].load(o.b_hi + b_base, g.b_ld, n0, t * BKT, g.N, g.K, tid);
      if (NS == 2) {
        la[SET][NS - 1].load(o.a_lo + a_base, g.a_ld, m0, t * BKT, g.M, g.K, tid);
        lb[SET][NS - 1].load(o.b_lo + b_base, g.b_ld, n0, t * BKT, g.N, g.K, tid);
      }
    }
  };
  auto stage = [&](auto setc, int st) {   // register set SET -> LDS stage st: [A hi][A lo][B hi][B lo]
    constexpr int SET = decltype(setc)::value;
    __bf16 *base = smem + st * STAGE;
#pragma unroll
    for (int h = 0; h < NS; ++h) { la[SET][h].store(base + h * A_IMG, tid); lb[SET][h].store(base + NS * A_IMG + h * B_IMG, tid); }
  };
  struct Frags { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  auto read_frags = [&](Frags &f, int st, int ks) {
    const __bf16 *Ah = smem + st * STAGE, *Al = Ah + (NS - 1) * A_IMG, *Bh = Ah + NS * A_IMG, *Bl = Bh + (NS - 1) * B_IMG;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f.ah[i] = read_frag_h<BM, false>(Ah, wm0 + i * 32, ks, lane);
      if (NS == 2) f.al[i] = read_frag_h<BM, false>(Al, wm0 + i * 32, ks, lane);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = read_frag_h<BN, false>(Bh, wn0 + j * 32, ks, lane);
      if (NS == 2) f.bl[j] = read_frag_h<BN, false>(Bl, wn0 + j * 32, ks, lane);
    }
  };
  auto mma = [&](const Frags &f, auto firstc) {
    constexpr bool FIRST = decltype(firstc)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (NS == 2) {   // the small cross terms first, hi * hi last (as before: only the first C operand changes)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.al[i], FIRST ? zero : acc[i][j], 0, 0, 0);   // transposed tiles, see gemm_epilogue
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.ah[i], (FIRST && NS == 1) ? zero : acc[i][j], 0, 0, 0);
      }
  };
  Frags f0, f1;
  using S0 = std::integral_constant<int, 0>;
  fetch(S0{}, 0);
  stage(S0{}, 0);
  {   // tiles 1 .. D on their way: tile j in set j % D
    auto pre = [&](auto jc) { constexpr int J = decltype(jc)::value; if (J <= D) fetch(std::integral_constant<int, J % D>{}, min(J, ktiles - 1)); };
    pre(std::integral_constant<int, 1>{}); pre(std::integral_constant<int, 2>{}); pre(std::integral_constant<int, 3>{}); pre(std::integral_constant<int, 4>{});
    static_assert(D <= 4, "prologue covers D <= 4");
  }
  __syncthreads();
  read_frags(f0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // The body is straight-line (no branch: hipcc's waits then COUNT the outstanding LDS / memory operations instead of draining
  // them at a join).  Past the last tile the staging simply runs on: the loads re-read the last tile, the other stage receives
  // a copy nobody multiplies.
  auto ktile = [&](auto uc, int t, auto firstc) {   // k-tile t + U of an unrolled trip (t % D == 0); firstc: the kernel's first trip
    constexpr int U = decltype(uc)::value;
    if (U >= D) return;
    const int st = D % 2 == 0 ? (U & 1) : ((t + U) & 1);
    using NEXT = std::integral_constant<int, (U + 1) % D>;   // the set that holds tile t + U + 1
    read_frags(f1, st, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0, std::integral_constant<bool, decltype(firstc)::value && U == 0>{});   // queued BEFORE the wave may have to wait for the next tile's global loads
    __builtin_amdgcn_sched_barrier(0);
    stage(NEXT{}, 1 - st);               // everyone left that stage before the barrier of the previous k-tile
    fetch(NEXT{}, min(t + U + 1 + D, ktiles - 1));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // all waves: group-1 fragments of this tile in registers, the next tile staged
    read_frags(f0, 1 - st, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto trip = [&](auto firstc, int t) {
    ktile(std::integral_constant<int, 0>{}, t, firstc); ktile(std::integral_constant<int, 1>{}, t, firstc);
    ktile(std::integral_constant<int, 2>{}, t, firstc); ktile(std::integral_constant<int, 3>{}, t, firstc);
  };
  trip(std::true_type{}, 0);   // peeled: its first MFMAs define the accumulators
#pragma unroll 1
  for (int t = D; t < ktiles; t += D) trip(std::false_type{}, t);
  __syncthreads();
  gemm_epilogue<BM, BN, WM, WN, EPI, FULL, SMEM_F>(acc, reinterpret_cast<float *>(smem), g, z, m0, n0, tn);
}

// Backward products of the bf16 variants on the same pipeline (full tiles only; other shapes keep gemm_bf16_kernel):
//   D[m][n] = sum over segments and k of A[m][k] B[n or k][...]:  A = the forward's bf16 image of r^ / a^ (k contiguous, TileLoaderKC16,
//   [row][k] LDS image, plain 16-byte fragment reads), B = the fp32 cotangent volume, k-contiguous (d_a: B_KC) or row-contiguous (d_r),
//   converted to bf16 (hi, lo) while it is staged.  Two LDS stages, one barrier per k-tile, D k-tiles of both operands in flight in
//   registers: the round-2 loop had ONE k-tile (24 KB per block) on its way from HBM, less than the loaded latency x the rate wanted.
template <int BM, int BN, int WM, int WN, int NS, bool B_KC, int D>
__global__ void __launch_bounds__(WM *WN * 64, 2) gemm_bf16bwd_kernel(GemmArgs g, Bf16Ops o) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_IMG = ImgSize<BM, true>::value, B_IMG = ImgSize<BN, B_KC>::value, STAGE = NS * (A_IMG + B_IMG);
  constexpr int SMEM_F = 2 * STAGE / 2;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_bf16bwd[];
  __bf16 *smem = smem_bf16bwd;

  const unsigned per_batch = g.tiles_m * g.tiles_n;
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned z = lin / per_batch, local = lin - z * per_batch;
  constexpr unsigned GROUP = 8;
  const unsigned gw = GROUP * g.tiles_n;
  const unsigned gid = local / gw, first_m = gid * GROUP;
  const unsigned gsz = min((unsigned)g.tiles_m - first_m, GROUP);
  const unsigned tm = first_m + (local % gw) % gsz, tn = (local % gw) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm0 = (wid / WN) * (BM / WM), wn0 = (wid % WN) * (BN / WN);

  const int64_t a_base = (int64_t)(z / g.a_bdiv) * g.a_bs + (int64_t)m0 * g.a_ld;
  const float *Bb = g.B + (int64_t)z * g.b_bs + (B_KC ? (int64_t)n0 * g.b_ld : n0);
  f32x16 acc[TM][TN];

  TileLoaderKC16<BM, NT> la[D][NS];
  TileLoaderH<BN, NT, B_KC, true, NS> lb[D];
  const int ktiles = g.K / BKH, total = ktiles * g.nseg;
  uint32_t a_off[la[0][0].NV], b_off[lb[0].NV];
  la[0][0].thread_offsets(a_off, g.a_ld, tid);
  lb[0].thread_offsets(b_off, g.b_ld, tid);
  __amdgpu_buffer_rsrc_t ra[NS], rb;
  ra[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.a_hi + a_base), 0, 0x7fffffff, 0x00020000);
  if (NS == 2) ra[NS - 1] = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(o.a_lo + a_base), 0, 0x7fffffff, 0x00020000);
  rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bb), 0, 0x7fffffff, 0x00020000);
  const int a_k = BKH, b_k = (int)lb[0].kstep(g.b_ld);   // elements from one k-tile to the next
  auto fetch = [&](auto setc, int t) {   // k-tile t (counted over the segments) -> register set SET
    constexpr int SET = decltype(setc)::value;
    const int seg = g.nseg == 1 ? 0 : t / ktiles, kt = t - seg * ktiles;
    const int sa = (seg * (int)g.a_ss + kt * a_k) * 2, sb = (seg * (int)g.b_ss + kt * b_k) * 4;
#pragma unroll
    for (int h = 0; h < NS; ++h) la[SET][h].load_buf(ra[h], a_off, sa);
    lb[SET].load_buf(rb, b_off, sb);
  };
  auto stage = [&](auto setc, int st) {   // register set SET -> LDS stage st: [A hi][A lo][B hi][B lo]
    constexpr int SET = decltype(setc)::value;
    __bf16 *base = smem + st * STAGE;
#pragma unroll
    for (int h = 0; h < NS; ++h) la[SET][h].store(base + h * A_IMG, tid);
    lb[SET].store(base + NS * A_IMG, base + NS * A_IMG + (NS - 1) * B_IMG, tid);
  };
  struct Frags { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; };
  auto read_frags = [&](Frags &f, int st, int ks) {
    const __bf16 *Ah = smem + st * STAGE, *Al = Ah + (NS - 1) * A_IMG, *Bh = Ah + NS * A_IMG, *Bl = Bh + (NS - 1) * B_IMG;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f.ah[i] = read_frag_h<BM, true>(Ah, wm0 + i * 32, ks, lane);
      if (NS == 2) f.al[i] = read_frag_h<BM, true>(Al, wm0 + i * 32, ks, lane);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = read_frag_h<BN, B_KC>(Bh, wn0 + j * 32, ks, lane);
      if (NS == 2) f.bl[j] = read_frag_h<BN, B_KC>(Bl, wn0 + j * 32, ks, lane);
    }
  };
  auto mma = [&](const Frags &f, auto firstc) {
    constexpr bool FIRST = decltype(firstc)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (NS == 2) {   // same term order as gemm_bf16_kernel: results stay bit-identical with the round-2 kernel
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.al[i], FIRST ? zero : acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.bh[j], f.ah[i], (FIRST && NS == 1) ? zero : acc[i][j], 0, 0, 0);
      }
  };
  Frags f0, f1;
  using S0 = std::integral_constant<int, 0>;
  fetch(S0{}, 0);
  stage(S0{}, 0);
  {
    auto pre = [&](auto jc) { constexpr int J = decltype(jc)::value; if (J <= D) fetch(std::integral_constant<int, J % D>{}, min(J, total - 1)); };
    pre(std::integral_constant<int, 1>{}); pre(std::integral_constant<int, 2>{});
    static_assert(D <= 2, "prologue covers D <= 2");
  }
  __syncthreads();
  read_frags(f0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  auto ktile = [&](auto uc, int t, auto firstc) {   // as in gemm_bf16op_kernel
    constexpr int U = decltype(uc)::value;
    if (U >= D) return;
    const int st = D % 2 == 0 ? (U & 1) : ((t + U) & 1);
    using NEXT = std::integral_constant<int, (U + 1) % D>;
    read_frags(f1, st, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0, std::integral_constant<bool, decltype(firstc)::value && U == 0>{});
    __builtin_amdgcn_sched_barrier(0);
    stage(NEXT{}, 1 - st);
    fetch(NEXT{}, min(t + U + 1 + D, total - 1));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(f0, 1 - st, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
  };
  auto trip = [&](auto firstc, int t) { ktile(std::integral_constant<int, 0>{}, t, firstc); ktile(std::integral_constant<int, 1>{}, t, firstc); };
  trip(std::true_type{}, 0);
#pragma unroll 1
  for (int t = D; t < total; t += D) trip(std::false_type{}, t);
  __syncthreads();
  gemm_epilogue<BM, BN, WM, WN, EPI_STORE, true, SMEM_F>(acc, reinterpret_cast<float *>(smem), g, z, m0, n0, tn);
}

// Measured at C2 (round 3, profiles/r3_bf16_kloop_ablations.txt): 128 x 128 tiles with two k-tiles in flight.  256 x 128 tiles
// (4 waves of 128 x 64, or 8 of 64 x 64): store variant 0.258 / 0.287 ms against 0.224, fused argmax 0.228 / 0.241 against 0.225;
// 256 x 256 tiles (4 waves of 128 x 128, one wave per SIMD, this same two-group pipeline): fused argmax kernel 223 us against 157;
// one or four k-tiles in flight: 0.215 / 0.246 ms against 0.222.  The matrix pipe is busy 0.40 of the kernel's cycles
// (SQ_VALU_MFMA_BUSY_CYCLES; 2.1 GHz under this load).  Where the k-loop's 157 us go (side builds without parts of the loop, no
// epilogue: the epilogues are hidden under other blocks' loops): fragment reads + MFMAs alone 98 us (a 64 x 64 wave tile needs as
// many LDS-array cycles for its transposing reads as matrix-pipe cycles for its MFMAs), the ds_write_b128 staging +41 us (13
// cycles of VGPR -> LDS transfer per wave instruction, which the MFMAs do not overlap), the global loads and their waits +18 us,
// the barrier nothing.  K = 1024 instead of 256 takes 4.0x as long: steady state, not per-tile start-up.
constexpr int BF16_D = 2;
#define TCL_BF16_TILE 128, 128, 2, 2
// dynamic LDS above 64 KB needs the kernel attribute raised once per device (cached under a mutex, like wide_kernel_ready)
template <typename K>
static bool bf16op_lds_ready(K kernel, size_t lds) {
  if (lds <= 64 * 1024) return true;
  // keyed by (kernel, device): every instantiation has the same function-pointer type, so a per-type cache would ask for one only
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, bool> asked;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair((const void *)kernel, dev);
  auto it = asked.find(key);
  if (it == asked.end()) {
    const bool granted = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!granted) (void)hipGetLastError();
    it = asked.emplace(key, granted).first;
  }
  return it->second;
}

template <int BM, int BN, int WM, int WN, int EPI>
static int launch_gemm_bf16op(GemmArgs g, const Bf16Ops &o, int nsplit, hipStream_t s, const char *what) {
  g.tiles_m = (int)cdiv(g.M, BM);
  g.tiles_n = (int)cdiv(g.N, BN);
  g.d_vec = aligned(g.D, 16) && g.d_ld % 4 == 0 && g.d_bs % 4 == 0;
  const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.nbatch;
  TCL_REQUIRE(blocks > 0 && blocks < (1LL << 31), TCL_ERR_SHAPE, "%s: grid of %ld blocks", what, (long)blocks);
  // full tiles, 16-byte aligned operand rows, and the byte offsets of one block's operand panels (K x ld) inside 31 bits
  const bool full = g.M % BM == 0 && g.N % BN == 0 && g.K % 32 == 0 && g.a_ld % 8 == 0 && g.b_ld % 8 == 0 && g.a_bs % 8 == 0 && g.b_bs % 8 == 0 &&
                    aligned(o.a_hi, 16) && aligned(o.b_hi, 16) && aligned(o.a_lo, 16) && aligned(o.b_lo, 16) &&
                    (int64_t)g.K * g.a_ld * 2 < (1LL << 31) && (int64_t)g.K * g.b_ld * 2 < (1LL << 31) &&
                    (EPI != EPI_STORE || (g.d_vec && (int64_t)BM * g.d_ld * 4 < (1LL << 31)));
  const dim3 grd((unsigned)blocks), blk(WM * WN * 64);
  const size_t lds = (size_t)2 * nsplit * 32 * (McLd<BM>::value + McLd<BN>::value) * sizeof(__bf16);
  const int ktiles = (int)cdiv(g.K, 32);
#define TCL_LAUNCH_BF16OP_D(NS, FULLV, DV)                                                                                \
  do {                                                                                                                     \
    auto kern = gemm_bf16op_kernel<BM, BN, WM, WN, NS, EPI, FULLV, DV>;                                                    \
    TCL_REQUIRE(bf16op_lds_ready(kern, lds), TCL_ERR_LAUNCH, "%s: the device refused %zu bytes of dynamic LDS", what, lds); \
    hipLaunchKernelGGL(kern, grd, blk, lds, s, g, o);                                                                      \
  } while (0)
#define TCL_LAUNCH_BF16OP(NS, FULLV)                                            \
  do {                                                                           \
    if (FULLV && ktiles % BF16_D == 0) TCL_LAUNCH_BF16OP_D(NS, FULLV, BF16_D); \
    else TCL_LAUNCH_BF16OP_D(NS, FULLV, 1);                                     \
  } while (0)
  if (nsplit == 2) {
    if (full) TCL_LAUNCH_BF16OP(2, true);
    else TCL_LAUNCH_BF16OP(2, false);
  } else {
    if (full) TCL_LAUNCH_BF16OP(1, true);
    else TCL_LAUNCH_BF16OP(1, false);
  }
#undef TCL_LAUNCH_BF16OP_D
#undef TCL_LAUNCH_BF16OP
  return check_launch(what);
}

static bool gemm_vec_ok(const GemmArgs &g, bool a_kc, bool b_kc) {
  return aligned(g.A, 16) && aligned(g.B, 16) && g.a_ld % 4 == 0 && g.b_ld % 4 == 0 && g.a_bs % 4 == 0 && g.b_bs % 4 == 0 &&
         g.a_ss % 4 == 0 && g.b_ss % 4 == 0 && (a_kc ? g.K % 4 == 0 : g.M % 4 == 0) && (b_kc ? g.K % 4 == 0 : g.N % 4 == 0);
}

// A operand from the bf16 images in `o` (a_hi, a_lo for nsplit == 2), B = g.B in fp32
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC>
static int launch_gemm_bf16(GemmArgs g, const Bf16Ops &o, int nsplit, hipStream_t s, const char *what) {
  TCL_REQUIRE(o.a_hi && (nsplit == 1 || o.a_lo), TCL_ERR_ARG, "%s: bf16 operand images missing", what);
  g.tiles_m = (int)cdiv(g.M, BM);
  g.tiles_n = (int)cdiv(g.N, BN);
  g.d_vec = aligned(g.D, 16) && g.d_ld % 4 == 0 && g.d_bs % 4 == 0;
  const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.nbatch;
  TCL_REQUIRE(blocks > 0 && blocks < (1LL << 31), TCL_ERR_SHAPE, "%s: grid of %ld blocks", what, (long)blocks);
  // B (fp32) alone decides the vector width of the ragged path; the full-tile path also needs 16-byte aligned rows of the bf16 images
  GemmArgs gb = g; gb.A = g.B; gb.a_ld = g.b_ld; gb.a_bs = g.b_bs; gb.a_ss = g.b_ss;
  const bool vec = gemm_vec_ok(gb, B_KC, B_KC);
  const bool a16_vec = aligned(o.a_hi, 16) && aligned(o.a_lo, 16) && g.a_ld % 8 == 0 && g.a_bs % 8 == 0 && g.a_ss % 8 == 0;
  const dim3 grd((unsigned)blocks), blk(WM * WN * 64);
  if (vec && a16_vec && g.M % BM == 0 && g.N % BN == 0 && g.K % BKH == 0 && (int64_t)BM * g.d_ld * 4 < (1LL << 31) &&
      (int64_t)g.nseg * g.a_ss * 4 + (int64_t)(A_KC ? BM : g.K) * g.a_ld * 4 < (1LL << 31) &&
      (int64_t)g.nseg * g.b_ss * 4 + (int64_t)(B_KC ? BN : g.K) * g.b_ld * 4 < (1LL << 31)) {   // full tiles (31-bit buffer offsets): cursor loads, no predication
    // the pipelined kernel on 256 x 128 eight-wave tiles when the shape divides (it has no ragged path), else the round-2 loop
    if (A_KC && g.M % 256 == 0 && g.N % 128 == 0 && (int64_t)g.nseg * g.b_ss * 4 + (int64_t)(B_KC ? 128 : g.K) * g.b_ld * 4 < (1LL << 31)) {
      GemmArgs p = g;
      p.tiles_m = g.M / 256; p.tiles_n = g.N / 128;
      const int64_t pblocks = (int64_t)p.tiles_m * p.tiles_n * p.nbatch;
      const int total = g.K / BKH * g.nseg;
#define TCL_BWD_PIPE(NSV, DV)                                                                                       \
  do {                                                                                                               \
    auto kern = gemm_bf16bwd_kernel<256, 128, 4, 2, NSV, B_KC, DV>;                                                  \
    constexpr size_t lds = (size_t)2 * NSV * (ImgSize<256, true>::value + ImgSize<128, B_KC>::value) * 2;            \
    if (pblocks < (1LL << 31) && bf16op_lds_ready(kern, lds)) {                                                      \
      hipLaunchKernelGGL(kern, dim3((unsigned)pblocks), dim3(512), lds, s, p, o);                                    \
      return check_launch(what);                                                                                     \
    }                                                                                                                \
  } while (0)
      // bf16x3 only: with one MFMA per product (bf16) the round-2 loop at four waves per SIMD is faster (0.842 against 0.864-0.885 ms
      // for the leg), with three the pipeline wins (1.47-1.53 against 1.57-1.63 ms)
      if (nsplit == 2 && total % 2 == 0) TCL_BWD_PIPE(2, 2);   // (both products have K = N = P: a multiple of 128 here, so always)
#undef TCL_BWD_PIPE
    }
    if (nsplit == 2) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, true, 2, true, true>), grd, blk, 0, s, g, o);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, true, 1, true, true>), grd, blk, 0, s, g, o);
    return check_launch(what);
  }
  if (nsplit == 2) {
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, true, 2, false, true>), grd, blk, 0, s, g, o);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, false, 2, false, true>), grd, blk, 0, s, g, o);
  } else {
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, true, 1, false, true>), grd, blk, 0, s, g, o);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, A_KC, B_KC, false, 1, false, true>), grd, blk, 0, s, g, o);
  }
  return check_launch(what);
}

// ------------------------------------------------------------------------------------------ workspace layout
struct CorrWs {
  float *a_hat, *r_hat, *n_t, *n_r;      // fp32 operands (NULL when not kept) and the norms
  __bf16 *a_hi, *r_hi, *a_lo, *r_lo;     // bf16 operand images (TCL_PREC_BF16: hi; TCL_PREC_BF16X3: hi + lo), same element order
  size_t bytes;
};
// Layout: [a^ r^ fp32 (TCL_PREC_F32 only)] [norms] [a^ r^ hi (bf16 variants)] [a^ r^ lo (bf16x3)]: every variant keeps exactly the
// operand images its forward and backward products multiply.
static CorrWs corr_ws(void *base, int64_t B, int64_t C, int64_t F, int64_t P, int precision) {
  const bool keep_f32 = precision == TCL_PREC_F32;   // the bf16 variants multiply the bf16 images in the forward AND the backward products
  CorrWs w{};
  size_t off = 0;
  auto take = [&](size_t nbytes) { void *p = (char *)base + off; off += round_up(nbytes, 256); return p; };
  if (keep_f32) {
    w.a_hat = (float *)take((size_t)B * C * P * 4);
    w.r_hat = (float *)take((size_t)B * F * C * P * 4);
  }
  w.n_t = (float *)take((size_t)B * P * 4);
  w.n_r = (float *)take((size_t)B * F * P * 4);
  if (precision != TCL_PREC_F32) {
    w.a_hi = (__bf16 *)take((size_t)B * C * P * 2);
    w.r_hi = (__bf16 *)take((size_t)B * F * C * P * 2);
    if (precision == TCL_PREC_BF16X3) {
      w.a_lo = (__bf16 *)take((size_t)B * C * P * 2);
      w.r_lo = (__bf16 *)take((size_t)B * F * C * P * 2);
    }
  }
  w.bytes = off;
  return w;
}

// ------------------------------------------------------------------------------------------ A2 per-pixel correlation
// Block = 64 consecutive pixels x 4 waves; wave w reduces the channels c = w, w+4, ... (8 channel pairs of loads in flight
// per step), the four partial sums are combined through LDS in wave order (deterministic).
constexpr int C1_U = 8;
__device__ __forceinline__ void corr1x1_sums(const float *p1, const float *p2, int64_t P, int C, int wv, float a, float c2,
                                             float (&red)[3][4][64], int lane, float &s1, float &s2, float &d) {
  float t1s = 0.f, t2s = 0.f, ds = 0.f;
  for (int c0 = wv; c0 < C; c0 += 4 * C1_U) {
    float x1[C1_U], x2[C1_U];
#pragma unroll
    for (int u = 0; u < C1_U; ++u) {
      const int c = min(c0 + 4 * u, C - 1);   // clamped, unconditional loads; padding channels are zeroed below
      x1[u] = p1[(int64_t)c * P];
      x2[u] = p2[(int64_t)c * P];
    }
    TCL_LOADS_FIRST(2 * C1_U);
#pragma unroll
    for (int u = 0; u < C1_U; ++u) {
      const float live = c0 + 4 * u < C ? 1.f : 0.f;
      const float t1 = x1[u] * a * live, t2 = x2[u] * c2 * live;
      t1s += t1 * t1; t2s += t2 * t2; ds += t1 * t2;
    }
  }
  red[0][wv][lane] = t1s; red[1][wv][lane] = t2s; red[2][wv][lane] = ds;
  __syncthreads();
  s1 = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
  s2 = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  d = (red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]);
}

__global__ void __launch_bounds__(256) corr1x1_fwd_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                          const float *__restrict__ m1, const float *__restrict__ m2,
                                                          float *__restrict__ out, int C, int P, int64_t total) {
  __shared__ float red[3][4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t i0 = (int64_t)blockIdx.x * 64 + lane;
  const bool ok = i0 < total;
  const int64_t i = ok ? i0 : total - 1;
  const int64_t b = i / P, p = i - b * P;
  float s1, s2, d;
  corr1x1_sums(f1 + b * C * P + p, f2 + b * C * P + p, P, C, wv, m1[i], m2[i], red, lane, s1, s2, d);
  if (ok && wv == 0) out[i] = d / ((sqrtf(s1) + kNormEps) * (sqrtf(s2) + kNormEps));
}

__global__ void __launch_bounds__(256) corr1x1_bwd_kernel(const float *__restrict__ go, const float *__restrict__ f1,
                                                          const float *__restrict__ f2, const float *__restrict__ m1,
                                                          const float *__restrict__ m2, float *__restrict__ g1,
                                                          float *__restrict__ g2, int C, int P, int64_t total) {
  __shared__ float red[3][4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t i0 = (int64_t)blockIdx.x * 64 + lane;
  const bool ok = i0 < total;
  const int64_t i = ok ? i0 : total - 1;
  const int64_t b = i / P, p = i - b * P;
  const float a = m1[i], c2 = m2[i], g = go[i];
  const float *p1 = f1 + b * C * P + p, *p2 = f2 + b * C * P + p;
  float s1, s2, d;
  corr1x1_sums(p1, p2, P, C, wv, a, c2, red, lane, s1, s2, d);
  const float n1 = sqrtf(s1), n2 = sqrtf(s2), e1 = n1 + kNormEps, e2 = n2 + kNormEps;
  // d_a1^ = g a2^ ; <d_a1^, t1> = g d / e2  (and symmetrically)
  const float k1 = n1 > 0.f ? (g * d / e2) / (e1 * e1 * n1) : 0.f;
  const float k2 = n2 > 0.f ? (g * d / e1) / (e2 * e2 * n2) : 0.f;
  if (!ok) return;
  for (int c = wv; c < C; c += 4) {   // second pass over the wave's channels (L2-resident by now)
    const float t1 = p1[(int64_t)c * P] * a, t2 = p2[(int64_t)c * P] * c2;
    if (g1) g1[b * C * P + (int64_t)c * P + p] = (g * (t2 / e2) / e1 - t1 * k1) * a;
    if (g2) g2[b * C * P + (int64_t)c * P + p] = (g * (t1 / e1) / e2 - t2 * k2) * c2;
  }
}

// ------------------------------------------------------------------------------------------ A10 argmax -> flow
// one wave per target pixel row of S*S similarities; first maximal index wins (torch.argmax on CPU).
__global__ void __launch_bounds__(256) corr_to_flow_kernel(const float *__restrict__ corr, float *__restrict__ flow,
                                                           int64_t rows, int S, float scale) {
  const int Q = S * S;
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
    const float *r = corr + row * Q;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int q = lane; q < Q; q += 64) {
      const float v = r[q];
      if (v > best || bi == 0x7fffffff) { best = v; bi = q; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
      const int x = bi % S, y = bi / S;
      reinterpret_cast<float2 *>(flow)[row] =
          make_float2(__fsub_rn(__fmul_rn((float)x, scale), 1.f), __fsub_rn(__fmul_rn((float)y, scale), 1.f));
    }
  }
}

// rows x ntile (value, column) pairs -> flow; tiles are in column order, so a strict > keeps the first maximum.
__global__ void __launch_bounds__(256) argmax_tiles_to_flow_kernel(const float2 *__restrict__ part, float *__restrict__ flow, int64_t rows,
                                                                  int ntile, int S, float scale) {
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    float2 b = part[row * ntile];
    for (int k = 1; k < ntile; ++k) {
      const float2 o = part[row * ntile + k];
      if (o.x > b.x) b = o;
    }
    const int bi = __float_as_int(b.y), x = bi % S, y = bi / S;
    reinterpret_cast<float2 *>(flow)[row] = make_float2(__fsub_rn(__fmul_rn((float)x, scale), 1.f), __fsub_rn(__fmul_rn((float)y, scale), 1.f));
  }
}

}  // namespace tcl

using namespace tcl;

// ============================================================================================ C ABI
extern "C" size_t tcl_corr4d_fwd_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision) {
  if (B <= 0 || C <= 0 || F <= 0 || P <= 0) return 0;
  return corr_ws(nullptr, B, C, F, P, precision).bytes;
}


// launches the (forward or backward) normalisation with 4 pixels per lane when every row is 16-byte aligned
static bool norm_vec_ok(int64_t P, const void *a, const void *b, const void *c, const void *d, const void *e) {
  return P % 4 == 0 && aligned(a, 16) && aligned(b, 16) && aligned(c, 16) && aligned(d, 16) && aligned(e, 16);
}
static void launch_normalise(hipStream_t s, const float *x, const float *v, HatOut hat, float *nrm, int64_t BF, int C, int F, int64_t P) {
  const bool keep = true;
  const bool vec = norm_vec_ok(P, x, v, hat.f, nrm, nullptr) && aligned(hat.hi, 16) && aligned(hat.lo, 16);
#define TCL_NORM_KEEP(NCG) hipLaunchKernelGGL((corr_normalise_keep_kernel<2, NCG, 16>), dim3((unsigned)cdiv(P, 128), (unsigned)BF), dim3(64 * NCG), 0, s, x, v, hat, nrm, C, F, (int)P)
  if (vec && keep && C <= 256) {   // single read of x: the thread's <= 16 channels stay in registers
    if (C <= 64) TCL_NORM_KEEP(4);
    else if (C <= 128) TCL_NORM_KEEP(8);
    else TCL_NORM_KEEP(16);
  } else
#undef TCL_NORM_KEEP
  if (!vec)
    hipLaunchKernelGGL((corr_normalise_kernel<1, 4>), dim3((unsigned)cdiv(P, 64), (unsigned)BF), dim3(256), 0, s, x, v, hat, nrm, C, F, (int)P);
  else if (cdiv(P, 256) * BF >= 4 * kNumCU || C < 128)   // enough blocks: 4 waves each; else 16 waves share the channels
    hipLaunchKernelGGL((corr_normalise_kernel<4, 4>), dim3((unsigned)cdiv(P, 256), (unsigned)BF), dim3(256), 0, s, x, v, hat, nrm, C, F, (int)P);
  else
    hipLaunchKernelGGL((corr_normalise_kernel<4, 16>), dim3((unsigned)cdiv(P, 256), (unsigned)BF), dim3(1024), 0, s, x, v, hat, nrm, C, F, (int)P);
}
static void launch_normalise_bwd(hipStream_t s, const float *x, const float *v, const float *dhat, const float *nrm, float *gx, int64_t BF,
                                 int C, int F, int64_t P) {
  const bool keep = true;
  const bool vec = norm_vec_ok(P, x, v, dhat, nrm, gx);
#define TCL_NORM_KEEP(NCG) hipLaunchKernelGGL((corr_normalise_bwd_keep_kernel<2, NCG, 16>), dim3((unsigned)cdiv(P, 128), (unsigned)BF), dim3(64 * NCG), 0, s, x, v, dhat, nrm, gx, C, F, (int)P)
  if (vec && keep && C <= 256) {   // x and d_hat read once; two pixels per lane so that 2 x 16 channels fit the register budget
    if (C <= 64) TCL_NORM_KEEP(4);
    else if (C <= 128) TCL_NORM_KEEP(8);
    else TCL_NORM_KEEP(16);
  } else
#undef TCL_NORM_KEEP
  if (!vec)
    hipLaunchKernelGGL((corr_normalise_bwd_kernel<1, 4>), dim3((unsigned)cdiv(P, 64), (unsigned)BF), dim3(256), 0, s, x, v, dhat, nrm, gx, C, F, (int)P);
  else if (cdiv(P, 256) * BF >= 4 * kNumCU || C < 128)
    hipLaunchKernelGGL((corr_normalise_bwd_kernel<4, 4>), dim3((unsigned)cdiv(P, 256), (unsigned)BF), dim3(256), 0, s, x, v, dhat, nrm, gx, C, F, (int)P);
  else
    hipLaunchKernelGGL((corr_normalise_bwd_kernel<4, 16>), dim3((unsigned)cdiv(P, 256), (unsigned)BF), dim3(1024), 0, s, x, v, dhat, nrm, gx, C, F, (int)P);
}

static int corr_check(const char *fn, int64_t B, int64_t C, int64_t F, int64_t P, int precision) {
  TCL_REQUIRE(B > 0 && C > 0 && F > 0 && P > 0, TCL_ERR_SHAPE, "%s: bad shape B=%ld C=%ld F=%ld P=%ld", fn, (long)B, (long)C, (long)F, (long)P);
  TCL_REQUIRE(B * F < 65536 && P < (1 << 24) && C < (1 << 20), TCL_ERR_SHAPE, "%s: shape out of range", fn);
  TCL_REQUIRE(precision == TCL_PREC_F32 || precision == TCL_PREC_BF16 || precision == TCL_PREC_BF16X3, TCL_ERR_UNSUPPORTED,
              "%s: unknown precision %d", fn, precision);
  return TCL_OK;
}

extern "C" int tcl_corr4d_fwd(const float *x_t, const float *v_t, const float *x_r, const float *v_r, float *out, int64_t B,
                              int64_t C, int64_t F, int64_t P, int precision, void *workspace, size_t workspace_bytes,
                              void *stream) {
  int rc = corr_check("tcl_corr4d_fwd", B, C, F, P, precision);
  if (rc) return rc;
  TCL_REQUIRE(x_t && x_r && out, TCL_ERR_ARG, "tcl_corr4d_fwd: NULL pointer");
  const size_t need = tcl_corr4d_fwd_workspace_bytes(B, C, F, P, precision);
  TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
              "tcl_corr4d_fwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const CorrWs w = corr_ws(workspace, B, C, F, P, precision);
  launch_normalise(s, x_t, v_t, HatOut{w.a_hat, w.a_hi, w.a_lo}, w.n_t, B, (int)C, 1, P);
  launch_normalise(s, x_r, v_r, HatOut{w.r_hat, w.r_hi, w.r_lo}, w.n_r, B * F, (int)C, (int)F, P);
  rc = check_launch("tcl_corr4d_fwd(normalise)");
  if (rc) return rc;
  return tcl_corr4d_contract(workspace, out, B, C, F, P, precision, stream);
}

// the contraction's operand description: out[p][q] = sum_c a^[c][p] r^[c][q], one GEMM per (b, f)
static GemmArgs corr_fwd_gemm(const CorrWs &w, float *out, int64_t B, int64_t C, int64_t F, int64_t P) {
  GemmArgs g{};
  g.A = w.a_hat; g.B = w.r_hat; g.D = out;
  g.M = (int)P; g.N = (int)P; g.K = (int)C; g.nseg = 1;
  g.a_ld = P; g.b_ld = P; g.d_ld = P;
  g.a_bs = C * P; g.b_bs = C * P; g.d_bs = P * P;
  g.a_ss = 0; g.b_ss = 0; g.a_bdiv = (int)F; g.nbatch = (int)(B * F);
  return g;
}

extern "C" int tcl_corr4d_contract(const void *fwd_workspace, float *out, int64_t B, int64_t C, int64_t F, int64_t P, int precision,
                                   void *stream) {
  int rc = corr_check("tcl_corr4d_contract", B, C, F, P, precision);
  if (rc) return rc;
  TCL_REQUIRE(fwd_workspace && aligned(fwd_workspace, 256) && out, TCL_ERR_ARG, "tcl_corr4d_contract: NULL / misaligned pointer");
  const CorrWs w = corr_ws(const_cast<void *>(fwd_workspace), B, C, F, P, precision);
  const GemmArgs g = corr_fwd_gemm(w, out, B, C, F, P);
  if (precision != TCL_PREC_F32)   // the bf16 images the normalisation left behind the fp32 ones
    return launch_gemm_bf16op<TCL_BF16_TILE, EPI_STORE>(g, Bf16Ops{w.a_hi, w.a_lo, w.r_hi, w.r_lo}, precision == TCL_PREC_BF16X3 ? 2 : 1,
                                                         (hipStream_t)stream, "tcl_corr4d_contract(bf16 gemm)");
  return launch_gemm<128, 128, 2, 2, false, false>(g, (hipStream_t)stream, "tcl_corr4d_contract(gemm)");
}

constexpr int ARGMAX_BN = 128;
extern "C" size_t tcl_corr4d_argmax_flow_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision) {
  if (B <= 0 || C <= 0 || F <= 0 || P <= 0) return 0;
  return corr_ws(nullptr, B, C, F, P, precision).bytes + round_up((size_t)B * F * P * cdiv(P, ARGMAX_BN) * sizeof(float2), 256);
}

extern "C" int tcl_corr4d_argmax_flow(const float *x_t, const float *v_t, const float *x_r, const float *v_r, float *flow, int64_t B,
                                      int64_t C, int64_t F, int64_t S, int precision, void *workspace, size_t workspace_bytes, void *stream) {
  const int64_t P = S * S;
  int rc = corr_check("tcl_corr4d_argmax_flow", B, C, F, P, precision);
  if (rc) return rc;
  TCL_REQUIRE(S > 1, TCL_ERR_SHAPE, "tcl_corr4d_argmax_flow: S must be > 1");
  TCL_REQUIRE(x_t && x_r && flow && aligned(flow, 8), TCL_ERR_ARG, "tcl_corr4d_argmax_flow: NULL / misaligned pointer");
  const size_t need = tcl_corr4d_argmax_flow_workspace_bytes(B, C, F, P, precision);
  TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
              "tcl_corr4d_argmax_flow: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const CorrWs w = corr_ws(workspace, B, C, F, P, precision);
  float2 *part = (float2 *)((char *)workspace + w.bytes);
  const dim3 blk(256);
  launch_normalise(s, x_t, v_t, HatOut{w.a_hat, w.a_hi, w.a_lo}, w.n_t, B, (int)C, 1, P);
  launch_normalise(s, x_r, v_r, HatOut{w.r_hat, w.r_hi, w.r_lo}, w.n_r, B * F, (int)C, (int)F, P);
  rc = check_launch("tcl_corr4d_argmax_flow(normalise)");
  if (rc) return rc;
  GemmArgs g = corr_fwd_gemm(w, (float *)part, B, C, F, P);
  g.d_bs = 0;
  if (precision != TCL_PREC_F32)
    rc = launch_gemm_bf16op<TCL_BF16_TILE, EPI_ARGMAX>(g, Bf16Ops{w.a_hi, w.a_lo, w.r_hi, w.r_lo}, precision == TCL_PREC_BF16X3 ? 2 : 1, s,
                                                              "tcl_corr4d_argmax_flow(bf16 gemm)");
  else rc = launch_gemm<128, ARGMAX_BN, 2, 2, false, false, EPI_ARGMAX>(g, s, "tcl_corr4d_argmax_flow(gemm)");
  if (rc) return rc;
  const int64_t rows = B * F * P;
  int64_t blocks = cdiv(rows, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(argmax_tiles_to_flow_kernel, dim3((unsigned)blocks), blk, 0, s, part, flow, rows, (int)cdiv(P, ARGMAX_BN), (int)S,
                     (float)(2.0 / (double)(S - 1)));
  return check_launch("tcl_corr4d_argmax_flow(reduce)");
}

extern "C" size_t tcl_corr4d_bwd_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision) {
  (void)precision;
  if (B <= 0 || C <= 0 || F <= 0 || P <= 0) return 0;
  return round_up((size_t)B * C * P * 4, 256) + round_up((size_t)B * F * C * P * 4, 256);
}

extern "C" int tcl_corr4d_bwd(const float *grad_out, const float *x_t, const float *v_t, const float *x_r, const float *v_r,
                              const void *fwd_workspace, float *g_x_t, float *g_x_r, int64_t B, int64_t C, int64_t F,
                              int64_t P, int precision, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = corr_check("tcl_corr4d_bwd", B, C, F, P, precision);
  if (rc) return rc;
  TCL_REQUIRE(grad_out && x_t && x_r && fwd_workspace, TCL_ERR_ARG, "tcl_corr4d_bwd: NULL pointer");
  TCL_REQUIRE(aligned(fwd_workspace, 256), TCL_ERR_WORKSPACE, "tcl_corr4d_bwd: fwd_workspace must be 256-byte aligned");
  const size_t need = tcl_corr4d_bwd_workspace_bytes(B, C, F, P, precision);
  TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
              "tcl_corr4d_bwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const CorrWs w = corr_ws(const_cast<void *>(fwd_workspace), B, C, F, P, precision);
  float *d_a = (float *)workspace;
  float *d_r = (float *)((char *)workspace + round_up((size_t)B * C * P * 4, 256));
  const int nsplit = precision == TCL_PREC_BF16X3 ? 2 : 1;
  if (g_x_t) {
    // d_a^[b][c][p] = sum_f sum_q r^[b,f][c][q] G[b,f][p][q]      (M = C, N = P, K = P per reference frame)
    GemmArgs g{};
    g.A = w.r_hat; g.B = grad_out; g.D = d_a;
    g.M = (int)C; g.N = (int)P; g.K = (int)P; g.nseg = (int)F;
    g.a_ld = P; g.b_ld = P; g.d_ld = P;
    g.a_bs = F * C * P; g.b_bs = F * P * P; g.d_bs = C * P;
    g.a_ss = C * P; g.b_ss = P * P; g.a_bdiv = 1; g.nbatch = (int)B;
    if (precision != TCL_PREC_F32) rc = launch_gemm_bf16<256, 128, 4, 2, true, true>(g, Bf16Ops{w.r_hi, w.r_lo, nullptr, nullptr}, nsplit, s, "tcl_corr4d_bwd(d_a bf16 gemm)");
    else if (gemm_wide_ok(g, true, true) && wide_kernel_ready<true, true>()) rc = launch_gemm_wide<true, true>(g, s, "tcl_corr4d_bwd(d_a gemm)");
    else rc = launch_gemm<256, 128, 4, 2, true, true>(g, s, "tcl_corr4d_bwd(d_a gemm)");
    if (rc) return rc;
    launch_normalise_bwd(s, x_t, v_t, d_a, w.n_t, g_x_t, B, (int)C, 1, P);
    rc = check_launch("tcl_corr4d_bwd(normalise bwd, target)");
    if (rc) return rc;
  }
  if (g_x_r) {
    // d_r^[b,f][c][q] = sum_p a^[b][c][p] G[b,f][p][q]              (M = C, N = P, K = P)
    GemmArgs g{};
    g.A = w.a_hat; g.B = grad_out; g.D = d_r;
    g.M = (int)C; g.N = (int)P; g.K = (int)P; g.nseg = 1;
    g.a_ld = P; g.b_ld = P; g.d_ld = P;
    g.a_bs = C * P; g.b_bs = P * P; g.d_bs = C * P;
    g.a_ss = 0; g.b_ss = 0; g.a_bdiv = (int)F; g.nbatch = (int)(B * F);
    if (precision != TCL_PREC_F32) rc = launch_gemm_bf16<256, 128, 4, 2, true, false>(g, Bf16Ops{w.a_hi, w.a_lo, nullptr, nullptr}, nsplit, s, "tcl_corr4d_bwd(d_r bf16 gemm)");
    else if (gemm_wide_ok(g, true, false) && wide_kernel_ready<true, false>()) rc = launch_gemm_wide<true, false>(g, s, "tcl_corr4d_bwd(d_r gemm)");
    else rc = launch_gemm<256, 128, 4, 2, true, false>(g, s, "tcl_corr4d_bwd(d_r gemm)");
    if (rc) return rc;
    launch_normalise_bwd(s, x_r, v_r, d_r, w.n_r, g_x_r, B * F, (int)C, (int)F, P);
    rc = check_launch("tcl_corr4d_bwd(normalise bwd, refs)");
  }
  return rc;
}

extern "C" int tcl_corr1x1_fwd(const float *f1, const float *f2, const float *m1, const float *m2, float *out, int64_t B,
                               int64_t C, int64_t P, void *stream) {
  TCL_REQUIRE(B > 0 && C > 0 && P > 0 && P < (1LL << 31) && C < (1 << 24), TCL_ERR_SHAPE, "tcl_corr1x1_fwd: bad shape");
  TCL_REQUIRE(f1 && f2 && m1 && m2 && out, TCL_ERR_ARG, "tcl_corr1x1_fwd: NULL pointer");
  const int64_t blocks = cdiv(B * P, 64);
  TCL_REQUIRE(blocks < (1LL << 31), TCL_ERR_SHAPE, "tcl_corr1x1_fwd: too many pixels for one launch");
  hipLaunchKernelGGL(corr1x1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f1, f2, m1, m2, out, (int)C, (int)P, B * P);
  return check_launch("tcl_corr1x1_fwd");
}

extern "C" int tcl_corr1x1_bwd(const float *grad_out, const float *f1, const float *f2, const float *m1, const float *m2,
                               float *g_f1, float *g_f2, int64_t B, int64_t C, int64_t P, void *stream) {
  TCL_REQUIRE(B > 0 && C > 0 && P > 0 && P < (1LL << 31) && C < (1 << 24), TCL_ERR_SHAPE, "tcl_corr1x1_bwd: bad shape");
  TCL_REQUIRE(grad_out && f1 && f2 && m1 && m2, TCL_ERR_ARG, "tcl_corr1x1_bwd: NULL pointer");
  if (!g_f1 && !g_f2) return TCL_OK;
  const int64_t blocks = cdiv(B * P, 64);
  TCL_REQUIRE(blocks < (1LL << 31), TCL_ERR_SHAPE, "tcl_corr1x1_bwd: too many pixels for one launch");
  hipLaunchKernelGGL(corr1x1_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_out, f1, f2, m1, m2, g_f1, g_f2, (int)C, (int)P, B * P);
  return check_launch("tcl_corr1x1_bwd");
}

extern "C" int tcl_corr_to_flow(const float *corr, float *flow, int64_t N, int64_t S, void *stream) {
  TCL_REQUIRE(N > 0 && S > 1 && S < 32768, TCL_ERR_SHAPE, "tcl_corr_to_flow: bad shape N=%ld S=%ld", (long)N, (long)S);
  TCL_REQUIRE(corr && flow && aligned(flow, 8), TCL_ERR_ARG, "tcl_corr_to_flow: NULL / misaligned pointer");
  const int64_t rows = N * S * S;
  int64_t blocks = cdiv(rows, 4);
  if (blocks > kMaxStreamBlocks * 4) blocks = kMaxStreamBlocks * 4;
  const float scale = (float)(2.0 / (double)(S - 1));
  hipLaunchKernelGGL(corr_to_flow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, corr, flow, rows, (int)S, scale);
  return check_launch("tcl_corr_to_flow");
}
