// What does the warp's ACCESS SHAPE cost without the warp?  320 images of 240x424, 4 planes in -> 4 planes out, (8, chunks, images/8) grid
// like warp_block().  Variants: dword per lane (the warp's stores), the warp's 8 unaligned 8-byte tap loads per pixel at identity
// positions (north + south row of 4 planes) feeding dword stores, and 4 consecutive pixels per lane (16-byte loads and stores).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) F2U { float a, b; };
constexpr int H = 240, W = 424, HW = H * W, NIMG = 320, NP = 4;

template <int PX>   // PX pixels per lane, 256 apart
__global__ void __launch_bounds__(256) dword_copy(const float *in, float *out) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  float v[PX][NP];
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const unsigned pix = min((blockIdx.y * PX + p) * 256 + threadIdx.x, (unsigned)HW - 1);
#pragma unroll
    for (int c = 0; c < NP; ++c) v[p][c] = ip[c * HW + pix];
  }
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const unsigned pix0 = (blockIdx.y * PX + p) * 256 + threadIdx.x;
    if (pix0 < HW)
#pragma unroll
      for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(v[p][c], op + c * HW + pix0);
  }
}
template <int PX>
__global__ void __launch_bounds__(256) tap_copy(const float *in, float *out) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  F2U a[PX][NP], b[PX][NP];
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const unsigned pix = min((blockIdx.y * PX + p) * 256 + threadIdx.x, (unsigned)HW - W - 2);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      a[p][c] = *reinterpret_cast<const F2U *>(reinterpret_cast<const char *>(ip + c * HW) + (pix << 2));
      b[p][c] = *reinterpret_cast<const F2U *>(reinterpret_cast<const char *>(ip + c * HW) + ((pix + W) << 2));
    }
  }
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    const unsigned pix0 = (blockIdx.y * PX + p) * 256 + threadIdx.x;
    if (pix0 < HW)
#pragma unroll
      for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(a[p][c].a * 0.25f + a[p][c].b * 0.25f + b[p][c].a * 0.25f + b[p][c].b * 0.25f, op + c * HW + pix0);
  }
}
__global__ void __launch_bounds__(256) f4_copy(const float *in, float *out) {   // 4 consecutive pixels per lane
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const unsigned q = blockIdx.y * 256 + threadIdx.x;
  if (q * 4 >= HW) return;
  f4 v[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) v[c] = reinterpret_cast<const f4 *>(ip + c * HW)[q];
#pragma unroll
  for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(v[c], reinterpret_cast<f4 *>(op + c * HW) + q);
}

// LDS-staged taps: a block owns a 64 x TH output tile, stages the (64 + 3) x (TH + 3) source window of the 4 planes with coalesced
// dword loads (flattened window, all loads issued before the LDS writes), then every lane takes its 2 x 2 taps from LDS.
template <int TH, bool DMA>
__global__ void __launch_bounds__(256) lds_tap_copy(const float *in, float *out, int tx) {
  constexpr int BW = 67, BH = TH + 3, NE = BW * BH, K = (NE + 255) / 256;
  __shared__ float sm[NP][K * 256];
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const int ty = blockIdx.y / tx, txi = blockIdx.y - ty * tx;
  const int x0 = txi * 64, y0 = ty * TH;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const int bx0 = max(x0 - 1, 0), by0 = max(y0 - 1, 0);
  if (DMA) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned e = k * 256 + threadIdx.x;
      const unsigned r = e / BW, c = e - r * BW;
      const unsigned off = min(by0 + r, (unsigned)H - 1) * W + min(bx0 + c, (unsigned)W - 1);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        __builtin_amdgcn_global_load_lds(ip + pl * HW + off, (__attribute__((address_space(3))) void *)&sm[pl][k * 256 + (threadIdx.x & ~63)], 4, 0, 0);
    }
  } else {
    float v[K][NP];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned e = k * 256 + threadIdx.x;
      const unsigned r = e / BW, c = e - r * BW;
      const unsigned off = min(by0 + r, (unsigned)H - 1) * W + min(bx0 + c, (unsigned)W - 1);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) v[k][pl] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ip + pl * HW) + (off << 2));
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) sm[pl][k * 256 + threadIdx.x] = v[k][pl];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = x0 + lane;
#pragma unroll 2
  for (int r = wv; r < TH; r += 4) {
    const int y = y0 + r;
    const int lx = min(x, W - 2) - bx0, ly = min(y, H - 2) - by0;
    float o[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const float *q = &sm[pl][ly * BW + lx];
      o[pl] = q[0] * 0.25f + q[1] * 0.25f + q[BW] * 0.25f + q[BW + 1] * 0.25f;
    }
    if (x < W && y < H)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) __builtin_nontemporal_store(o[pl], op + pl * HW + y * W + x);
  }
}

// 64 x TH tiles without LDS: every wave instruction moves one 64-pixel row segment (32-byte aligned only: W = 424)
template <int TH>
__global__ void __launch_bounds__(256) tile_copy(const float *in, float *out, int tx) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const int ty = blockIdx.y / tx, txi = blockIdx.y - ty * tx;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x = min(txi * 64 + lane, W - 1);
  float v[TH / 4][NP];
#pragma unroll
  for (int i = 0; i < TH / 4; ++i) {
    const int y = min(ty * TH + wv + 4 * i, H - 1);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) v[i][pl] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ip + pl * HW) + ((unsigned)(y * W + x) << 2));
  }
#pragma unroll
  for (int i = 0; i < TH / 4; ++i) {
    const int y = ty * TH + wv + 4 * i;
    if (txi * 64 + lane < W && y < H)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) __builtin_nontemporal_store(v[i][pl], op + pl * HW + y * W + x);
  }
}
// flattened chunks, aligned dword loads of the pixel's row AND the row below (the taps' L2 footprint without their misalignment)
__global__ void __launch_bounds__(256) tworow_copy(const float *in, float *out) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const unsigned pix0 = blockIdx.y * 256 + threadIdx.x, pix = min(pix0, (unsigned)HW - W - 1);
  float a[NP], b[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) { a[c] = ip[c * HW + pix]; b[c] = ip[c * HW + pix + W]; }
  if (pix0 < HW)
#pragma unroll
    for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(a[c] * 0.5f + b[c] * 0.5f, op + c * HW + pix0);
}

// what exactly do the tap loads pay for?  MODE 0: 16 dword loads (west and east separately), 1: 8 dwordx2 at 8-byte ALIGNED addresses,
// 2: 8 aligned dwordx4 (the aligned 16 bytes holding the pair when xa % 4 != 3), 3: only the 4 north-row dwordx2 (unaligned)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) tap_mode_copy(const float *in, float *out) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const unsigned pix0 = blockIdx.y * 256 + threadIdx.x, pix = min(pix0, (unsigned)HW - W - 4);
  float r[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    const char *pb = reinterpret_cast<const char *>(ip + c * HW);
    if (MODE == 0) {
      const float a = *reinterpret_cast<const float *>(pb + (pix << 2)), b = *reinterpret_cast<const float *>(pb + ((pix + 1) << 2));
      const float d = *reinterpret_cast<const float *>(pb + ((pix + W) << 2)), e = *reinterpret_cast<const float *>(pb + ((pix + W + 1) << 2));
      r[c] = a * 0.25f + b * 0.25f + d * 0.25f + e * 0.25f;
    } else if (MODE == 1) {
      const f2 a = *reinterpret_cast<const f2 *>(pb + ((pix & ~1u) << 2)), d = *reinterpret_cast<const f2 *>(pb + (((pix + W) & ~1u) << 2));
      r[c] = a.x * 0.25f + a.y * 0.25f + d.x * 0.25f + d.y * 0.25f;
    } else if (MODE == 2) {
      const f4 a = *reinterpret_cast<const f4 *>(pb + ((pix & ~3u) << 2)), d = *reinterpret_cast<const f4 *>(pb + (((pix + W) & ~3u) << 2));
      r[c] = a.x * 0.25f + a.y * 0.25f + d.z * 0.25f + d.w * 0.25f + a.z + a.w + d.x + d.y;
    } else if (MODE == 3) {
      const F2U a = *reinterpret_cast<const F2U *>(pb + (pix << 2));
      r[c] = a.a * 0.5f + a.b * 0.5f;
    } else if (MODE == 4 || MODE == 5) {   // west taps by every lane, east taps from lane + 1 (DPP) except lane 63 (4) / every 16th lane (5), which load theirs
      const float a = *reinterpret_cast<const float *>(pb + (pix << 2)), d = *reinterpret_cast<const float *>(pb + ((pix + W) << 2));
      float b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xf, 0xf, false));
      float e = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x130, 0xf, 0xf, false));
      const bool own = MODE == 4 ? (threadIdx.x & 63) == 63 : (threadIdx.x & 15) == 15;
      if (own) { b = *reinterpret_cast<const float *>(pb + ((pix + 1) << 2)); e = *reinterpret_cast<const float *>(pb + ((pix + W + 1) << 2)); }
      r[c] = a * 0.25f + b * 0.25f + d * 0.25f + e * 0.25f;
    } else {   // 6: 63 pixels per wave, lane 63 only provides taps
      const unsigned wv = (blockIdx.y * 4 + (threadIdx.x >> 6)), q = min(wv * 63 + (threadIdx.x & 63), (unsigned)HW - W - 4);
      const float a = *reinterpret_cast<const float *>(pb + (q << 2)), d = *reinterpret_cast<const float *>(pb + ((q + W) << 2));
      const float b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x130, 0xf, 0xf, false));
      const float e = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x130, 0xf, 0xf, false));
      r[c] = a * 0.25f + b * 0.25f + d * 0.25f + e * 0.25f;
    }
  }
  if (MODE == 6) {
    const unsigned wv = (blockIdx.y * 4 + (threadIdx.x >> 6)), q = wv * 63 + (threadIdx.x & 63);
    if ((threadIdx.x & 63) < 63 && q < HW)
#pragma unroll
      for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(r[c], op + c * HW + q);
    return;
  }
  if (pix0 < HW)
#pragma unroll
    for (int c = 0; c < NP; ++c) __builtin_nontemporal_store(r[c], op + c * HW + pix0);
}

// skewed tiles: the 64-pixel row segment of tile column k in row y starts at 64 k - ((y W) mod 32), so that EVERY segment starts on a
// 128-byte line (rows of W = 424 floats start 32 bytes further into a line each); one extra, mostly empty column at the right edge
template <int TH>
__global__ void __launch_bounds__(256) skew_copy(const float *in, float *out, int tx) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const int ty = blockIdx.y / tx, txi = blockIdx.y - ty * tx;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float v[TH / 4][NP];
#pragma unroll
  for (int i = 0; i < TH / 4; ++i) {
    const int y = min(ty * TH + wv + 4 * i, H - 1);
    const int x = txi * 64 - ((y * W) & 31) + lane;
    const unsigned off = (unsigned)(y * W + min(max(x, 0), W - 1));
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) v[i][pl] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ip + pl * HW) + (off << 2));
  }
#pragma unroll
  for (int i = 0; i < TH / 4; ++i) {
    const int y = ty * TH + wv + 4 * i;
    const int x = txi * 64 - ((y * W) & 31) + lane;
    if (x >= 0 && x < W && y < H)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) __builtin_nontemporal_store(v[i][pl], op + pl * HW + y * W + x);
  }
}
// skewed 64 x TH output tiles + LDS-staged source window: rows by0 .. by0 + TH + 2, each row's window = the 128 floats (4 lines) from
// the line holding the segment's start minus one line... here: [seg_start - 32, seg_start + 96), loaded as aligned 16-byte pieces
template <int TH>
__global__ void __launch_bounds__(256) skew_lds_taps(const float *in, float *out, int tx) {
  constexpr int BH = TH + 2, RW = 128;                 // window rows, floats per window row
  __shared__ __attribute__((aligned(16))) float sm[NP][BH * RW];
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const int ty = blockIdx.y / tx, txi = blockIdx.y - ty * tx;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const int y0 = ty * TH;
  // stage: BH rows x 32 float4 pieces per plane; piece index e -> (row, q)
  constexpr int NE = BH * 32, K = (NE + 255) / 256;
  f4 v[K][NP];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * 256 + threadIdx.x, r = e >> 5, q = e & 31;
    const int y = min(y0 + r, H - 1);
    const int xs = txi * 64 - ((min(y0 + max(r - 1, 0), H - 1) * W) & 31) - 32;   // window start of this source row (float index in the row), line-aligned for row y0 + r - 1 ... timing only
    const int off = y * W + min(max(xs + 4 * q, 0), W - 4);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) v[k][pl] = *reinterpret_cast<const f4 *>(reinterpret_cast<const char *>(ip + pl * HW) + ((unsigned)(off & ~3) << 2));
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = k * 256 + threadIdx.x;
    if (e < NE)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<f4 *>(&sm[pl][e * 4]) = v[k][pl];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll 2
  for (int r = wv; r < TH; r += 4) {
    const int y = y0 + r;
    const int x = txi * 64 - ((y * W) & 31) + lane;
    float o[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const float *q = &sm[pl][r * RW + 32 + lane];
      o[pl] = q[0] * 0.25f + q[1] * 0.25f + q[RW] * 0.25f + q[RW + 1] * 0.25f;
    }
    if (x >= 0 && x < W && y < H)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) __builtin_nontemporal_store(o[pl], op + pl * HW + y * W + x);
  }
}

// 4 consecutive pixels per lane: the same 8 tap loads per pixel (lanes now 16 bytes apart), 16-byte stores
__global__ void __launch_bounds__(256) tap_x4store_copy(const float *in, float *out) {
  const int n = blockIdx.z * 8 + blockIdx.x;
  if (n >= NIMG) return;
  const float *ip = in + (size_t)n * NP * HW; float *op = out + (size_t)n * NP * HW;
  const unsigned q0 = blockIdx.y * 256 + threadIdx.x, q = min(q0 * 4, (unsigned)HW - W - 8);
  F2U a[4][NP], b[4][NP];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      a[j][c] = *reinterpret_cast<const F2U *>(reinterpret_cast<const char *>(ip + c * HW) + ((q + j) << 2));
      b[j][c] = *reinterpret_cast<const F2U *>(reinterpret_cast<const char *>(ip + c * HW) + ((q + j + W) << 2));
    }
  if (q0 * 4 < HW)
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      f4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = a[j][c].a * 0.25f + a[j][c].b * 0.25f + b[j][c].a * 0.25f + b[j][c].b * 0.25f;
      __builtin_nontemporal_store(r, reinterpret_cast<f4 *>(op + c * HW) + q0);
    }
}
template <typename F> float timeit(F f, int it = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < it; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}
int main() {
  const size_t bytes = (size_t)NIMG * NP * HW * 4;
  const int NS = 5;   // rotating sets: cold
  float *in[NS], *out[NS];
  for (int i = 0; i < NS; ++i) { CK(hipMalloc(&in[i], bytes + 4096)); CK(hipMalloc(&out[i], bytes)); CK(hipMemset(in[i], 1, bytes + 4096)); CK(hipMemset(out[i], 0, bytes)); }
  int k = 0;
  auto rep = [&](const char *name, auto launch) {
    const float t = timeit([&] { launch(in[k % NS], out[k % NS]); ++k; });
    printf("%-34s %7.1f us %6.2f TB/s\n", name, t * 1e3, 2.0 * bytes / t / 1e9);
  };
  const int c1 = (HW + 255) / 256, c2 = (HW + 511) / 512, c4 = (HW + 1023) / 1024;
  rep("dword copy, 1 px/lane", [&](const float *i, float *o) { dword_copy<1><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("dword copy, 2 px/lane", [&](const float *i, float *o) { dword_copy<2><<<dim3(8, c2, NIMG / 8), 256>>>(i, o); });
  rep("dword copy, 4 px/lane", [&](const float *i, float *o) { dword_copy<4><<<dim3(8, c4, NIMG / 8), 256>>>(i, o); });
  rep("8 tap loads -> dword, 1 px/lane", [&](const float *i, float *o) { tap_copy<1><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 tap loads -> dword, 2 px/lane", [&](const float *i, float *o) { tap_copy<2><<<dim3(8, c2, NIMG / 8), 256>>>(i, o); });
  rep("float4 copy, 4 consecutive px/lane", [&](const float *i, float *o) { f4_copy<<<dim3(8, (HW / 4 + 255) / 256, NIMG / 8), 256>>>(i, o); });
  const int tx = (W + 63) / 64;
  rep("16 dword tap loads", [&](const float *i, float *o) { tap_mode_copy<0><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 ALIGNED dwordx2 loads", [&](const float *i, float *o) { tap_mode_copy<1><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 aligned dwordx4 loads", [&](const float *i, float *o) { tap_mode_copy<2><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("4 unaligned dwordx2 (north only)", [&](const float *i, float *o) { tap_mode_copy<3><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 dword + DPP, lane 63 loads own east", [&](const float *i, float *o) { tap_mode_copy<4><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 dword + DPP, every 16th lane loads", [&](const float *i, float *o) { tap_mode_copy<5><<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  rep("8 dword + DPP, 63 px per wave", [&](const float *i, float *o) { tap_mode_copy<6><<<dim3(8, (HW + 251) / 252, NIMG / 8), 256>>>(i, o); });
  rep("8 tap loads, 4 consecutive px/lane, x4 stores", [&](const float *i, float *o) { tap_x4store_copy<<<dim3(8, (HW / 4 + 255) / 256, NIMG / 8), 256>>>(i, o); });
  rep("tile copy 64x32 (no LDS)", [&](const float *i, float *o) { tile_copy<32><<<dim3(8, tx * ((H + 31) / 32), NIMG / 8), 256>>>(i, o, tx); });
  rep("tile copy 64x16 (no LDS)", [&](const float *i, float *o) { tile_copy<16><<<dim3(8, tx * ((H + 15) / 16), NIMG / 8), 256>>>(i, o, tx); });
  rep("two aligned rows -> dword", [&](const float *i, float *o) { tworow_copy<<<dim3(8, c1, NIMG / 8), 256>>>(i, o); });
  const int txs = (W + 31 + 63) / 64;
  rep("skewed tile copy 64x32 (aligned)", [&](const float *i, float *o) { skew_copy<32><<<dim3(8, txs * ((H + 31) / 32), NIMG / 8), 256>>>(i, o, txs); });
  rep("skewed tile copy 64x16 (aligned)", [&](const float *i, float *o) { skew_copy<16><<<dim3(8, txs * ((H + 15) / 16), NIMG / 8), 256>>>(i, o, txs); });
  rep("skewed LDS taps 64x32, x4 window", [&](const float *i, float *o) { skew_lds_taps<32><<<dim3(8, txs * ((H + 31) / 32), NIMG / 8), 256>>>(i, o, txs); });
  rep("skewed LDS taps 64x16, x4 window", [&](const float *i, float *o) { skew_lds_taps<16><<<dim3(8, txs * ((H + 15) / 16), NIMG / 8), 256>>>(i, o, txs); });
  rep("LDS-staged taps, 64x32 tile", [&](const float *i, float *o) { lds_tap_copy<32, false><<<dim3(8, tx * ((H + 31) / 32), NIMG / 8), 256>>>(i, o, tx); });
  rep("LDS-staged taps, 64x16 tile", [&](const float *i, float *o) { lds_tap_copy<16, false><<<dim3(8, tx * ((H + 15) / 16), NIMG / 8), 256>>>(i, o, tx); });
  rep("LDS-DMA staged taps, 64x32 tile", [&](const float *i, float *o) { lds_tap_copy<32, true><<<dim3(8, tx * ((H + 31) / 32), NIMG / 8), 256>>>(i, o, tx); });
  rep("LDS-DMA staged taps, 64x16 tile", [&](const float *i, float *o) { lds_tap_copy<16, true><<<dim3(8, tx * ((H + 15) / 16), NIMG / 8), 256>>>(i, o, tx); });
  return 0;
}
