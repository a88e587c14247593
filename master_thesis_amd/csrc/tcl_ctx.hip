// Masked cross-frame context matching and aggregation (CPN "copy" step), forward and backward, gfx950.
//
// Replaces CPNContextMatchingDefault.forward (models/cpn.py:94-135), CPNContextMatchingComplete.forward
// (models/cpn.py:142-195), CM_Module.forward (models/cpn_original.py:201-243), masked_softmax
// (models/cpn.py:7-15), add_match_border / add_border (models/cpn.py:23-44) and the 3x3-cross dilation of
// utils/transforms.py:87-105; see include/tcl.h.
//
// The reference builds a (B,C,R,h,w) "cos_sim" tensor that is constant over C, h and w and runs a 7-kernel
// masked softmax over it inside a Python loop over the R reference frames.  Here the path is:
//   resize  : v_t, v_aligned -> (bilinear > 0.5) masks at feature resolution              [bit-exact]
//   sim     : gs[b,r] = sum_{c,x} vmap_r[x] f_t[c,x] f_r[c,x]     one pass over c_feats, wave shuffles,
//             per-block partial sums (summed in block order: deterministic, no float atomics)
//   finalize: v_sum, the `zeros` flag and the normalisation                                [flags bit-exact]
//   aggregate: per pixel softmax over r (channel independent), weighted sum of the reference features,
//             c_mask and the concatenated decoder input, written once
// All of it is HBM-bound (0.5 flop/byte): coalesced along the pixel axis of the NCHW-like (B,C,R+1,h,w) layout.
#include "tcl_common.h"
#include <mutex>

namespace tcl {

// ------------------------------------------------------------------------------------------ mask helpers
// out[n][y][x] = bilinear(v[n], align_corners=False)(y,x) > 0.5
__global__ void __launch_bounds__(256) resize_mask_kernel(const float *__restrict__ v, float *__restrict__ out, int64_t N,
                                                          int H, int W, int h, int w) {
  const float sh = __fdiv_rn((float)H, (float)h), sw = __fdiv_rn((float)W, (float)w);
  const int64_t total = N * h * w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / (h * w);
    const int r = (int)(i - n * h * w), y = r / w, x = r - y * w;
    const LinTap ty = linear_tap(y, H, sh), tx = linear_tap(x, W, sw);
    const float *p = v + n * H * W;
    const float val = bilerp(p[ty.i0 * W + tx.i0], p[ty.i0 * W + tx.i1], p[ty.i1 * W + tx.i0], p[ty.i1 * W + tx.i1], tx, ty);
    out[i] = val > 0.5f ? 1.f : 0.f;
  }
}

__device__ __forceinline__ float dilate_at(const float *img, int y, int x, int H, int W) {
  float m = img[y * W + x];
  if (y > 0) m = fmaxf(m, img[(y - 1) * W + x]);
  if (y < H - 1) m = fmaxf(m, img[(y + 1) * W + x]);
  if (x > 0) m = fmaxf(m, img[y * W + x - 1]);
  if (x < W - 1) m = fmaxf(m, img[y * W + x + 1]);
  return m;
}

__global__ void __launch_bounds__(256) dilate_cross3_kernel(const float *__restrict__ img, float *__restrict__ out, int64_t N, int H, int W) {
  const int64_t total = N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / (H * W);
    const int r = (int)(i - n * H * W), y = r / W, x = r - y * W;
    out[i] = dilate_at(img + n * H * W, y, x, H, W);
  }
}

// PX consecutive floats per lane as one 4 / 8 / 16-byte access (the pointer is PX*4-byte aligned by construction)
template <int PX> __device__ __forceinline__ void ldpx(const float *p, float (&v)[PX]) {
  if (PX == 4) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[PX > 1 ? 1 : 0] = t.y; v[PX > 2 ? 2 : 0] = t.z; v[PX - 1] = t.w; }
  else if (PX == 2) { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[PX - 1] = t.y; }
  else v[0] = p[0];
}
template <int PX> __device__ __forceinline__ void stpx(float *p, const float (&v)[PX]) {
  if (PX == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[PX > 1 ? 1 : 0], v[PX > 2 ? 2 : 0], v[PX - 1]);
  else if (PX == 2) *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[PX - 1]);
  else p[0] = v[0];
}
template <int PX> __device__ __forceinline__ void stpx_stream(float *p, const float (&v)[PX]) {
  if (PX == 4) st_stream4(p, v[0], v[PX > 1 ? 1 : 0], v[PX > 2 ? 2 : 0], v[PX - 1]);
  else if (PX == 2) st_stream2(p, v[0], v[PX - 1]);
  else st_stream(p, v[0]);
}

// ref_importance (B,3,R+1,H,W): planes = upsampled c_match with a middle plane inserted at R/2, hole border painted red.
// 2-D grid (chunk of 256*PX pixels, b * (R+1) + j); PX consecutive pixels of one row per lane (PX = 4 when W % 4 == 0:
// 16-byte loads and stores, 4x fewer workgroups: the one-pixel version was bound by workgroup dispatch, not by HBM).
// Straight-line: every load is unconditional (clamped neighbours / clamped plane index), selected afterwards.
template <int PX>
__global__ void __launch_bounds__(256) ref_importance_kernel(const float *__restrict__ c_match, const float *__restrict__ v_t,
                                                             float *__restrict__ out, int B, int R, int h, int w, int H, int W,
                                                             int middle_is_vt, float sh, float sw) {
  const int HW = H * W;
  const int pix = (blockIdx.x * 256 + threadIdx.x) * PX;
  if (pix >= HW) return;
  const int bj = blockIdx.y, b = bj / (R + 1), j = bj - b * (R + 1);
  const int y = pix / W, x = pix - y * W;   // PX | W: the PX pixels share the row
  const float *vt = v_t + (int64_t)b * HW;
  const bool middle = j == R / 2;
  const int r = min(j < R / 2 ? j : j - 1, R - 1);
  const float *p = c_match + ((int64_t)b * R + r) * h * w;
  const LinTap ty = linear_tap(y, h, sh);
  // loads
  float vc[PX], vn[PX], vs[PX], p00[PX], p01[PX], p10[PX], p11[PX];
  LinTap tx[PX];
  ldpx<PX>(vt + pix, vc);
  ldpx<PX>(vt + (y > 0 ? pix - W : pix), vn);
  ldpx<PX>(vt + (y < H - 1 ? pix + W : pix), vs);
  const float vw = vt[x > 0 ? pix - 1 : pix], ve = vt[x + PX < W ? pix + PX : pix + PX - 1];
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    tx[e] = linear_tap(x + e, w, sw);
    p00[e] = p[ty.i0 * w + tx[e].i0]; p01[e] = p[ty.i0 * w + tx[e].i1];
    p10[e] = p[ty.i1 * w + tx[e].i0]; p11[e] = p[ty.i1 * w + tx[e].i1];
  }
  float red[PX], keep[PX];
#pragma unroll
  for (int e = 0; e < PX; ++e) {
    // border = dilate(hole) - hole > 0 with hole = 1 - v_t; dilation of (1 - v) over the 3x3 cross
    const float west = e == 0 ? vw : vc[e > 0 ? e - 1 : 0], east = e == PX - 1 ? ve : vc[e < PX - 1 ? e + 1 : 0];
    const float hole = 1.f - vc[e];
    const float dil = fmaxf(fmaxf(fmaxf(hole, 1.f - vn[e]), fmaxf(1.f - vs[e], 1.f - west)), 1.f - east);
    const float border = (dil - hole) > 0.f ? 1.f : 0.f;
    const float up = bilerp(p00[e], p01[e], p10[e], p11[e], tx[e], ty);
    const float val = middle ? (middle_is_vt ? vc[e] : 0.f) : up;
    keep[e] = val * (1.f - border);
    red[e] = keep[e] + border;
  }
  float *o = out + ((int64_t)b * 3 * (R + 1) + j) * HW + pix;
  stpx_stream<PX>(o, red);   // 430 MB of write-only output at C4
  stpx_stream<PX>(o + (int64_t)(R + 1) * HW, keep);
  stpx_stream<PX>(o + (int64_t)2 * (R + 1) * HW, keep);
}

// ------------------------------------------------------------------------------------------ A5 generic masked softmax
// vec (O,R,I), mask (Om,R,I); thread per (o,i), sequential over r.
__global__ void __launch_bounds__(256) masked_softmax_fwd_kernel(const float *__restrict__ vec, const float *__restrict__ mask,
                                                                 float *__restrict__ out, int64_t O, int64_t Om, int R, int64_t I) {
  const int64_t total = O * I, rep = O / Om;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / I, i = t - o * I;
    const float *vp = vec + o * R * I + i, *mp = mask + (o / rep) * R * I + i;
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, vp[r * I] * mp[r * I]);
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += expf(vp[r * I] * mp[r * I] - mx) * mp[r * I];
    if (s < 1e-4f) s += 1.f;
    float *op = out + o * R * I + i;
    for (int r = 0; r < R; ++r) op[r * I] = expf(vp[r * I] * mp[r * I] - mx) * mp[r * I] / s;
  }
}

// R <= 16 (the reference's use: R = reference frames): one (o,i) per lane, all 2R (3R) loads of the lane in flight
// before the first use, values kept in registers, outputs written once as streaming stores.  Same arithmetic, same
// operation order as the generic kernels above / below.
template <int R>
__global__ void __launch_bounds__(256) masked_softmax_fwd_small_kernel(const float *__restrict__ vec, const float *__restrict__ mask,
                                                                       float *__restrict__ out, int64_t O, int64_t Om, int64_t I) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= O * I) return;
  const int64_t o = t / I, i = t - o * I;
  const float *vp = vec + o * R * I + i, *mp = mask + (o / (O / Om)) * R * I + i;
  float v[R], m[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { v[r] = vp[r * I]; m[r] = mp[r * I]; }
  TCL_LOADS_FIRST(2 * R);
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < R; ++r) mx = fmaxf(mx, v[r] * m[r]);
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) { v[r] = expf(v[r] * m[r] - mx) * m[r]; s += v[r]; }
  if (s < 1e-4f) s += 1.f;
  float *op = out + o * R * I + i;
#pragma unroll
  for (int r = 0; r < R; ++r) st_stream(op + r * I, v[r] / s);
}

template <int R>
__global__ void __launch_bounds__(256) masked_softmax_bwd_small_kernel(const float *__restrict__ go, const float *__restrict__ vec,
                                                                       const float *__restrict__ mask, float *__restrict__ gv,
                                                                       int64_t O, int64_t Om, int64_t I) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= O * I) return;
  const int64_t o = t / I, i = t - o * I;
  const float *vp = vec + o * R * I + i, *mp = mask + (o / (O / Om)) * R * I + i, *gp = go + o * R * I + i;
  float v[R], m[R], g[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { v[r] = vp[r * I]; m[r] = mp[r * I]; g[r] = gp[r * I]; }
  TCL_LOADS_FIRST(3 * R);
  float mx = -INFINITY;
  int am = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float z = v[r] * m[r];
    if (z > mx) { mx = z; am = r; }
  }
  float s = 0.f, ge = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    v[r] = expf(v[r] * m[r] - mx) * m[r];   // e_r
    s += v[r];
    ge += g[r] * v[r];
  }
  const float sp = s < 1e-4f ? s + 1.f : s;
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    g[r] = (g[r] / sp - ge / (sp * sp)) * v[r];   // d_r before the max term
    tot += g[r];
  }
  float *op = gv + o * R * I + i;
#pragma unroll
  for (int r = 0; r < R; ++r) st_stream(op + r * I, (r == am ? g[r] - tot : g[r]) * m[r]);
}

__global__ void __launch_bounds__(256) masked_softmax_bwd_kernel(const float *__restrict__ go, const float *__restrict__ vec,
                                                                 const float *__restrict__ mask, float *__restrict__ gv,
                                                                 int64_t O, int64_t Om, int R, int64_t I) {
  const int64_t total = O * I, rep = O / Om;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / I, i = t - o * I;
    const float *vp = vec + o * R * I + i, *mp = mask + (o / rep) * R * I + i, *gp = go + o * R * I + i;
    float mx = -INFINITY;
    int am = 0;
    for (int r = 0; r < R; ++r) {
      const float z = vp[r * I] * mp[r * I];
      if (z > mx) { mx = z; am = r; }
    }
    float s = 0.f, ge = 0.f;
    for (int r = 0; r < R; ++r) {
      const float e = expf(vp[r * I] * mp[r * I] - mx) * mp[r * I];
      s += e;
      ge += gp[r * I] * e;
    }
    const float sp = s < 1e-4f ? s + 1.f : s;
    float tot = 0.f;
    for (int r = 0; r < R; ++r) {
      const float e = expf(vp[r * I] * mp[r * I] - mx) * mp[r * I];
      tot += (gp[r * I] / sp - ge / (sp * sp)) * e;
    }
    float *op = gv + o * R * I + i;
    for (int r = 0; r < R; ++r) {
      const float e = expf(vp[r * I] * mp[r * I] - mx) * mp[r * I];
      float d = (gp[r * I] / sp - ge / (sp * sp)) * e;
      if (r == am) d -= tot;
      op[r * I] = d * mp[r * I];
    }
  }
}

// ------------------------------------------------------------------------------------------ context matching
// Kernel style (all of them): straight-line bodies whose loads are unconditional (indices clamped, contributions of
// padding lanes / padding references zeroed afterwards).  A run-time condition around a load makes hipcc wait for
// every outstanding load at the join, which serialises the memory round trips (the first version of these kernels
// ran at 3 TB/s for that reason).  The number of references handled per pass, RU, is a template parameter.
struct CtxWs {
  float *vt;       // (B,hw)   resized target visibility
  float *vr;       // (B,R,hw) resized reference visibility
  float *gs;       // (B,R)    similarity after normalisation / zeroing
  float *dscale;   // (B,R)    d gs / d raw sum  (0 where v_sum was zero)
  float *partial;  // (B,R,C,xs) per-channel, per-pixel-range partial sums of the raw similarity
  float *vsp;      // (B,R,xs)  overlap area  sum_x vt*vr, per pixel range
  unsigned *mbits; // (passes,B,hw) per pixel: bit 0 = vt, bit 1+k = vr of reference pass*ru + k (the similarity kernel's masks in one load)
  int passes, ru;  // references are handled in `passes` launches of `ru` <= CTX_RMAX each
  int xs;          // pixel ranges the similarity kernel splits a channel slab into (enough waves to fill the chip at small B*C)
  size_t bytes;
};
constexpr int CTX_PXW = 64;  // lanes along the pixel axis; each lane owns PX (1 or 2) adjacent pixels
constexpr int CTX_RC = 8;    // reference frames per pass in the backward match kernel

static int ctx_px(int64_t hw) { return hw % 2 == 0 ? 2 : 1; }
constexpr int CTX_RMAX = 16;  // references per launch of the slab kernels (one template instance per count); more -> several passes

static int ctx_xblocks(int64_t hw) { return (int)cdiv(hw, CTX_PXW * ctx_px(hw)); }

// Slab kernels run one wave per (target, channel [pair]); at C4 (B = 4, C = 128) that is 256 / 512 waves on 256 CUs, far too
// few loads in flight for HBM.  The pixel axis is therefore cut into `xs` ranges (grid.y) until about `want` waves exist, each
// range keeping at least 4 steps of 64 * PX pixels.  A function of the shape only: the summation order stays deterministic.
static int ctx_xsplit(int64_t waves, int64_t hw, int64_t want) {
  const int forced = 0;
  const int64_t nst = cdiv(hw, CTX_PXW * ctx_px(hw));
  int64_t xs = forced > 0 ? forced : cdiv(want, waves);
  if (xs > nst / 4) xs = nst / 4;
  if (xs > 16) xs = 16;
  return (int)(xs < 1 ? 1 : xs);
}

static CtxWs ctx_ws(void *base, int64_t B, int64_t C, int64_t R, int64_t hw) {
  CtxWs w;
  size_t off = 0;
  auto take = [&](size_t n) { float *p = (float *)((char *)base + off); off += round_up(n * sizeof(float), 256); return p; };
  w.vt = take((size_t)B * hw);
  w.vr = take((size_t)B * R * hw);
  w.gs = take((size_t)B * R);
  w.dscale = take((size_t)B * R);
  w.xs = ctx_xsplit(B * ((C + 1) / 2), hw, 2048);
  w.partial = take((size_t)B * R * C * w.xs);
  w.vsp = take((size_t)B * R * w.xs);
  w.passes = (int)cdiv(R, CTX_RMAX);
  w.ru = (int)cdiv(R, w.passes);
  w.mbits = (unsigned *)take((size_t)w.passes * B * hw);
  w.bytes = off;
  return w;
}

// Both mask resizes of the forward pass in one launch, one thread per (target, feature pixel): the target map and the R
// reference maps share the bilinear taps, and the thread assembles the pixel's bit field(s) itself, so the bit fields
// need neither a memset nor atomics.
template <int RU>
__global__ void __launch_bounds__(256) ctx_resize_masks_kernel(const float *__restrict__ v_t, const float *__restrict__ v_al,
                                                               float *__restrict__ vt, float *__restrict__ vr, unsigned *__restrict__ mbits,
                                                               int B, int R, int H, int W, int h, int w, int passes) {
  const float sh = __fdiv_rn((float)H, (float)h), sw = __fdiv_rn((float)W, (float)w);
  const int hw = h * w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * hw) return;
  const int b = i / hw, r = i - b * hw, y = r / w, x = r - y * w;
  const LinTap ty = linear_tap(y, H, sh), tx = linear_tap(x, W, sw);
  const int o00 = ty.i0 * W + tx.i0, o01 = ty.i0 * W + tx.i1, o10 = ty.i1 * W + tx.i0, o11 = ty.i1 * W + tx.i1;
  const float *pt = v_t + (int64_t)b * H * W;
  const float t00 = pt[o00], t01 = pt[o01], t10 = pt[o10], t11 = pt[o11];
  bool t_on = false;
  for (int q = 0; q < passes; ++q) {
    float v[RU][4];
#pragma unroll
    for (int k = 0; k < RU; ++k) {   // the 4 * RU taps of the pass in flight together (clamped plane index, selected below)
      const float *p = v_al + ((int64_t)b * R + min(q * RU + k, R - 1)) * H * W;
      v[k][0] = p[o00]; v[k][1] = p[o01]; v[k][2] = p[o10]; v[k][3] = p[o11];
    }
    TCL_LOADS_FIRST(4 * RU);
    if (q == 0) {
      t_on = bilerp(t00, t01, t10, t11, tx, ty) > 0.5f;
      vt[i] = t_on ? 1.f : 0.f;
    }
    unsigned bits = t_on ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      const int j = q * RU + k;
      const bool on = bilerp(v[k][0], v[k][1], v[k][2], v[k][3], tx, ty) > 0.5f;
      if (j < R) {
        vr[((int64_t)b * R + j) * hw + r] = on ? 1.f : 0.f;
        bits |= on ? 2u << k : 0u;
      }
    }
    mbits[((int64_t)q * B + b) * hw + r] = bits;
  }
}

// One wave per (target b, channel c): it streams the contiguous (R+1) x hw slab of that channel, 64*PX pixels of all
// R+1 frames per step (measured: the slab walk reads at 6.8 TB/s, an x-tiled channel loop at 5 TB/s).
//   partial[b][r][c] = sum_x vt[x] vr_r[x] f_t[c,x] f_r[c,x]     vsp[b][r] = sum_x vt vr_r  (written by the c == 0 wave)
// Sums run in a fixed order (x ascending per lane, then the wave tree): deterministic.  RU = R (template) so that the
// per-reference accumulators live in registers.
template <int RU, int PX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) ctx_sim_kernel(const float *cf, const unsigned *mbits, float *partial, float *vsp, int B, int C,
                                                      int hw, int R, int r0) {
  // (no __restrict__ on purpose: it lets hipcc spread the loads among their uses)
  // A wave owns the channel pair (c, c+1) of target b.  The wave index is made scalar so that the slab pointers live in
  // SGPRs.  The R+1 masks of a pixel arrive as one bit field (ctx_resize_masks_kernel), not as R+1 float loads.
  const int cpairs = (C + 1) >> 1;
  const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= B * cpairs) return;
  const int b = w / cpairs, c = (w - b * cpairs) * 2;
  const bool has2 = c + 1 < C;
  const int64_t cstride = (int64_t)(R + 1) * hw;
  const float *s1 = cf + ((int64_t)b * C + c) * cstride, *s2 = s1 + (has2 ? cstride : 0);
  const unsigned *mb = mbits + ((int64_t)(r0 / RU) * B + b) * hw;
  float acc1[RU], acc2[RU], vs[RU];
  int rr[RU];   // references r0 .. r0+RU-1 of this pass (clamped: the padding ones are loaded but never written)
#pragma unroll
  for (int k = 0; k < RU; ++k) { acc1[k] = acc2[k] = vs[k] = 0.f; rr[k] = min(r0 + k, R - 1); }
  const int xs = gridDim.y, xj = blockIdx.y, per = (hw + CTX_PXW * PX * xs - 1) / (CTX_PXW * PX * xs) * (CTX_PXW * PX);
  const int xbeg = xj * per, xend = min(hw, xbeg + per);   // this block's pixel range (whole steps; may be empty: zeros are written)
  for (int x0 = xbeg + lane * PX; x0 < xend + (CTX_PXW - 1) * PX; x0 += CTX_PXW * PX) {   // every lane takes the same number of steps
    const bool ok = x0 < xend;
    const int x = ok ? x0 : 0;
    float t1[PX], t2[PX], f1[RU][PX], f2[RU][PX];
    unsigned bits[PX];
    ldpx<PX>(s1 + x, t1);
    ldpx<PX>(s2 + x, t2);
    if (PX == 2) { const uint2 u = *reinterpret_cast<const uint2 *>(mb + x); bits[0] = u.x; bits[PX - 1] = u.y; }
    else bits[0] = mb[x];
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      ldpx<PX>(s1 + (rr[k] + 1) * hw + x, f1[k]);
      ldpx<PX>(s2 + (rr[k] + 1) * hw + x, f2[k]);
    }
    TCL_LOADS_FIRST(2 * RU + 3);   // all loads of the step in flight before the first use
#pragma unroll
    for (int e = 0; e < PX; ++e) bits[e] = (ok && (bits[e] & 1u)) ? bits[e] : 0u;   // vt == 0 (or a padding lane): no reference counts
#pragma unroll
    for (int k = 0; k < RU; ++k)
#pragma unroll
      for (int e = 0; e < PX; ++e) {
        const float mm = (float)((bits[e] >> (k + 1)) & 1u);
        acc1[k] = fmaf(t1[e] * f1[k][e], mm, acc1[k]);
        acc2[k] = fmaf(t2[e] * f2[k][e], mm, acc2[k]);
        vs[k] += mm;
      }
  }
#pragma unroll
  for (int k = 0; k < RU; ++k) {
    const float a = wave_sum(acc1[k]), a2 = wave_sum(acc2[k]);
    if (lane == 0 && r0 + k < R) {
      partial[(((int64_t)b * R + r0 + k) * C + c) * xs + xj] = a;
      if (has2) partial[(((int64_t)b * R + r0 + k) * C + c + 1) * xs + xj] = a2;
    }
  }
  if (c == 0) {   // wave-uniform: the overlap area, a count of {0,1} products (exact)
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      const float v = wave_sum(vs[k]);
      if (lane == 0 && r0 + k < R) vsp[(b * R + r0 + k) * xs + xj] = v;
    }
  }
}

// gs[b][r] from the C channel partials (v_sum, the `zeros` flag and the normalisation; flags bit-exact), by one wave.
__device__ __forceinline__ void ctx_gs(const float *partial, const float *vsp, int br, int C, int xs, int variant, int normalize, int lane,
                                       float &g_out, float &ds_out) {
  float raw = 0.f, vs = 0.f;
  const int n = C * xs;
  for (int k = lane; k < n; k += 64) raw += partial[(int64_t)br * n + k];
  raw = wave_sum(raw);
  for (int j = 0; j < xs; ++j) vs += vsp[br * xs + j];   // counts: exact in any order
  const bool zero = vs < 1e-4f;
  const float v_sum = zero ? vs + 1.f : vs;
  float g, ds;
  if (variant == TCL_CTX_CM_MODULE) {        // gs / v_sum / C            (cpn_original.py:226)
    g = (raw / v_sum) / (float)C;
    ds = (1.f / v_sum) / (float)C;
  } else if (normalize) {                    // gs / (v_sum * C)          (cpn.py:115-116)
    g = raw / (v_sum * (float)C);
    ds = 1.f / (v_sum * (float)C);
  } else {
    g = raw; ds = 1.f;
  }
  g_out = zero ? 0.f : g;
  ds_out = zero ? 0.f : ds;
}

// One thread per (b, pixel): softmax over the references -> c_match (B,R,hw), c_mask and the mask channel of p_in.
// The block first reduces the similarities of the (at most NBMAX) targets its 256 pixels belong to (what used to be a
// separate finalize launch; every block that needs a target's gs computes the same value in the same order) and the block
// that holds a target's first pixel stores gs / dscale for the backward pass.
constexpr int CTX_WB = 4;   // targets a block of 256 pixels can straddle when hw >= 86
template <int RU>   // RU = R when R <= CTX_RMAX (masks of the pixel loaded once, together), 0 = any R
__global__ void __launch_bounds__(256) ctx_weights_kernel(const float *__restrict__ vt, const float *__restrict__ vr,
                                                          const float *__restrict__ partial, const float *__restrict__ vsp,
                                                          float *__restrict__ gs, float *__restrict__ dscale, float *__restrict__ p_in,
                                                          float *__restrict__ c_mask, float *__restrict__ c_match, int B, int C, int R,
                                                          int hw, int variant, int normalize, int xs) {
  __shared__ float g_s[CTX_WB][64];
  const int i0 = blockIdx.x * 256, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b_first = i0 / hw, b_last = min((i0 + 255) / hw, B - 1), nb = b_last - b_first + 1;   // nb <= CTX_WB (host checks hw)
  for (int t = wv; t < nb * R; t += 4) {
    const int bl = t / R, r = t - bl * R, br = (b_first + bl) * R + r;
    float g, ds;
    ctx_gs(partial, vsp, br, C, xs, variant, normalize, lane, g, ds);
    if (lane == 0) {
      g_s[bl][r] = g;
      if ((int64_t)(b_first + bl) * hw >= i0) { gs[br] = g; dscale[br] = ds; }   // this block holds the target's first pixel
    }
  }
  __syncthreads();
  const int i = i0 + threadIdx.x;
  if (i >= B * hw) return;
  const int b = i / hw, x = i - b * hw;
  const float *vrb = vr + (int64_t)b * R * hw + x, *g = g_s[b - b_first];
  const float vtx = vt[i];
  const float hole = variant == TCL_CTX_COMPLETE ? 1.f - vtx : 1.f;
  float cm = 0.f;
  if (RU > 0) {
    float m[RU > 0 ? RU : 1], e[RU > 0 ? RU : 1];
#pragma unroll
    for (int r = 0; r < RU; ++r) m[r] = vrb[(int64_t)r * hw];
    TCL_LOADS_FIRST(RU);
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < RU; ++r) mx = fmaxf(mx, g[r] * m[r]);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RU; ++r) { e[r] = expf(g[r] * m[r] - mx) * m[r]; s += e[r]; }
    if (s < 1e-4f) s += 1.f;
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      const float wgt = (e[r] / s) * hole;
      c_match[((int64_t)b * R + r) * hw + x] = wgt;
      cm += wgt * m[r];
    }
  } else {
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, g[r] * vrb[(int64_t)r * hw]);
    float s = 0.f;
    for (int r = 0; r < R; ++r) {
      const float m = vrb[(int64_t)r * hw];
      s += expf(g[r] * m - mx) * m;
    }
    if (s < 1e-4f) s += 1.f;
    for (int r = 0; r < R; ++r) {
      const float m = vrb[(int64_t)r * hw];
      const float wgt = (expf(g[r] * m - mx) * m / s) * hole;
      c_match[((int64_t)b * R + r) * hw + x] = wgt;
      cm += wgt * m;
    }
  }
  cm = 1.f - cm;
  c_mask[i] = cm;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  p_in[((int64_t)b * (cch + 1) + cch) * hw + x] = cm;
}

// Small feature maps (hw < 64: a block would straddle more than CTX_WB targets): the two-launch form.
__global__ void __launch_bounds__(256) ctx_finalize_kernel(const float *__restrict__ partial, const float *__restrict__ vsp,
                                                           float *__restrict__ gs, float *__restrict__ dscale, int BR, int C,
                                                           int variant, int normalize, int xs) {
  const int br = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (br >= BR) return;
  float g, ds;
  ctx_gs(partial, vsp, br, C, xs, variant, normalize, lane, g, ds);
  if (lane == 0) { gs[br] = g; dscale[br] = ds; }
}

__global__ void __launch_bounds__(256) ctx_weights_plain_kernel(const float *__restrict__ vt, const float *__restrict__ vr,
                                                                const float *__restrict__ gs, float *__restrict__ p_in,
                                                                float *__restrict__ c_mask, float *__restrict__ c_match, int B, int C, int R,
                                                                int hw, int variant) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * hw) return;
  const int b = i / hw, x = i - b * hw;
  const float *vrb = vr + (int64_t)b * R * hw + x, *g = gs + b * R;
  const float vtx = vt[i];
  float mx = -INFINITY;
  for (int r = 0; r < R; ++r) mx = fmaxf(mx, g[r] * vrb[(int64_t)r * hw]);
  float s = 0.f;
  for (int r = 0; r < R; ++r) {
    const float m = vrb[(int64_t)r * hw];
    s += expf(g[r] * m - mx) * m;
  }
  if (s < 1e-4f) s += 1.f;
  const float hole = variant == TCL_CTX_COMPLETE ? 1.f - vtx : 1.f;
  float cm = 0.f;
  for (int r = 0; r < R; ++r) {
    const float m = vrb[(int64_t)r * hw];
    const float wgt = (expf(g[r] * m - mx) * m / s) * hole;
    c_match[((int64_t)b * R + r) * hw + x] = wgt;
    cm += wgt * m;
  }
  cm = 1.f - cm;
  c_mask[i] = cm;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  p_in[((int64_t)b * (cch + 1) + cch) * hw + x] = cm;
}

// One wave per (b, c), same slab walk: c_out[c,x] = sum_r f_r[c,x] w_r[x] (r ascending), written with the target
// features into the concatenated decoder input.
template <int RU, int PX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) ctx_aggregate_kernel(const float *cf, const float *vt, const float *c_match, float *p_in, int BC,
                                                            int C, int hw, int variant, int R, int r0) {
  // (no __restrict__ on purpose, see ctx_sim_kernel)
  const int bc = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar, see ctx_sim_kernel
  if (bc >= BC) return;
  const int b = bc / C, c = bc - b * C;
  const float *slab = cf + (int64_t)bc * (R + 1) * hw;
  const float *vtb = vt + (int64_t)b * hw, *wb = c_match + (int64_t)b * R * hw;
  const bool complete = variant == TCL_CTX_COMPLETE, last = r0 + RU >= R;
  float *o = p_in + (int64_t)b * ((complete ? C : 2 * C) + 1) * hw;
  float *oacc = o + (int64_t)(complete ? c : C + c) * hw;   // running sum over the passes (R > CTX_RMAX only)
  int rr[RU];
  float live[RU];
#pragma unroll
  for (int k = 0; k < RU; ++k) { rr[k] = min(r0 + k, R - 1); live[k] = r0 + k < R ? 1.f : 0.f; }
  const int xs = gridDim.y, per = (hw + CTX_PXW * PX * xs - 1) / (CTX_PXW * PX * xs) * (CTX_PXW * PX);
  const int xbeg = blockIdx.y * per, xend = min(hw, xbeg + per);   // this block's pixel range, see ctx_xsplit
  for (int x = xbeg + lane * PX; x < xend; x += CTX_PXW * PX) {
    float ft[PX], vtx[PX], fr[RU][PX], w[RU][PX], acc[PX];
    ldpx<PX>(slab + x, ft);
    ldpx<PX>(vtb + x, vtx);
#pragma unroll
    for (int e = 0; e < PX; ++e) acc[e] = 0.f;
    if (r0 > 0) ldpx<PX>(oacc + x, acc);
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      ldpx<PX>(slab + (rr[k] + 1) * hw + x, fr[k]);
      ldpx<PX>(wb + rr[k] * hw + x, w[k]);
    }
    TCL_LOADS_FIRST(2 * RU + 3);   // all loads of the step in flight before the first use
#pragma unroll
    for (int e = 0; e < PX; ++e)
#pragma unroll
      for (int k = 0; k < RU; ++k) acc[e] += fr[k][e] * (w[k][e] * live[k]);
    if (complete) {
      if (last) {
#pragma unroll
        for (int e = 0; e < PX; ++e) acc[e] = ft[e] * vtx[e] + acc[e];
      }
    } else if (r0 == 0) {
      stpx_stream<PX>(o + (int64_t)c * hw + x, ft);
    }
    if (last) stpx_stream<PX>(oacc + x, acc);   // the running sum is only read back by a later pass (R > CTX_RMAX)
    else stpx<PX>(oacc + x, acc);
  }
}

// ---- backward
// Pass 1a: dpart[cs][b][r][x] = sum over the block's channels of g_out[c,x] f_r[c,x].
//   grid (xblocks * csplit, B), block 256 = 4 waves: lane = PX pixels, wave = a strided channel subset, two channels per
//   step, RU references per pass (all loads of a step unconditional and in flight together); the four waves are
//   combined through LDS in a fixed order.
template <int RU, int PX>
__global__ void __launch_bounds__(256) ctx_bwd_dmatch_kernel(const float *g_p_in, const float *cf, float *dpart, int B, int C, int R, int hw,
                                                             int variant, int xblocks, int csplit) {
  constexpr int NPX = CTX_PXW * PX;
  __shared__ float red[4][RU][NPX];
  // grid (8 = XCD, x-chunk and channel split, batch item / 8): workgroups are dealt round-robin over the XCDs in linear order, so
  // blockIdx.x IS the XCD and all chunks of one batch item run on one XCD.  A 512-byte chunk of a 6,360-byte plane row starts at
  // no line boundary; with the (chunk, item) grid of round 2 neighbouring chunks sat on different XCDs and each fetched the
  // shared boundary line into its own L2: 1.36x the algorithmic bytes (TCC fetch counters), now one L2 serves both.
  const int b = blockIdx.z * kNumXCD + blockIdx.x;
  if (b >= B) return;   // block-uniform (padding blocks of the last group)
  const int bxi = blockIdx.y % xblocks, cs = blockIdx.y / xblocks;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = (bxi * CTX_PXW + lane) * PX;
  const bool ok = x0 < hw;
  const int x = ok ? x0 : 0;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  const int out_off = variant == TCL_CTX_COMPLETE ? 0 : C;   // channel offset of g_out inside g_p_in
  const float *gpb = g_p_in + ((int64_t)b * (cch + 1) + out_off) * hw + x;
  const int64_t cstride = (int64_t)(R + 1) * hw;
  const float *cfb = cf + (int64_t)b * C * cstride + x;
  const int cstep = 4 * csplit, c0 = cs * 4 + wv;
  for (int r0 = 0; r0 < R; r0 += RU) {
    float acc[RU][PX];
    int roff[RU];
#pragma unroll
    for (int k = 0; k < RU; ++k) {
      roff[k] = (min(r0 + k, R - 1) + 1) * hw;
#pragma unroll
      for (int e = 0; e < PX; ++e) acc[k][e] = 0.f;
    }
    for (int c = c0; c < C; c += 2 * cstep) {
      const bool has2 = c + cstep < C;
      const int c2 = has2 ? c + cstep : c;
      const float *f1 = cfb + c * cstride, *f2 = cfb + c2 * cstride;
      float g1[PX], g2[PX], r1[RU][PX], r2[RU][PX];
      ldpx<PX>(gpb + (int64_t)c * hw, g1);
      ldpx<PX>(gpb + (int64_t)c2 * hw, g2);
#pragma unroll
      for (int k = 0; k < RU; ++k) {
        ldpx<PX>(f1 + roff[k], r1[k]);
        ldpx<PX>(f2 + roff[k], r2[k]);
      }
      TCL_LOADS_FIRST(2 * RU + 2);
      const float w2 = has2 ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < RU; ++k)
#pragma unroll
        for (int e = 0; e < PX; ++e) {
          acc[k][e] += g1[e] * r1[k][e];
          acc[k][e] += w2 * (g2[e] * r2[k][e]);
        }
    }
#pragma unroll
    for (int k = 0; k < RU; ++k) stpx<PX>(&red[wv][k][lane * PX], acc[k]);
    __syncthreads();
    for (int i = tid; i < RU * NPX; i += 256) {
      const int k = i / NPX, px = i - k * NPX, xx = bxi * NPX + px;
      if (r0 + k < R && xx < hw)
        dpart[(((int64_t)cs * B + b) * R + r0 + k) * hw + xx] = (red[0][k][px] + red[1][k][px]) + (red[2][k][px] + red[3][k][px]);
    }
    __syncthreads();
  }
}

// Pass 1b: one thread per (b, pixel): d_match[r] = sum_cs dpart - g_mask vr_r, the softmax Jacobian (incl. the
//   `sums < 1e-4` branch and the max's subgradient), and per-block partial sums of d_gs[b,r] = sum_x d_cos[r][x].
//   grid (ceil(hw/256), B), dynamic LDS = R*256 floats.
__global__ void __launch_bounds__(256) ctx_bwd_jacobian_kernel(const float *__restrict__ g_p_in, const float *__restrict__ g_c_mask,
                                                               const float *__restrict__ dpart, const float *__restrict__ vt,
                                                               const float *__restrict__ vr, const float *__restrict__ gs,
                                                               float *__restrict__ partial, int B, int C, int R, int hw, int variant,
                                                               int csplit, int nbx) {
  extern __shared__ __attribute__((aligned(16))) float dm[];  // [R][256]
  __shared__ float red[4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int x0 = blockIdx.x * 256 + tid;
  const bool ok = x0 < hw;
  const int x = ok ? x0 : 0;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  const float *vrb = vr + (int64_t)b * R * hw + x;
  const float *g = gs + (int64_t)b * R;
  const float gmask = g_p_in[((int64_t)b * (cch + 1) + cch) * hw + x] + (g_c_mask ? g_c_mask[(int64_t)b * hw + x] : 0.f);
  const float hole = variant == TCL_CTX_COMPLETE ? 1.f - vt[(int64_t)b * hw + x] : 1.f;
  float mx = -INFINITY, s = 0.f, ge = 0.f, tot = 0.f;
  int am = 0;
  for (int r = 0; r < R; ++r) {
    const float z = g[r] * vrb[(int64_t)r * hw];
    if (z > mx) { mx = z; am = r; }
  }
  for (int r = 0; r < R; ++r) {
    const float m = vrb[(int64_t)r * hw];
    const float e = expf(g[r] * m - mx) * m;
    float d = 0.f;
    for (int q = 0; q < csplit; ++q) d += dpart[(((int64_t)q * B + b) * R + r) * hw + x];
    d = (d - gmask * m) * hole;
    dm[r * 256 + tid] = d;   // d loss / d softmax_r
    s += e;
    ge += d * e;
  }
  const float sp = s < 1e-4f ? s + 1.f : s;
  for (int r = 0; r < R; ++r) {
    const float m = vrb[(int64_t)r * hw];
    const float e = expf(g[r] * m - mx) * m;
    tot += (dm[r * 256 + tid] / sp - ge / (sp * sp)) * e;
  }
  for (int r = 0; r < R; ++r) {
    const float m = vrb[(int64_t)r * hw];
    const float e = expf(g[r] * m - mx) * m;
    float d = (dm[r * 256 + tid] / sp - ge / (sp * sp)) * e;
    if (r == am) d -= tot;
    d = ok ? d * m : 0.f;
    d = wave_sum(d);
    if (lane == 0) red[wv] = d;
    __syncthreads();
    if (tid == 0) partial[((int64_t)b * R + r) * nbx + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
}

// R <= 16: one wave per block over 64 pixels, the R masks, exponentials and d_match sums stay in registers (all loads issued
// before the first use), no LDS and no barrier; 4x the blocks of the generic kernel, which at C4's 60 x 106 map is 400 instead of 100.
template <int RU>
__global__ void __launch_bounds__(64) ctx_bwd_jacobian_small_kernel(const float *__restrict__ g_p_in, const float *__restrict__ g_c_mask,
                                                                    const float *__restrict__ dpart, const float *__restrict__ vt,
                                                                    const float *__restrict__ vr, const float *__restrict__ gs,
                                                                    float *__restrict__ partial, int B, int C, int hw, int variant,
                                                                    int csplit, int nbx) {
  const int b = blockIdx.y, lane = threadIdx.x;
  const int x0 = blockIdx.x * 64 + lane;
  const bool ok = x0 < hw;
  const int x = ok ? x0 : 0;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  const float *vrb = vr + (int64_t)b * RU * hw + x;
  const float *g = gs + (int64_t)b * RU;
  float m[RU], d[RU], e[RU];
#pragma unroll
  for (int r = 0; r < RU; ++r) m[r] = vrb[(int64_t)r * hw];
#pragma unroll
  for (int r = 0; r < RU; ++r) d[r] = dpart[((int64_t)b * RU + r) * hw + x];
  float gmask = g_p_in[((int64_t)b * (cch + 1) + cch) * hw + x];
  if (g_c_mask) gmask += g_c_mask[(int64_t)b * hw + x];
  const float hole = variant == TCL_CTX_COMPLETE ? 1.f - vt[(int64_t)b * hw + x] : 1.f;
  for (int q = 1; q < csplit; ++q) {
#pragma unroll
    for (int r = 0; r < RU; ++r) d[r] += dpart[(((int64_t)q * B + b) * RU + r) * hw + x];
  }
  float mx = -INFINITY, s = 0.f, ge = 0.f, tot = 0.f;
  int am = 0;
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    e[r] = g[r] * m[r];
    if (e[r] > mx) { mx = e[r]; am = r; }
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    e[r] = expf(e[r] - mx) * m[r];
    d[r] = (d[r] - gmask * m[r]) * hole;   // d loss / d softmax_r
    s += e[r];
    ge += d[r] * e[r];
  }
  const float sp = s < 1e-4f ? s + 1.f : s;
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    d[r] = (d[r] / sp - ge / (sp * sp)) * e[r];
    tot += d[r];
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    float o = r == am ? d[r] - tot : d[r];
    o = wave_sum(ok ? o * m[r] : 0.f);
    if (lane == 0) partial[((int64_t)b * RU + r) * nbx + blockIdx.x] = o;
  }
}

// coef[b,r] = dscale[b,r] * sum_blocks partial   (one wave per (b,r))
__global__ void __launch_bounds__(256) ctx_bwd_finalize_kernel(const float *__restrict__ partial, const float *__restrict__ dscale,
                                                               float *__restrict__ coef, int BR, int nbx) {
  const int br = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (br >= BR) return;
  float s = 0.f;
  for (int k = lane; k < nbx; k += 64) s += partial[(int64_t)br * nbx + k];
  s = wave_sum(s);
  if (lane == 0) coef[br] = s * dscale[br];
}

// Pass 2: grid (xblocks, ceil(C/4), B), block (64,4): lane = PX pixels, threadIdx.y = channel; writes all R+1 frames.
template <int PX>
__global__ void __launch_bounds__(256) ctx_bwd_feats_kernel(const float *__restrict__ g_p_in, const float *__restrict__ cf,
                                                            const float *__restrict__ c_match, const float *__restrict__ vt,
                                                            const float *__restrict__ vr, const float *__restrict__ coef,
                                                            float *__restrict__ g_cf, int C, int R, int hw, int variant, unsigned xblocks) {
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x), cgroups = (C + 3) / 4;
  const unsigned bxi = lin % xblocks, rest = lin / xblocks;
  const int b = rest / cgroups, c = (rest % cgroups) * 4 + threadIdx.y;
  const int x = (bxi * CTX_PXW + threadIdx.x) * PX;
  if (x >= hw || c >= C) return;
  const int cch = variant == TCL_CTX_COMPLETE ? C : 2 * C;
  const float *gpb = g_p_in + (int64_t)b * (cch + 1) * hw + x;
  float vtx[PX], g_ft[PX], g_out[PX], ft[PX];
  ldpx<PX>(vt + (int64_t)b * hw + x, vtx);
  if (variant == TCL_CTX_COMPLETE) {
    ldpx<PX>(gpb + (int64_t)c * hw, g_out);
#pragma unroll
    for (int e = 0; e < PX; ++e) g_ft[e] = g_out[e] * vtx[e];
  } else {
    ldpx<PX>(gpb + (int64_t)c * hw, g_ft);
    ldpx<PX>(gpb + (int64_t)(C + c) * hw, g_out);
  }
  const int64_t base = ((int64_t)b * C + c) * (R + 1) * hw + x;
  ldpx<PX>(cf + base, ft);
  for (int r = 0; r < R; ++r) {
    float m[PX], fr[PX], cm[PX], o[PX];
    ldpx<PX>(vr + ((int64_t)b * R + r) * hw + x, m);
    ldpx<PX>(cf + base + (int64_t)(r + 1) * hw, fr);
    ldpx<PX>(c_match + ((int64_t)b * R + r) * hw + x, cm);
    const float cr = coef[b * R + r];
#pragma unroll
    for (int e = 0; e < PX; ++e) {
      const float k = cr * (vtx[e] * m[e]);
      g_ft[e] += k * fr[e];
      o[e] = g_out[e] * cm[e] + k * ft[e];
    }
    stpx<PX>(g_cf + base + (int64_t)(r + 1) * hw, o);   // plain stores: streaming ones measured 2 % slower here
  }
  stpx<PX>(g_cf + base, g_ft);
}

// R <= 16: the same pass with the reference count as a template parameter: every load of the thread (3 per reference and
// channel-independent ones shared by its CH channels) is issued before the first use, then the R+1 stores follow.
template <int RU, int PX, int CH>
__global__ void __launch_bounds__(256) ctx_bwd_feats_small_kernel(const float *__restrict__ g_p_in, const float *__restrict__ cf,
                                                                  const float *__restrict__ c_match, const float *__restrict__ vt,
                                                                  const float *__restrict__ vr, const float *__restrict__ coef,
                                                                  float *__restrict__ g_cf, int C, int hw, int variant, unsigned xblocks) {
  const unsigned lin = xcd_remap(blockIdx.x, gridDim.x), cgroups = (C + 4 * CH - 1) / (4 * CH);
  const unsigned bxi = lin % xblocks, rest = lin / xblocks;
  const int b = rest / cgroups, c0 = ((rest % cgroups) * 4 + threadIdx.y) * CH;
  const int x = (bxi * CTX_PXW + threadIdx.x) * PX;
  if (x >= hw || c0 >= C) return;
  const bool complete = variant == TCL_CTX_COMPLETE;
  const int cch = complete ? C : 2 * C;
  const float *gpb = g_p_in + (int64_t)b * (cch + 1) * hw + x;
  float vtx[PX], g_ft[CH][PX], g_out[CH][PX], ft[CH][PX], m[RU][PX], cm[RU][PX], fr[CH][RU][PX];
  int64_t base[CH];
  ldpx<PX>(vt + (int64_t)b * hw + x, vtx);
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = min(c0 + j, C - 1);   // a padding channel repeats the last one and is not stored
    base[j] = ((int64_t)b * C + c) * (RU + 1) * hw + x;
    ldpx<PX>(gpb + (int64_t)c * hw, g_ft[j]);
    ldpx<PX>(gpb + (int64_t)(complete ? c : C + c) * hw, g_out[j]);
    ldpx<PX>(cf + base[j], ft[j]);
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    ldpx<PX>(vr + ((int64_t)b * RU + r) * hw + x, m[r]);
    ldpx<PX>(c_match + ((int64_t)b * RU + r) * hw + x, cm[r]);
#pragma unroll
    for (int j = 0; j < CH; ++j) ldpx<PX>(cf + base[j] + (int64_t)(r + 1) * hw, fr[j][r]);
  }
  TCL_LOADS_FIRST(1 + 3 * CH + (2 + CH) * RU);
  if (complete) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int e = 0; e < PX; ++e) g_ft[j][e] = g_out[j][e] * vtx[e];
  }
#pragma unroll
  for (int r = 0; r < RU; ++r) {
    const float cr = coef[b * RU + r];
    float k[PX];
#pragma unroll
    for (int e = 0; e < PX; ++e) k[e] = cr * (vtx[e] * m[r][e]);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float o[PX];
#pragma unroll
      for (int e = 0; e < PX; ++e) {
        g_ft[j][e] += k[e] * fr[j][r][e];
        o[e] = g_out[j][e] * cm[r][e] + k[e] * ft[j][e];
      }
      if (c0 + j < C) stpx<PX>(g_cf + base[j] + (int64_t)(r + 1) * hw, o);
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j)
    if (c0 + j < C) stpx<PX>(g_cf + base[j], g_ft[j]);
}

static unsigned stream_blocks(int64_t total) {
  int64_t blocks = cdiv(total, 256);
  if (blocks > kMaxStreamBlocks * 2) blocks = kMaxStreamBlocks * 2;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace tcl

using namespace tcl;

// ============================================================================================ C ABI
extern "C" int tcl_resize_mask(const float *v, float *out, int64_t N, int64_t H, int64_t W, int64_t h, int64_t w, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && h > 0 && w > 0 && H * W < (1LL << 31) && h * w < (1LL << 31), TCL_ERR_SHAPE, "tcl_resize_mask: bad shape");
  TCL_REQUIRE(v && out, TCL_ERR_ARG, "tcl_resize_mask: NULL pointer");
  hipLaunchKernelGGL(resize_mask_kernel, dim3(stream_blocks(N * h * w)), dim3(256), 0, (hipStream_t)stream, v, out, N, (int)H, (int)W, (int)h, (int)w);
  return check_launch("tcl_resize_mask");
}

extern "C" int tcl_dilate_cross3(const float *img, float *out, int64_t N, int64_t H, int64_t W, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && H * W < (1LL << 31), TCL_ERR_SHAPE, "tcl_dilate_cross3: bad shape");
  TCL_REQUIRE(img && out && img != out, TCL_ERR_ARG, "tcl_dilate_cross3: NULL pointer or in-place call");
  hipLaunchKernelGGL(dilate_cross3_kernel, dim3(stream_blocks(N * H * W)), dim3(256), 0, (hipStream_t)stream, img, out, N, (int)H, (int)W);
  return check_launch("tcl_dilate_cross3");
}

extern "C" int tcl_ref_importance(const float *c_match, const float *v_t, float *ref_importance, int64_t B, int64_t R, int64_t h,
                                  int64_t w, int64_t H, int64_t W, int middle_is_vt, void *stream) {
  TCL_REQUIRE(B > 0 && R > 0 && h > 0 && w > 0 && H > 0 && W > 0 && H * W < (1LL << 31), TCL_ERR_SHAPE, "tcl_ref_importance: bad shape");
  TCL_REQUIRE(c_match && v_t && ref_importance, TCL_ERR_ARG, "tcl_ref_importance: NULL pointer");
  TCL_REQUIRE(B * (R + 1) < 65536, TCL_ERR_SHAPE, "tcl_ref_importance: too many planes for one launch (%ld)", (long)(B * (R + 1)));
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;   // IEEE division, as __fdiv_rn
  if (W % 4 == 0 && aligned(v_t, 16) && aligned(ref_importance, 16))
    hipLaunchKernelGGL(ref_importance_kernel<4>, dim3((unsigned)cdiv(H * W, 1024), (unsigned)(B * (R + 1))), dim3(256), 0, (hipStream_t)stream,
                       c_match, v_t, ref_importance, (int)B, (int)R, (int)h, (int)w, (int)H, (int)W, middle_is_vt, sh, sw);
  else
    hipLaunchKernelGGL(ref_importance_kernel<1>, dim3((unsigned)cdiv(H * W, 256), (unsigned)(B * (R + 1))), dim3(256), 0, (hipStream_t)stream,
                       c_match, v_t, ref_importance, (int)B, (int)R, (int)h, (int)w, (int)H, (int)W, middle_is_vt, sh, sw);
  return check_launch("tcl_ref_importance");
}

extern "C" int tcl_masked_softmax_fwd(const float *vec, const float *mask, float *out, int64_t O, int64_t Om, int64_t R, int64_t I, void *stream) {
  TCL_REQUIRE(O > 0 && Om > 0 && R > 0 && I > 0 && O % Om == 0 && R < (1 << 20), TCL_ERR_SHAPE, "tcl_masked_softmax_fwd: bad shape O=%ld Om=%ld R=%ld I=%ld", (long)O, (long)Om, (long)R, (long)I);
  TCL_REQUIRE(vec && mask && out, TCL_ERR_ARG, "tcl_masked_softmax_fwd: NULL pointer");
  const dim3 grd((unsigned)cdiv(O * I, 256)), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define TCL_MSM(RR) case RR: hipLaunchKernelGGL(masked_softmax_fwd_small_kernel<RR>, grd, blk, 0, s, vec, mask, out, O, Om, I); break;
  if (R <= 16 && cdiv(O * I, 256) < (1LL << 31)) {
    switch ((int)R) {
      TCL_MSM(1) TCL_MSM(2) TCL_MSM(3) TCL_MSM(4) TCL_MSM(5) TCL_MSM(6) TCL_MSM(7) TCL_MSM(8)
      TCL_MSM(9) TCL_MSM(10) TCL_MSM(11) TCL_MSM(12) TCL_MSM(13) TCL_MSM(14) TCL_MSM(15) TCL_MSM(16)
    }
  } else
    hipLaunchKernelGGL(masked_softmax_fwd_kernel, dim3(stream_blocks(O * I)), dim3(256), 0, s, vec, mask, out, O, Om, (int)R, I);
#undef TCL_MSM
  return check_launch("tcl_masked_softmax_fwd");
}

extern "C" int tcl_masked_softmax_bwd(const float *grad_out, const float *vec, const float *mask, float *g_vec, int64_t O, int64_t Om, int64_t R, int64_t I, void *stream) {
  TCL_REQUIRE(O > 0 && Om > 0 && R > 0 && I > 0 && O % Om == 0 && R < (1 << 20), TCL_ERR_SHAPE, "tcl_masked_softmax_bwd: bad shape");
  TCL_REQUIRE(grad_out && vec && mask && g_vec, TCL_ERR_ARG, "tcl_masked_softmax_bwd: NULL pointer");
  const dim3 grd((unsigned)cdiv(O * I, 256)), blk(256);
  hipStream_t s = (hipStream_t)stream;
#define TCL_MSM(RR) case RR: hipLaunchKernelGGL(masked_softmax_bwd_small_kernel<RR>, grd, blk, 0, s, grad_out, vec, mask, g_vec, O, Om, I); break;
  if (R <= 16 && cdiv(O * I, 256) < (1LL << 31)) {
    switch ((int)R) {
      TCL_MSM(1) TCL_MSM(2) TCL_MSM(3) TCL_MSM(4) TCL_MSM(5) TCL_MSM(6) TCL_MSM(7) TCL_MSM(8)
      TCL_MSM(9) TCL_MSM(10) TCL_MSM(11) TCL_MSM(12) TCL_MSM(13) TCL_MSM(14) TCL_MSM(15) TCL_MSM(16)
    }
  } else
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3(stream_blocks(O * I)), dim3(256), 0, s, grad_out, vec, mask, g_vec, O, Om, (int)R, I);
#undef TCL_MSM
  return check_launch("tcl_masked_softmax_bwd");
}

static int ctx_check(const char *fn, int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int variant) {
  TCL_REQUIRE(B > 0 && C > 0 && R > 0 && h > 0 && w > 0, TCL_ERR_SHAPE, "%s: bad shape B=%ld C=%ld R=%ld h=%ld w=%ld", fn, (long)B, (long)C, (long)R, (long)h, (long)w);
  TCL_REQUIRE(B < 65536 && h * w < (1 << 24) && C * h * w < (1LL << 31) && R <= 64 && B * C < (1LL << 31) && B * h * w < (1LL << 31), TCL_ERR_SHAPE, "%s: shape out of range (R <= 64, B < 65536)", fn);
  TCL_REQUIRE(variant >= TCL_CTX_DEFAULT && variant <= TCL_CTX_CM_MODULE, TCL_ERR_ARG, "%s: unknown variant %d", fn, variant);
  return TCL_OK;
}

extern "C" size_t tcl_ctx_workspace_bytes(int64_t B, int64_t C, int64_t R, int64_t h, int64_t w) {
  if (B <= 0 || C <= 0 || R <= 0 || h <= 0 || w <= 0) return 0;
  return ctx_ws(nullptr, B, C, R, h * w).bytes;
}

extern "C" int tcl_ctx_fwd(const float *c_feats, const float *v_t, const float *v_aligned, float *p_in, float *c_mask, float *c_match,
                           int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int64_t H, int64_t W, int variant, int normalize,
                           void *workspace, size_t workspace_bytes, void *stream) {
  int rc = ctx_check("tcl_ctx_fwd", B, C, R, h, w, variant);
  if (rc) return rc;
  TCL_REQUIRE(H > 0 && W > 0 && H * W < (1LL << 31), TCL_ERR_SHAPE, "tcl_ctx_fwd: bad mask shape");
  TCL_REQUIRE(c_feats && v_t && v_aligned && p_in && c_mask && c_match, TCL_ERR_ARG, "tcl_ctx_fwd: NULL pointer");
  const size_t need = tcl_ctx_workspace_bytes(B, C, R, h, w);
  TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
              "tcl_ctx_fwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const int hw = (int)(h * w);
  const CtxWs ws = ctx_ws(workspace, B, C, R, hw);

  const int px = ctx_px(hw);
  const int BC = (int)(B * C);
  const dim3 grd((unsigned)cdiv(BC, 4), (unsigned)ctx_xsplit(BC, hw, 4096)), blk(256);
  const int ru = ws.ru;
  int r0 = 0;
#define TCL_CTX_SIM(RU, PX) hipLaunchKernelGGL((ctx_sim_kernel<RU, PX>), dim3((unsigned)cdiv(B * ((C + 1) / 2), 4), (unsigned)ws.xs), blk, 0, s, c_feats, ws.mbits, ws.partial, ws.vsp, (int)B, (int)C, hw, (int)R, r0)
#define TCL_CTX_AGG(RU, PX) hipLaunchKernelGGL((ctx_aggregate_kernel<RU, PX>), grd, blk, 0, s, c_feats, ws.vt, c_match, p_in, BC, (int)C, hw, variant, (int)R, r0)
#define TCL_CTX_BY_R(WHAT)                                                                  \
  do {                                                                                      \
    switch (ru * 2 + (px - 1)) {                                                            \
      case 2: WHAT(1, 1); break;   case 3: WHAT(1, 2); break;   case 4: WHAT(2, 1); break;   case 5: WHAT(2, 2); break;    \
      case 6: WHAT(3, 1); break;   case 7: WHAT(3, 2); break;   case 8: WHAT(4, 1); break;   case 9: WHAT(4, 2); break;    \
      case 10: WHAT(5, 1); break;  case 11: WHAT(5, 2); break;  case 12: WHAT(6, 1); break;  case 13: WHAT(6, 2); break;   \
      case 14: WHAT(7, 1); break;  case 15: WHAT(7, 2); break;  case 16: WHAT(8, 1); break;  case 17: WHAT(8, 2); break;   \
      case 18: WHAT(9, 1); break;  case 19: WHAT(9, 2); break;  case 20: WHAT(10, 1); break; case 21: WHAT(10, 2); break;  \
      case 22: WHAT(11, 1); break; case 23: WHAT(11, 2); break; case 24: WHAT(12, 1); break; case 25: WHAT(12, 2); break;  \
      case 26: WHAT(13, 1); break; case 27: WHAT(13, 2); break; case 28: WHAT(14, 1); break; case 29: WHAT(14, 2); break;  \
      case 30: WHAT(15, 1); break; case 31: WHAT(15, 2); break; case 32: WHAT(16, 1); break; default: WHAT(16, 2); break;  \
    }                                                                                       \
  } while (0)
#define TCL_CTX_RSZ(RU, PX) hipLaunchKernelGGL((ctx_resize_masks_kernel<RU>), dim3((unsigned)cdiv(B * hw, 256)), blk, 0, s, v_t, v_aligned, ws.vt, ws.vr, ws.mbits, (int)B, (int)R, (int)H, (int)W, (int)h, (int)w, ws.passes)
  TCL_CTX_BY_R(TCL_CTX_RSZ);
#undef TCL_CTX_RSZ
  for (r0 = 0; r0 < R; r0 += ru) TCL_CTX_BY_R(TCL_CTX_SIM);
  if (hw >= 86) {   // 256 consecutive pixels touch at most ceil(255 / hw) + 1 <= CTX_WB targets
#define TCL_CTX_WGT(RU, PX) hipLaunchKernelGGL((ctx_weights_kernel<RU>), dim3((unsigned)cdiv(B * hw, 256)), blk, 0, s, ws.vt, ws.vr, ws.partial, ws.vsp, ws.gs, ws.dscale, p_in, c_mask, c_match, (int)B, (int)C, (int)R, hw, variant, normalize, ws.xs)
    if (ws.passes == 1) TCL_CTX_BY_R(TCL_CTX_WGT);   // ru == R
    else TCL_CTX_WGT(0, 1);
#undef TCL_CTX_WGT
  } else {
    hipLaunchKernelGGL(ctx_finalize_kernel, dim3((unsigned)cdiv(B * R, 4)), dim3(256), 0, s, ws.partial, ws.vsp, ws.gs, ws.dscale, (int)(B * R),
                       (int)C, variant, normalize, ws.xs);
    hipLaunchKernelGGL(ctx_weights_plain_kernel, dim3((unsigned)cdiv(B * hw, 256)), dim3(256), 0, s, ws.vt, ws.vr, ws.gs, p_in, c_mask, c_match,
                       (int)B, (int)C, (int)R, hw, variant);
  }
  for (r0 = 0; r0 < R; r0 += ru) TCL_CTX_BY_R(TCL_CTX_AGG);
#undef TCL_CTX_SIM
#undef TCL_CTX_AGG
  return check_launch("tcl_ctx_fwd");
}

// ------------------------------------------------------------------------------------------ the entry point that allocates nothing
// tcl_ctx_fwd for callers that cannot hand over a workspace because they cannot allocate one (HIP graph capture, tight
// serving loops): the forward's scratch (resized masks, bit fields, per-channel partial sums) lives in this file-scope device
// array instead.  Every GPU the library runs on holds its own 4 MiB copy until the process ends.  In host code the array's
// name is only a stand-in; its device address on the CURRENT device comes from hipGetSymbolAddress, looked up once per
// device and kept.
constexpr size_t kCtxStaticBytes = (size_t)4 << 20;
static __device__ __align__(256) unsigned char scratch[kCtxStaticBytes];
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

static int ctx_static_scratch(void **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices)
    return set_error(TCL_ERR_LAUNCH, "tcl_ctx_fwd_static: hipGetDevice failed or device index %d out of range (%s)", dev, hipGetErrorString(e));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void *ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return set_error(TCL_ERR_LAUNCH, "tcl_ctx_fwd_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return TCL_OK;
}

extern "C" int tcl_ctx_fwd_static(const float *c_feats, const float *v_t, const float *v_aligned, float *p_in, float *c_mask, float *c_match,
                                  int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int64_t H, int64_t W, int variant, int normalize,
                                  void *stream) {
  int rc = ctx_check("tcl_ctx_fwd_static", B, C, R, h, w, variant);
  if (rc) return rc;
  const size_t need = tcl_ctx_workspace_bytes(B, C, R, h, w);
  TCL_REQUIRE(need <= kCtxStaticBytes, TCL_ERR_WORKSPACE,
              "tcl_ctx_fwd_static: this shape needs %zu bytes of scratch, the static buffer holds %zu (use tcl_ctx_fwd with a workspace)", need,
              kCtxStaticBytes);
  void *ptr = nullptr;
  rc = ctx_static_scratch(&ptr);
  if (rc) return rc;
  // every word of the scratch the kernels read is written earlier in the same call (no accumulation across calls, nothing
  // that has to start at zero), so the previous call's contents need no clearing
  return tcl_ctx_fwd(c_feats, v_t, v_aligned, p_in, c_mask, c_match, B, C, R, h, w, H, W, variant, normalize, ptr, kCtxStaticBytes, stream);
}

static int ctx_bwd_csplit(int64_t B, int64_t C, int64_t hw) {
  // blocks of the d_match kernel = B * xblocks * csplit: as many as are resident at once (8 per CU), never more
  int64_t want = ((int64_t)kNumCU * 8) / (B * ctx_xblocks(hw));
  const int64_t maxs = cdiv(C, 4 * 4);  // at least 4 channels per wave
  if (want > maxs) want = maxs;
  return (int)(want < 1 ? 1 : want);
}
static int ctx_bwd_ru(int64_t R) { return (int)cdiv(R, cdiv(R, CTX_RC)); }   // R = 10 -> two passes of 5

extern "C" size_t tcl_ctx_bwd_workspace_bytes(int64_t B, int64_t C, int64_t R, int64_t h, int64_t w) {
  if (B <= 0 || C <= 0 || R <= 0 || h <= 0 || w <= 0) return 0;
  return round_up((size_t)B * R * cdiv(h * w, 64) * 4, 256) + round_up((size_t)B * R * 4, 256) +
         round_up((size_t)ctx_bwd_csplit(B, C, h * w) * B * R * h * w * 4, 256);
}

extern "C" int tcl_ctx_bwd(const float *g_p_in, const float *g_c_mask, const float *c_feats, const float *c_match, const void *fwd_workspace,
                           float *g_c_feats, int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int variant, int normalize,
                           void *workspace, size_t workspace_bytes, void *stream) {
  (void)normalize;  // already folded into the forward's dscale
  int rc = ctx_check("tcl_ctx_bwd", B, C, R, h, w, variant);
  if (rc) return rc;
  TCL_REQUIRE(g_p_in && c_feats && c_match && fwd_workspace && g_c_feats, TCL_ERR_ARG, "tcl_ctx_bwd: NULL pointer");
  TCL_REQUIRE(aligned(fwd_workspace, 256), TCL_ERR_WORKSPACE, "tcl_ctx_bwd: fwd_workspace must be 256-byte aligned");
  const size_t need = tcl_ctx_bwd_workspace_bytes(B, C, R, h, w);
  TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
              "tcl_ctx_bwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const bool small = R <= 16;   // one wave per 64 pixels (registers only), else 256 pixels per block through LDS
  const int hw = (int)(h * w), nbx = (int)cdiv(hw, small ? 64 : 256);
  const CtxWs ws = ctx_ws(const_cast<void *>(fwd_workspace), B, C, R, hw);
  float *partial = (float *)workspace;
  float *coef = (float *)((char *)workspace + round_up((size_t)B * R * cdiv(hw, 64) * 4, 256));
  float *dpart = (float *)((char *)coef + round_up((size_t)B * R * 4, 256));
  const int csplit = ctx_bwd_csplit(B, C, hw), xb = ctx_xblocks(hw), px = ctx_px(hw), ru = ctx_bwd_ru(R);
  TCL_REQUIRE(xb * csplit < 65536, TCL_ERR_SHAPE, "tcl_ctx_bwd: feature map too large (%d pixels)", hw);
  const dim3 grd(kNumXCD, (unsigned)(xb * csplit), (unsigned)cdiv(B, kNumXCD)), blk(256);
#define TCL_CTX_DM(RU, PX) hipLaunchKernelGGL((ctx_bwd_dmatch_kernel<RU, PX>), grd, blk, 0, s, g_p_in, c_feats, dpart, (int)B, (int)C, (int)R, hw, variant, xb, csplit)
#define TCL_CTX_DM_PX(RU)                \
  do {                                   \
    if (px == 2) TCL_CTX_DM(RU, 2);      \
    else TCL_CTX_DM(RU, 1);              \
  } while (0)
  switch (ru) {
    case 1: TCL_CTX_DM_PX(1); break;
    case 2: TCL_CTX_DM_PX(2); break;
    case 3: TCL_CTX_DM_PX(3); break;
    case 4: TCL_CTX_DM_PX(4); break;
    case 5: TCL_CTX_DM_PX(5); break;
    case 6: TCL_CTX_DM_PX(6); break;
    case 7: TCL_CTX_DM_PX(7); break;
    default: TCL_CTX_DM_PX(8); break;
  }
#undef TCL_CTX_DM_PX
#undef TCL_CTX_DM
#define TCL_CTX_JAC(RU) case RU: hipLaunchKernelGGL(ctx_bwd_jacobian_small_kernel<RU>, dim3((unsigned)nbx, (unsigned)B), dim3(64), 0, s, g_p_in, g_c_mask, dpart, ws.vt, ws.vr, ws.gs, partial, (int)B, (int)C, hw, variant, csplit, nbx); break
  switch (small ? (int)R : 0) {
    TCL_CTX_JAC(1); TCL_CTX_JAC(2); TCL_CTX_JAC(3); TCL_CTX_JAC(4); TCL_CTX_JAC(5); TCL_CTX_JAC(6); TCL_CTX_JAC(7); TCL_CTX_JAC(8);
    TCL_CTX_JAC(9); TCL_CTX_JAC(10); TCL_CTX_JAC(11); TCL_CTX_JAC(12); TCL_CTX_JAC(13); TCL_CTX_JAC(14); TCL_CTX_JAC(15); TCL_CTX_JAC(16);
    default:
      hipLaunchKernelGGL(ctx_bwd_jacobian_kernel, dim3((unsigned)nbx, (unsigned)B), dim3(256), (size_t)R * 256 * sizeof(float), s, g_p_in, g_c_mask,
                         dpart, ws.vt, ws.vr, ws.gs, partial, (int)B, (int)C, (int)R, hw, variant, csplit, nbx);
  }
#undef TCL_CTX_JAC
  hipLaunchKernelGGL(ctx_bwd_finalize_kernel, dim3((unsigned)cdiv(B * R, 4)), dim3(256), 0, s, partial, ws.dscale, coef, (int)(B * R), nbx);
  // two channels per thread: the masks and match weights of a pixel are loaded once for both (0.216 -> 0.200 ms at C4 over one)
#define TCL_CTX_FE3(RU, PX, CH) hipLaunchKernelGGL((ctx_bwd_feats_small_kernel<RU, PX, CH>), dim3((unsigned)(ctx_xblocks(hw) * cdiv(C, 4 * CH) * B)), dim3(64, 4), 0, s, g_p_in, c_feats, c_match, ws.vt, ws.vr, coef, g_c_feats, (int)C, hw, variant, (unsigned)ctx_xblocks(hw))
#define TCL_CTX_FE(RU) case RU: if (px == 2) TCL_CTX_FE3(RU, 2, 2); else TCL_CTX_FE3(RU, 1, 2); break
  switch (small ? (int)R : 0) {
    TCL_CTX_FE(1); TCL_CTX_FE(2); TCL_CTX_FE(3); TCL_CTX_FE(4); TCL_CTX_FE(5); TCL_CTX_FE(6); TCL_CTX_FE(7); TCL_CTX_FE(8);
    TCL_CTX_FE(9); TCL_CTX_FE(10); TCL_CTX_FE(11); TCL_CTX_FE(12); TCL_CTX_FE(13); TCL_CTX_FE(14); TCL_CTX_FE(15); TCL_CTX_FE(16);
    default:
  if (ctx_px(hw) == 2)
    hipLaunchKernelGGL(ctx_bwd_feats_kernel<2>, dim3((unsigned)(ctx_xblocks(hw) * cdiv(C, 4) * B)), dim3(64, 4), 0, s, g_p_in, c_feats,
                       c_match, ws.vt, ws.vr, coef, g_c_feats, (int)C, (int)R, hw, variant, (unsigned)ctx_xblocks(hw));
  else
    hipLaunchKernelGGL(ctx_bwd_feats_kernel<1>, dim3((unsigned)(ctx_xblocks(hw) * cdiv(C, 4) * B)), dim3(64, 4), 0, s, g_p_in, c_feats,
                       c_match, ws.vt, ws.vr, coef, g_c_feats, (int)C, (int)R, hw, variant, (unsigned)ctx_xblocks(hw));
  }
#undef TCL_CTX_FE
#undef TCL_CTX_FE3
  return check_launch("tcl_ctx_bwd");
}
