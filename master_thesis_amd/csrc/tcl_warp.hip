// Grid / affine warp (bilinear + nearest), forward and backward, for gfx950.
//
// Replaces the F.grid_sample pairs of align_set (utils/flow.py:93-101) and of CPNet.align
// (models/cpn.py:280-295) / CPNOriginal.align (models/cpn_original.py:276-284); see include/tcl.h.
//
// HBM-bound streaming kernels: one lane per output pixel, the sampling position is read (or derived from
// theta) once and shared by all channels and by the visibility map; frames are addressed through strides,
// so the reference's (B,C,F,H,W) tensors are read/written in place.  Index math is done with explicitly
// rounded operations (no fma contraction) in the order ATen's CPU kernel uses, so that nearest indices
// and the (> 0.5) threshold are bit-exact with the reference.
#include "tcl_common.h"

namespace tcl {

struct WarpArgs {
  const float *x;  int64_t x_sb, x_sf, x_sc;
  const float *v;  int64_t v_sb, v_sf;
  const float *grid, *theta;
  float *xo; int64_t xo_sb, xo_sf, xo_sc;
  float *vo; int64_t vo_sb, vo_sf;
  int NB, NF, C, H, W;
  int align_corners, v_mode;
  float stepx, stepy;  // 2/(W-1), 2/(H-1) (torch.linspace step), divided on the host
  int gf;              // frames per XCD unit (see warp_block): NF when the frames of a batch item share one source frame
  FastDiv fd_w, fd_nf; // / W (pixel -> row), / NF (image -> batch item)
  // fused epilogue of the inference unit (tcl_warp_fwd_fused): v holds HOLE masks (visibility = 1 - v, flipped per tap);
  // v_map = clamp(v_out - v_target, 0, 1) written beside v_out (T1, thesis_dfpn/runner.py:158)
  int v_flip, vt_flip;
  const float *vt; int64_t vt_sb;
  float *vm; int64_t vm_sb, vm_sf;
};

// Grid (8, chunks per unit, units / 8) -> (image n, 256-pixel chunk).  Workgroups are dealt round-robin over the 8 XCDs in
// linear order (x fastest), so blockIdx.x IS the XCD: unit u (an image, or all `gf` frames of a batch item when they
// sample ONE shared source frame) lives on XCD u % 8, which walks the unit's chunks in order.  The taps of neighbouring
// chunks share source rows and now hit in ONE L2 instead of being fetched by up to eight (align_set fwd 5.7 -> 6.7 TB/s).
// No integer division in the common gf == 1 case.  Returns false for the padding blocks of the last unit group.
__device__ __forceinline__ bool warp_block(int nchunk, int gf, int nimg, int &n, int &chunk) {
  const int unit = blockIdx.z * kNumXCD + blockIdx.x;
  const int fi = gf == 1 ? 0 : (int)blockIdx.y / nchunk;
  n = unit * gf + fi;
  chunk = blockIdx.y - fi * nchunk;
  return n < nimg;
}
static inline dim3 warp_grid(int64_t nchunk, int64_t gf, int64_t nimg) {
  return dim3(kNumXCD, (unsigned)(nchunk * gf), (unsigned)cdiv(nimg / gf, kNumXCD));
}

__device__ __forceinline__ float unnormalise(float g, int size, bool align_corners) {
  if (align_corners) return __fmul_rn(__fmul_rn(__fadd_rn(g, 1.f), 0.5f), (float)(size - 1));
  return __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(g, 1.f), (float)size), 1.f), 0.5f);
}

struct Taps {
  float nw, ne, sw, se;  // weights
  float wx, ex, ny, sy;  // distances
  int xi, yi;            // integer corner (valid where a flag is set)
  int xa, ya, yb;        // clamped column of the 2-wide load and clamped north / south rows: always inside the image
  bool xw, xe, yn, ys;   // west / east column and north / south row inside the image
};

__device__ __forceinline__ Taps make_taps(float ix, float iy, int H, int W) {
  Taps t;
  const float x0 = floorf(ix), y0 = floorf(iy);
  t.wx = __fsub_rn(ix, x0);
  t.ex = __fsub_rn(1.f, t.wx);
  t.ny = __fsub_rn(iy, y0);
  t.sy = __fsub_rn(1.f, t.ny);
  t.nw = __fmul_rn(t.sy, t.ex);
  t.ne = __fmul_rn(t.sy, t.wx);
  t.sw = __fmul_rn(t.ny, t.ex);
  t.se = __fmul_rn(t.ny, t.wx);
  // bounds in the float domain (NaN / huge coordinates are simply out of bounds)
  t.xw = (x0 >= 0.f) & (x0 <= (float)(W - 1));
  t.xe = (x0 >= -1.f) & (x0 <= (float)(W - 2));
  t.yn = (y0 >= 0.f) & (y0 <= (float)(H - 1));
  t.ys = (y0 >= -1.f) & (y0 <= (float)(H - 2));
  t.xi = (t.xw | t.xe) ? (int)x0 : 0;
  t.yi = (t.yn | t.ys) ? (int)y0 : 0;
  t.xa = max(0, min(t.xi, W - 2));
  t.ya = max(0, min(t.yi, H - 1));
  t.yb = max(0, min(t.yi + 1, H - 1));
  return t;
}

// The west/east neighbours of one image row with ONE unconditional 8-byte load at the clamped column (planes are only
// 4-byte aligned), then selects: straight-line code, so all taps of all channels are in flight together.  W2 = (W >= 2).
struct __attribute__((packed, aligned(4))) F2U { float a, b; };
struct TapVals { F2U n, s; };

template <bool W2>
__device__ __forceinline__ TapVals fetch_taps(const float *plane, const Taps &t, int W) {
  TapVals v;
  if (W2) {
    // unsigned 32-bit offsets off the (wave-uniform) plane pointer: one scalar base + vector offset per load, no 64-bit VALU
    v.n = *reinterpret_cast<const F2U *>(plane + (unsigned)(t.ya * W + t.xa));
    v.s = *reinterpret_cast<const F2U *>(plane + (unsigned)(t.yb * W + t.xa));
  } else {  // one-pixel-wide image
    v.n.a = plane[t.ya * W]; v.n.b = 0.f;
    v.s.a = plane[t.yb * W]; v.s.b = 0.f;
  }
  return v;
}

template <bool W2>
__device__ __forceinline__ void select_taps(const TapVals &v, const Taps &t, float &a, float &b, float &c, float &d) {
  const bool same = t.xi == t.xa;   // false only at xi = -1 (east = .a; also the one-pixel-wide image, where xa = 0) and xi = W-1 (west = .b)
  a = (t.yn & t.xw) ? (same ? v.n.a : v.n.b) : 0.f;
  b = (t.yn & t.xe) ? (same ? v.n.b : v.n.a) : 0.f;
  c = (t.ys & t.xw) ? (same ? v.s.a : v.s.b) : 0.f;
  d = (t.ys & t.xe) ? (same ? v.s.b : v.s.a) : 0.f;
}

__device__ __forceinline__ float blend(float a, float b, float c, float d, const Taps &t) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, t.nw), __fmul_rn(b, t.ne)), __fmul_rn(c, t.sw)), __fmul_rn(d, t.se));
}

template <bool W2>
__device__ __forceinline__ void load_taps(const float *plane, const Taps &t, int W, float &a, float &b, float &c, float &d) {
  const TapVals v = fetch_taps<W2>(plane, t, W);
  select_taps<W2>(v, t, a, b, c, d);
}

// torch.linspace(-1,1,n)[i] with the step precomputed on the host (same IEEE division), see base_coord()
__device__ __forceinline__ float base_coord_step(int i, int n, float step, bool align_corners) {
  if (n <= 1) return 0.f;
  float r = (i < n / 2) ? __fadd_rn(-1.f, __fmul_rn(step, (float)i)) : __fsub_rn(1.f, __fmul_rn(step, (float)(n - 1 - i)));
  if (!align_corners) r = __fdiv_rn(__fmul_rn(r, (float)(n - 1)), (float)n);
  return r;
}

template <bool HAS_THETA>
__device__ __forceinline__ void sample_pos(const float *grid, const float *theta, int n, int y, int x, int H, int W,
                                           float stepx, float stepy, bool ac, float &ix, float &iy, float &xs, float &ys) {
  float gx, gy;
  if (HAS_THETA) {
    const float *th = theta + (int64_t)n * 6;
    xs = base_coord_step(x, W, stepx, ac);
    ys = base_coord_step(y, H, stepy, ac);
    gx = __fadd_rn(__fadd_rn(__fmul_rn(th[0], xs), __fmul_rn(th[1], ys)), th[2]);
    gy = __fadd_rn(__fadd_rn(__fmul_rn(th[3], xs), __fmul_rn(th[4], ys)), th[5]);
  } else {
    const float2 g = reinterpret_cast<const float2 *>(grid)[((int64_t)n * H + y) * W + x];
    gx = g.x; gy = g.y; xs = 0.f; ys = 0.f;
  }
  ix = unnormalise(gx, W, ac);
  iy = unnormalise(gy, H, ac);
}

constexpr int FWD_MAXC = 3;             // channels whose taps are fetched together (straight-line loads)

// 2-D grid (chunk of 256 consecutive pixels of the flattened image, image), block 256: one output pixel per lane,
// straight-line code.
// Order: position -> ALL tap loads of the pixel (visibility + up to three channels, unconditional, clamped addresses)
// -> selects + blend -> stores.  No XCD remap (nothing is shared between blocks; round-robin placement keeps the
// eight XCDs streaming through the same DRAM pages) and no data-dependent branches: hipcc drains every outstanding
// load at a branch join, which costs a memory round trip (measured: 4.0 -> 4.9 TB/s on align_set at 256x256).
// VMODE and C3 (C == 3, the reference's frames) are template parameters for the same reason.
// Consecutive 256-pixel chunks per workgroup.  Measured (C3 / C4, cold): with a grid source (align_set: 8 loads + 4 stores per
// lane, ~90 vector instructions) two chunks per workgroup are 7 % faster than one (0.0664 vs 0.0718 ms: the waves are so short
// that their launch shows), four and eight are not; with theta (240 vector instructions per pixel) two / four / eight are
// 13-20 % SLOWER (the iterations of a wave do not overlap, and the kernel is bound by vector issue, not by dispatch).
template <bool HAS_THETA> struct FwdIters { static constexpr int value = HAS_THETA ? 1 : 2; };

template <bool HAS_THETA, bool W2, int VMODE, bool C3, bool VMAP>
__device__ __forceinline__ void warp_fwd_body(const WarpArgs &a, int n, int chunk);

template <bool HAS_THETA, bool W2, int VMODE, bool C3, bool VMAP = false>
__global__ void __launch_bounds__(256) warp_fwd_kernel(WarpArgs a) {
  const int nchunk = (a.H * a.W + 255) >> 8;
  int n, cgroup;
  constexpr int ITERS = FwdIters<HAS_THETA>::value;
  if (!warp_block((nchunk + ITERS - 1) / ITERS, a.gf, a.NB * a.NF, n, cgroup)) return;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int chunk = cgroup * ITERS + it;
    if (chunk < nchunk) warp_fwd_body<HAS_THETA, W2, VMODE, C3, VMAP>(a, n, chunk);
  }
}

// The value lane + 1 holds (DPP wave shift: a register-to-register move, no LDS); lane 63 gets 0.
__device__ __forceinline__ int lane_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_next(float v) { return __int_as_float(lane_next(__float_as_int(v))); }

// The taps of one plane: west column of the north / south row by one 4-byte load each; the EAST column comes from lane + 1
// when that lane's west taps are this lane's east taps (same two rows, next column: the common case of a smooth flow, where
// neighbouring output pixels sample neighbouring source columns), else from a second load that only the remaining lanes
// issue (lane 63, the lanes where the sampled row changes, columns an upscaling flow skips).  Against the two unaligned
// 8-byte gathers per plane this halves the bytes the vector L1 / texture addresser returns per pixel; the values are the
// same memory words either way, so results are bit-identical.
struct ShareTaps {
  unsigned on_w, os_w, on_e, os_e;   // element offsets inside a plane: (north | south row, west | east column), always inside the image
  bool share;
};
__device__ __forceinline__ ShareTaps make_share(const Taps &t, int W, bool live) {
  ShareTaps s;
  const int cw = max(0, min(t.xi, W - 1)), ce = max(0, min(t.xi + 1, W - 1));
  s.on_w = (unsigned)(t.ya * W + cw); s.os_w = (unsigned)(t.yb * W + cw);
  s.on_e = (unsigned)(t.ya * W + ce); s.os_e = (unsigned)(t.yb * W + ce);
  const unsigned nn = (unsigned)lane_next((int)s.on_w), ns = (unsigned)lane_next((int)s.os_w);
  s.share = ((threadIdx.x & 63) < 63) & (nn == s.on_e) & (ns == s.os_e);
  (void)live;
  return s;
}
struct PlaneTaps { float nw, ne, sw, se; };

template <bool HAS_THETA, bool W2, int VMODE, bool C3, bool VMAP>
__device__ __forceinline__ void warp_fwd_body(const WarpArgs &a, int n, int chunk) {
  const int H = a.H, W = a.W;
  const int b = (int)a.fd_nf.div((unsigned)n), f = n - b * a.NF;
  const int pix0 = chunk * 256 + threadIdx.x;   // flattened: every wave reads / writes 256 contiguous, aligned bytes per plane
  const bool ok = pix0 < H * W;
  const int pix = ok ? pix0 : H * W - 1;        // lanes past the image repeat its last pixel (they take part in the lane exchanges), store nothing
  const int y = (int)a.fd_w.div((unsigned)pix), x = pix - y * W;
  const bool ac = a.align_corners != 0;
  float gx, gy;
  if (HAS_THETA) {
    const float *th = a.theta + (int64_t)n * 6;
    const float xs = base_coord_step(x, W, a.stepx, ac), ys = base_coord_step(y, H, a.stepy, ac);
    gx = __fadd_rn(__fadd_rn(__fmul_rn(th[0], xs), __fmul_rn(th[1], ys)), th[2]);
    gy = __fadd_rn(__fadd_rn(__fmul_rn(th[3], xs), __fmul_rn(th[4], ys)), th[5]);
  } else {
    const float2 g = reinterpret_cast<const float2 *>(a.grid)[(int64_t)n * H * W + pix];
    gx = g.x; gy = g.y;
  }
  const float ix = unnormalise(gx, W, ac), iy = unnormalise(gy, H, ac);
  const Taps t = make_taps(ix, iy, H, W);
  const float *vp = a.v + b * a.v_sb + f * a.v_sf;
  const float *xp = a.x + b * a.x_sb + f * a.x_sf;
  float *op = a.xo + b * a.xo_sb + f * a.xo_sf + pix;
  const bool flip = a.v_flip != 0;
  float vtgt = 0.f;
  if (VMAP) vtgt = a.vt[b * a.vt_sb + pix];   // unconditional for this instantiation: in flight with the taps
  // visibility value -> v_out (and v_map = clamp(v - v_target, 0, 1), v_target = 1 - hole mask when vt_flip)
  auto put_v = [&](float vr) {
    st_stream(a.vo + b * a.vo_sb + f * a.vo_sf + pix, vr);
    if (VMAP) {
      const float vt1 = a.vt_flip ? __fsub_rn(1.f, vtgt) : vtgt;
      st_stream(a.vm + b * a.vm_sb + f * a.vm_sf + pix, fminf(fmaxf(__fsub_rn(vr, vt1), 0.f), 1.f));
    }
  };
  // Measured (round 3, cache-cold legs, A/B in the same runs): align_set (grid source, nearest visibility) 65.8 -> 63.9 us with
  // the shared taps; the affine forward (theta source, bilinear visibility: 8 west + 8 masked east loads instead of 8 gathers)
  // 0.261 -> 0.308 ms, i.e. that kernel pays per vector-memory INSTRUCTION, not per byte the L1 returns.  So: grid sources only.
  if (C3 && !HAS_THETA) {
    // ---- west taps of every plane (unconditional), then the east taps of the lanes that cannot take them from lane + 1
    constexpr bool VB = VMODE == TCL_VMODE_BILINEAR_THRESHOLD;
    constexpr int NP = VB ? 4 : 3;
    const ShareTaps s = make_share(t, W, ok);
    const float *pl[4] = {xp, xp + a.x_sc, xp + 2 * a.x_sc, vp};
    float wn[NP], ws[NP], en[NP], es[NP];
    float vnear = 0.f;
    bool inb = false;
    if (VMODE == TCL_VMODE_NEAREST) {
      const float xn = rintf(ix), yn = rintf(iy);  // v_rndne: half to even, as std::nearbyint
      inb = (xn >= 0.f) & (xn <= (float)(W - 1)) & (yn >= 0.f) & (yn <= (float)(H - 1));
      vnear = vp[(inb ? (int)yn : 0) * W + (inb ? (int)xn : 0)];   // unconditional load, then select
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) { wn[p] = pl[p][s.on_w]; ws[p] = pl[p][s.os_w]; en[p] = 0.f; es[p] = 0.f; }
    if (!s.share) {
#pragma unroll
      for (int p = 0; p < NP; ++p) { en[p] = pl[p][s.on_e]; es[p] = pl[p][s.os_e]; }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float nn = lane_next(wn[p]), ns = lane_next(ws[p]);
      en[p] = s.share ? nn : en[p];
      es[p] = s.share ? ns : es[p];
    }
    if (VMODE == TCL_VMODE_NEAREST) { if (ok) put_v(inb ? (flip ? __fsub_rn(1.f, vnear) : vnear) : 0.f); }
    else if (VB) {
      if (flip) { wn[NP - 1] = __fsub_rn(1.f, wn[NP - 1]); en[NP - 1] = __fsub_rn(1.f, en[NP - 1]); ws[NP - 1] = __fsub_rn(1.f, ws[NP - 1]); es[NP - 1] = __fsub_rn(1.f, es[NP - 1]); }
      const float vr = blend((t.yn & t.xw) ? wn[NP - 1] : 0.f, (t.yn & t.xe) ? en[NP - 1] : 0.f, (t.ys & t.xw) ? ws[NP - 1] : 0.f, (t.ys & t.xe) ? es[NP - 1] : 0.f, t);
      if (ok) put_v(vr > 0.5f ? 1.f : 0.f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float val = blend((t.yn & t.xw) ? wn[c] : 0.f, (t.yn & t.xe) ? en[c] : 0.f, (t.ys & t.xw) ? ws[c] : 0.f, (t.ys & t.xe) ? es[c] : 0.f, t);
      if (ok) st_stream(op + c * a.xo_sc, val);
    }
    return;
  }
  if (!ok) return;
  // ---- loads
  float vnear = 0.f;
  bool inb = false;
  TapVals tvv, tv[FWD_MAXC];
  if (VMODE == TCL_VMODE_NEAREST) {
    const float xn = rintf(ix), yn = rintf(iy);  // v_rndne: half to even, as std::nearbyint
    inb = (xn >= 0.f) & (xn <= (float)(W - 1)) & (yn >= 0.f) & (yn <= (float)(H - 1));
    vnear = vp[(inb ? (int)yn : 0) * W + (inb ? (int)xn : 0)];   // unconditional load, then select
  } else if (VMODE == TCL_VMODE_BILINEAR_THRESHOLD) {
    tvv = fetch_taps<W2>(vp, t, W);
  }
  if (C3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) tv[c] = fetch_taps<W2>(xp + c * a.x_sc, t, W);
  }
  // ---- visibility
  if (VMODE != TCL_VMODE_NONE) {
    float vr;
    if (VMODE == TCL_VMODE_NEAREST) vr = inb ? (flip ? __fsub_rn(1.f, vnear) : vnear) : 0.f;
    else {
      if (flip) { tvv.n.a = __fsub_rn(1.f, tvv.n.a); tvv.n.b = __fsub_rn(1.f, tvv.n.b); tvv.s.a = __fsub_rn(1.f, tvv.s.a); tvv.s.b = __fsub_rn(1.f, tvv.s.b); }
      float p, q, r, s;
      select_taps<W2>(tvv, t, p, q, r, s);
      vr = blend(p, q, r, s, t) > 0.5f ? 1.f : 0.f;
    }
    put_v(vr);
  }
  // ---- image channels
  if (C3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float p, q, r, s;
      select_taps<W2>(tv[c], t, p, q, r, s);
      st_stream(op + c * a.xo_sc, blend(p, q, r, s, t));
    }
  } else if (a.xo) {
    for (int c0 = 0; c0 < a.C; c0 += FWD_MAXC) {
#pragma unroll
      for (int c = 0; c < FWD_MAXC; ++c) tv[c] = fetch_taps<W2>(xp + min(c0 + c, a.C - 1) * a.x_sc, t, W);
#pragma unroll
      for (int c = 0; c < FWD_MAXC; ++c) {
        float p, q, r, s;
        select_taps<W2>(tv[c], t, p, q, r, s);
        const float val = blend(p, q, r, s, t);
        if (c0 + c < a.C) st_stream(op + (c0 + c) * a.xo_sc, val);
      }
    }
  }
}

struct WarpBwdArgs {
  const float *go; int64_t go_sb, go_sf, go_sc;
  const float *x;  int64_t x_sb, x_sf, x_sc;
  const float *grid, *theta;
  float *gx; int64_t gx_sb, gx_sf, gx_sc;
  float *g_grid;
  float *partial;  // (N, blocks_per_image, 6) when theta
  int NB, NF, C, H, W;
  int align_corners;
  float stepx, stepy;
  FastDiv fd_w, fd_nf; // / W (pixel -> row), / NF (image -> batch item)
};

// 2-D grid (chunk of 256 consecutive pixels of the flattened image, image), block 256; same straight-line structure as
// the forward kernel: position -> cotangents and taps of up to three channels in flight together -> gradients.
// HAS_GX adds the scatter-add of the input gradient (float atomics into a zeroed buffer).
template <bool HAS_THETA, bool HAS_GX, bool W2, bool C3>
__global__ void __launch_bounds__(256) warp_bwd_kernel(WarpBwdArgs a) {
  const int H = a.H, W = a.W, HW = H * W, nchunk = (HW + 255) >> 8;
  // the forward's (XCD, chunk, image / 8) order: one XCD walks one image, so the source rows that neighbouring chunks share are
  // fetched into ONE L2 (round 2 measured 1.28x the algorithmic bytes with the (chunk, image) grid, at the same speed)
  int n, chunk;
  if (!warp_block(nchunk, 1, a.NB * a.NF, n, chunk)) return;
  const int b = (int)a.fd_nf.div((unsigned)n), f = n - b * a.NF;
  const int pix0 = chunk * 256 + threadIdx.x;
  const bool ok = pix0 < HW;
  const int pix = ok ? pix0 : HW - 1;          // lanes past the image repeat its last pixel; nothing of theirs is stored
  const int y = (int)a.fd_w.div((unsigned)pix), x = pix - y * W;
  const bool ac = a.align_corners != 0;
  const float mx = ac ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
  const float my = ac ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
  const float *xp = a.x + b * a.x_sb + f * a.x_sf;
  const float *gp = a.go + b * a.go_sb + f * a.go_sf + pix;
  float ix, iy, xs, ys;
  sample_pos<HAS_THETA>(a.grid, a.theta, n, y, x, H, W, a.stepx, a.stepy, ac, ix, iy, xs, ys);
  const Taps t = make_taps(ix, iy, H, W);
  float gix = 0.f, giy = 0.f;
  // Input gradient: neighbouring output pixels of a smooth flow hit neighbouring source columns, so lane l's EAST taps and
  // lane l+1's WEST taps are usually the same two addresses.  When they are (same rows, column + 1, same validity), lane
  // l hands its east contributions to lane l+1, which adds them to its west ones: two float atomics per channel and pixel
  // instead of four.  Both lanes evaluate the same predicate on the same pair of tap descriptors.
  bool hand_east = false, take_west = false;
  if (HAS_GX) {
    const int key = ok ? ((t.xi & 0x3fff) | ((t.yi & 0x3fff) << 14) | (t.yn ? 1 << 28 : 0) | (t.ys ? 1 << 29 : 0)) : -1;
    const int kx = ok ? ((t.xw ? 1 : 0) | (t.xe ? 2 : 0)) : 0;
    const int lane = threadIdx.x & 63;
    const int key_up = __shfl_up(key, 1, 64), kx_up = __shfl_up(kx, 1, 64);       // lane - 1
    const int key_dn = __shfl_down(key, 1, 64), kx_dn = __shfl_down(kx, 1, 64);   // lane + 1
    // pair (p, q = p + 1): rows and row validity equal, xi_q == xi_p + 1 (no wrap at 14 bits: W < 16384 on this path),
    // east of p valid exactly when west of q is
    auto pair_ok = [](int kp, int xp, int kq, int xq) {
      return kp != -1 && kq != -1 && (kq - kp) == 1 && ((kp & 0x3fff) != 0x3fff) && (((xp >> 1) & 1) == (xq & 1));
    };
    const bool small = (W < 16384) & (H < 16384);   // the 14-bit key fields
    hand_east = small && lane < 63 && pair_ok(key, kx, key_dn, kx_dn);
    take_west = small && lane > 0 && pair_ok(key_up, kx_up, key, kx);
  }
  const int C = C3 ? 3 : a.C;   // C3: the reference's frames; the loop below then is one straight-line pass
  for (int c0 = 0; c0 < C; c0 += FWD_MAXC) {
    float g[FWD_MAXC];
    TapVals tv[FWD_MAXC];
#pragma unroll
    for (int c = 0; c < FWD_MAXC; ++c) {
      const int cc = min(c0 + c, C - 1);
      g[c] = gp[cc * a.go_sc];
      tv[c] = fetch_taps<W2>(xp + cc * a.x_sc, t, W);
    }
#pragma unroll
    for (int c = 0; c < FWD_MAXC; ++c) {
      float vnw, vne, vsw, vse;
      select_taps<W2>(tv[c], t, vnw, vne, vsw, vse);
      const float gc = (c0 + c < C) ? g[c] : 0.f;
      gix += gc * ((vne - vnw) * t.sy + (vse - vsw) * t.ny);
      giy += gc * ((vsw - vnw) * t.ex + (vse - vne) * t.wx);
      if (HAS_GX) {
        float cnw = gc * t.nw, csw = gc * t.sw;
        const float cne = gc * t.ne, cse = gc * t.se;
        const float ne_up = __shfl_up(cne, 1, 64), se_up = __shfl_up(cse, 1, 64);   // all lanes take part in the shuffles
        if (take_west) { cnw += ne_up; csw += se_up; }
        if (ok & (c0 + c < C)) {
          float *gplane = a.gx + b * a.gx_sb + f * a.gx_sf + (c0 + c) * a.gx_sc + t.yi * W + t.xi;
          if (t.xw & t.yn) atomicAdd(gplane, cnw);
          if (t.xw & t.ys) atomicAdd(gplane + W, csw);
          if (!hand_east) {
            if (t.xe & t.yn) atomicAdd(gplane + 1, cne);
            if (t.xe & t.ys) atomicAdd(gplane + W + 1, cse);
          }
        }
      }
    }
  }
  gix *= mx;
  giy *= my;
  if (HAS_THETA) {
    if (a.partial) {
      float acc[6] = {gix * xs, gix * ys, gix, giy * xs, giy * ys, giy};
      __shared__ float red[4][6];
      const int tid = threadIdx.x;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float s = wave_sum(ok ? acc[k] : 0.f);
        if ((tid & 63) == 0) red[tid >> 6][k] = s;
      }
      __syncthreads();
      if (tid < 6) a.partial[((int64_t)n * nchunk + chunk) * 6 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
  } else if (a.g_grid) {
    if (ok) st_stream2(a.g_grid + ((int64_t)n * HW + pix) * 2, gix, giy);
  }
}

// ------------------------------------------------------------------------------------------ tiled backward
// Backward over TILE_H x TILE_W output pixels per block: each of the four waves walks TILE_K consecutive rows of a
// 64-pixel-wide column strip, one column per lane.  Used for the two cases the one-pixel-per-lane kernel handles badly:
//   * the input gradient (scatter-add).  Float atomics execute at the memory side at one chip-wide rate, ~1.3 TB/s of added
//     bytes in 64-byte requests, whatever their placement (MI355X_MICROARCH.md, Global float atomics); four scattered adds per
//     pixel and channel are 4x the bytes of the gradient itself.  Under a smooth flow the four output pixels around a source
//     pixel are lane neighbours (east / west) and consecutive rows of one lane (north / south), so their contributions are
//     summed in registers first: a lane keeps the SOUTH-row contributions of its pixel pending and adds them to the NORTH-row
//     contributions of its next pixel when that one lands exactly one source row lower at the same column (otherwise the
//     pending pair is flushed as it is); east / west neighbours meet in emit_row_aligned, whose lanes own aligned source
//     addresses.  What leaves is ~(1 + 1/TILE_K) adds per source pixel and channel in 64-byte aligned requests.  Any flow is
//     handled: where the merges do not apply the adds simply go out unmerged.
//   * the theta gradient: six block sums per 2048 pixels instead of per 256 (the reductions, not HBM, bound the small kernel).
// The block's four waves: TILE_WX side by side x TILE_WY stacked.  Measured (round 3, cold legs): four side by side (8 rows x 256
// columns per block: 1 KB contiguous per plane row instead of 256 B) change nothing for the scatter-add (0.160 vs 0.163 ms) and
// cost the theta gradient 7 % at W = 424 (idle lanes in the second tile of a row): the stacked arrangement stays.
constexpr int TILE_WX = 1, TILE_WY = 4 / TILE_WX;
constexpr int TILE_K = 8, TILE_BW = 64 * TILE_WX, TILE_BH = TILE_K * TILE_WY;   // block tile in pixels
constexpr int TILE_MIN_W = 64, TILE_MIN_H = 32;                                  // smaller images stay on the one-pixel-per-lane kernel

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

// One row pair (source row r, columns xi and xi + 1) of contributions per lane -> global float atomics whose lanes OWN
// consecutive, 64-byte aligned source addresses.
//
// Why: a float-atomic wave instruction leaves L2 as one memory-side request per 64-byte segment and per group of 16 lanes; the
// chip retires ~20 G such requests per second (1.3 TB/s, MI355X_MICROARCH.md "Global float atomics"), and that rate is what
// bounds this kernel (measured: 4.46 M requests per C3 launch = 9 per instruction when every lane adds at its own tap address:
// 16 unaligned lanes always straddle two segments).  Here each lane pushes its west / east value to the lane that owns the
// address (ds_permute: a lane-to-lane move, no LDS memory), the owner adds what it received, and ONE instruction per channel
// adds 63 consecutive floats starting at a multiple of 16: four requests.  East / west neighbours meet in the owner's
// sum, so no separate hand-over is needed.  Lanes whose address lies beyond the 63-float window take a second round with
// a window of their own (rows of an upscaling or rotated flow); whatever is left after MAX_ROUNDS, and the losers of a push
// collision (two lanes with the same tap: downscaling flows), add at their own address as before.  Any flow is handled.
// Every lane of the wave must call.  `on`: this lane has a pair; vw / ve: its west / east column lies inside the image.
template <int NC>
__device__ __forceinline__ void emit_row_aligned(float *gbase, int64_t sc, int W, bool on, int r, int xi, bool vw, bool ve,
                                                 float (&w)[NC], float (&e)[NC], int nc) {
  constexpr int MAX_ROUNDS = 3;
  const int lane = threadIdx.x & 63;
  const int off = r * W + xi;   // offset of the west tap (-1 at r = 0, xi = -1: vw is false then)
  bool todo = on & (vw | ve);
  // Downscaling flows: two neighbouring lanes with the SAME tap.  The second lane of such a pair takes the first one's values
  // (one pass: pairs only; the third lane of a longer run loses the push below and adds at its own address).
  {
    // the key is the PAIR (r, xi), not the flat offset: (r, -1) and (r - 1, W - 1) share an offset but are different taps
    // (one has only an east column, the other only a west one); r, xi + 1 < 2^14 on this path
    const int key = todo ? ((r << 15) | (xi + 1)) : 0x7fffffff - lane;   // distinct for idle lanes
    const int k1 = __shfl_up(key, 1, 64), k2 = __shfl_up(key, 2, 64), kd = __shfl_down(key, 1, 64);
    const bool take = todo & (lane >= 1) & (k1 == key) & ((lane < 2) | (k2 != key));   // lane - 1 heads a run that I continue
    const bool give = todo & (lane < 63) & (kd == key) & ((lane < 1) | (k1 != key));   // I head a run and lane + 1 takes me
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float w1 = __shfl_up(w[c], 1, 64), e1 = __shfl_up(e[c], 1, 64);
      w[c] += take ? w1 : 0.f;
      e[c] += take ? e1 : 0.f;
    }
    todo = todo & !give;
  }
#pragma unroll 1
  for (int round = 0; round < MAX_ROUNDS && __any(todo); ++round) {
    const int base = wave_min_i(todo ? off : 0x7fffffff) & ~15;
    const int dw = off - base;
    const bool part = todo & (dw <= 61);        // both taps inside lanes 0 .. 62 (lane 63 is the dump of the lanes that do not push)
    const bool pw = part & vw, pe = part & ve;
    const int aw = (pw ? dw : 63) << 2, ae = (pe ? dw + 1 : 63) << 2;
    const int idw = __builtin_amdgcn_ds_permute(aw, lane + 1), ide = __builtin_amdgcn_ds_permute(ae, lane + 1);   // 0 = nobody pushed here
    const bool lostw = pw & (__builtin_amdgcn_ds_bpermute(aw, idw) != lane + 1);   // another lane with the same tap won the slot
    const bool loste = pe & (__builtin_amdgcn_ds_bpermute(ae, ide) != lane + 1);
    const bool recv = (lane < 63) & ((idw | ide) != 0);
    const bool any_lost = __any(lostw | loste);
    float *g = gbase + base + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float rw = __int_as_float(__builtin_amdgcn_ds_permute(aw, __float_as_int(w[c])));
      const float re = __int_as_float(__builtin_amdgcn_ds_permute(ae, __float_as_int(e[c])));
      if (c < nc) {
        if (recv) atomicAdd(g + c * sc, (idw != 0 ? rw : 0.f) + (ide != 0 ? re : 0.f));
        if (any_lost) {   // wave-uniform: no instruction at all in the common case
          if (lostw) atomicAdd(gbase + off + c * sc, w[c]);
          if (loste) atomicAdd(gbase + off + 1 + c * sc, e[c]);
        }
      }
    }
    todo = todo & !part;
  }
  if (__any(todo)) {   // wild flows: the remaining lanes add at their own addresses
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (todo && c < nc) {
        if (vw) atomicAdd(gbase + off + c * sc, w[c]);
        if (ve) atomicAdd(gbase + off + 1 + c * sc, e[c]);
      }
  }
}

struct BwdRow {   // one output row of a lane in the tiled backward: taps, cotangents and gathered values of a channel group
  Taps t;
  int pix;
  bool ok;
  float xs, ys, g[FWD_MAXC];
  TapVals tv[FWD_MAXC];
};

template <bool HAS_THETA, bool HAS_GX, bool C3>
__global__ void __launch_bounds__(256) warp_bwd_tile_kernel(WarpBwdArgs a, int tiles_x, int tiles_y) {
  __shared__ float red[4][6];
  const int H = a.H, W = a.W, HW = H * W;
  int n, tile;   // (XCD, tile, image / 8): an image's tiles, which share source rows and columns, stay on one XCD's L2
  if (!warp_block(tiles_x * tiles_y, 1, a.NB * a.NF, n, tile)) return;
  const int trow = tile / tiles_x;
  const int lx = threadIdx.x & 63, ly = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wy = ly / TILE_WX, wx = ly - wy * TILE_WX;
  const int ty0 = trow * TILE_BH + wy * TILE_K, tx0 = (tile - trow * tiles_x) * TILE_BW + wx * 64;   // this wave's strip: rows ty0 .. ty0 + TILE_K - 1
  const int b = (int)a.fd_nf.div((unsigned)n), f = n - b * a.NF;
  const bool ac = a.align_corners != 0;
  const float mx = ac ? 0.5f * (float)(W - 1) : 0.5f * (float)W;
  const float my = ac ? 0.5f * (float)(H - 1) : 0.5f * (float)H;
  const float *xp = a.x + b * a.x_sb + f * a.x_sf;
  const float *gp = a.go + b * a.go_sb + f * a.go_sf;
  float *gxb = HAS_GX ? a.gx + b * a.gx_sb + f * a.gx_sf : nullptr;
  const int x = min(tx0 + lx, W - 1);
  const bool okx = tx0 + lx < W;
  const int C = C3 ? 3 : a.C;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // theta source: the base-grid coordinate of the column (two IEEE divisions for align_corners = False) and theta[0] * xs once
  // per lane, the row's coordinate and theta[1] * ys once per block (LDS table, read as a broadcast).  The products enter the
  // same additions in the same order as in sample_pos: bit-identical positions.
  __shared__ float4 ytab[HAS_THETA ? TILE_BH : 1];
  float xs_l = 0.f, ax = 0.f, dx = 0.f, th2 = 0.f, th5 = 0.f;
  if (HAS_THETA) {
    const float *th = a.theta + (int64_t)n * 6;
    if (threadIdx.x < TILE_BH) {
      const float ysr = base_coord_step(min(trow * TILE_BH + (int)threadIdx.x, H - 1), H, a.stepy, ac);
      ytab[threadIdx.x] = make_float4(__fmul_rn(th[1], ysr), __fmul_rn(th[4], ysr), ysr, 0.f);
    }
    xs_l = base_coord_step(x, W, a.stepx, ac);
    ax = __fmul_rn(th[0], xs_l); dx = __fmul_rn(th[3], xs_l); th2 = th[2]; th5 = th[5];
    __syncthreads();
  }
  for (int c0 = 0; c0 < C; c0 += FWD_MAXC) {   // one pass for the reference's three-channel frames
    const int nc = min(FWD_MAXC, C - c0);
    // pending south-row pair of the previous pixel of this lane
    bool p_on = false, p_vw = false, p_ve = false;
    int p_r = 0, p_xi = 0;
    float p_w[FWD_MAXC], p_e[FWD_MAXC];
#pragma unroll
    for (int c = 0; c < FWD_MAXC; ++c) p_w[c] = p_e[c] = 0.f;
    // Software-pipelined by one row: the loads of row k+1 are issued BEFORE the atomics / stores of row k.  vmcnt counts loads,
    // stores and atomics in issue order, so a wave that adds and then loads waits for its own atomics (which complete at the
    // memory side, thousands of cycles under load) before it can use the next row's loads.
    auto fetch = [&](int k, BwdRow &r) {
      const int yy = ty0 + k;   // wave-uniform
      r.ok = okx & (yy < H);
      const int y = min(yy, H - 1);
      r.pix = y * W + x;
      float ix, iy;
      if (HAS_THETA) {
        const float4 yt = ytab[min(wy * TILE_K + k, TILE_BH - 1)];
        r.xs = xs_l; r.ys = yt.z;
        ix = unnormalise(__fadd_rn(__fadd_rn(ax, yt.x), th2), W, ac);
        iy = unnormalise(__fadd_rn(__fadd_rn(dx, yt.y), th5), H, ac);
      } else sample_pos<false>(a.grid, a.theta, n, y, x, H, W, a.stepx, a.stepy, ac, ix, iy, r.xs, r.ys);
      r.t = make_taps(ix, iy, H, W);
#pragma unroll
      for (int c = 0; c < FWD_MAXC; ++c) {
        const int cc = min(c0 + c, C - 1);
        r.g[c] = gp[cc * a.go_sc + r.pix];
        r.tv[c] = fetch_taps<true>(xp + cc * a.x_sc, r.t, W);
      }
    };
    auto process = [&](const BwdRow &r) {
      const Taps &t = r.t;
      const bool ok = r.ok;
      float gix = 0.f, giy = 0.f, n_w[FWD_MAXC], n_e[FWD_MAXC], s_w[FWD_MAXC], s_e[FWD_MAXC];
#pragma unroll
      for (int c = 0; c < FWD_MAXC; ++c) {
        float vnw, vne, vsw, vse;
        select_taps<true>(r.tv[c], t, vnw, vne, vsw, vse);
        const float gc = (ok & (c < nc)) ? r.g[c] : 0.f;
        gix += gc * ((vne - vnw) * t.sy + (vse - vsw) * t.ny);
        giy += gc * ((vsw - vnw) * t.ex + (vse - vne) * t.wx);
        n_w[c] = gc * t.nw; n_e[c] = gc * t.ne; s_w[c] = gc * t.sw; s_e[c] = gc * t.se;
      }
      if (HAS_GX) {
        const bool cols = t.xw | t.xe;
        // the pending pair sits on source row p_r at columns p_xi, p_xi + 1: it merges into this pixel's north pair when that
        // is the same row and column (validity of the row and of the columns is then the same on both sides)
        const bool merge = p_on & ok & t.yn & cols & (t.yi == p_r) & (t.xi == p_xi);
        if (__any(p_on & !merge)) emit_row_aligned<FWD_MAXC>(gxb + c0 * a.gx_sc, a.gx_sc, W, p_on & !merge, p_r, p_xi, p_vw, p_ve, p_w, p_e, nc);
#pragma unroll
        for (int c = 0; c < FWD_MAXC; ++c) {
          n_w[c] += merge ? p_w[c] : 0.f;
          n_e[c] += merge ? p_e[c] : 0.f;
        }
        emit_row_aligned<FWD_MAXC>(gxb + c0 * a.gx_sc, a.gx_sc, W, ok & t.yn & cols, t.yi, t.xi, t.xw, t.xe, n_w, n_e, nc);
        p_on = ok & t.ys & cols; p_r = t.yi + 1; p_xi = t.xi; p_vw = t.xw; p_ve = t.xe;
#pragma unroll
        for (int c = 0; c < FWD_MAXC; ++c) { p_w[c] = s_w[c]; p_e[c] = s_e[c]; }
      }
      gix *= mx; giy *= my;
      if (HAS_THETA) {
        if (ok) { acc[0] += gix * r.xs; acc[1] += gix * r.ys; acc[2] += gix; acc[3] += giy * r.xs; acc[4] += giy * r.ys; acc[5] += giy; }
      } else if (a.g_grid && ok) {
        float *o = a.g_grid + ((int64_t)n * HW + r.pix) * 2;
        if (C3) st_stream2(o, gix, giy);
        else if (c0 == 0) { o[0] = gix; o[1] = giy; }
        else { o[0] += gix; o[1] += giy; }   // generic channel counts: later groups add to this lane's own earlier store
      }
    };
    BwdRow row[2];
    fetch(0, row[0]);
#pragma unroll
    for (int k = 0; k < TILE_K; ++k) {
      if (k + 1 < TILE_K) fetch(k + 1, row[(k + 1) & 1]);   // rows past H: clamped loads, nothing emitted
      process(row[k & 1]);
    }
    if (HAS_GX && __any(p_on)) emit_row_aligned<FWD_MAXC>(gxb + c0 * a.gx_sc, a.gx_sc, W, p_on, p_r, p_xi, p_vw, p_ve, p_w, p_e, nc);
  }
  if (HAS_THETA && a.partial) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float s = wave_sum(acc[k]);
      if (lx == 0) red[ly][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6)
      a.partial[((int64_t)n * (tiles_x * tiles_y) + tile) * 6 + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// g_theta[n][k] = sum over the image's blocks, in block order (deterministic)
__global__ void warp_theta_reduce_kernel(const float *partial, float *g_theta, int N, int nblk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 6) return;
  const int n = i / 6, k = i - n * 6;
  float s = 0.f;
  for (int j = 0; j < nblk; ++j) s += partial[((int64_t)n * nblk + j) * 6 + k];
  g_theta[i] = s;
}

__global__ void vmap_kernel(const float *va, const float *vt, float *out, int64_t B, int64_t F, int64_t P) {
  const int64_t total = B * F * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (F * P), p = i % P;
    out[i] = fminf(fmaxf(va[i] - vt[b * P + p], 0.f), 1.f);
  }
}

// F.affine_grid(theta, (N,*,H,W), align_corners): the same positions sample_pos<true> derives on the fly, materialised
// (utils/movement.py:54-55 returns the inverse movement as a grid).  One float2 per lane, 2-D grid like the warp.
__global__ void __launch_bounds__(256) affine_grid_kernel(const float *__restrict__ theta, float2 *__restrict__ out, int H, int W,
                                                          float stepx, float stepy, int ac) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const float *th = theta + (int64_t)n * 6;
  const float xs = base_coord_step(x, W, stepx, ac), ys = base_coord_step(y, H, stepy, ac);
  st_stream2(reinterpret_cast<float *>(out + (int64_t)n * H * W + p), __fadd_rn(__fadd_rn(__fmul_rn(th[0], xs), __fmul_rn(th[1], ys)), th[2]),
             __fadd_rn(__fadd_rn(__fmul_rn(th[3], xs), __fmul_rn(th[4], ys)), th[5]));
}

static inline float lin_step(int64_t n) { return n > 1 ? 2.f / (float)(n - 1) : 0.f; }

}  // namespace tcl

using namespace tcl;

extern "C" int tcl_warp_fwd(const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc, const float *v, int64_t v_sb,
                            int64_t v_sf, const float *grid, const float *theta, float *x_out, int64_t xo_sb,
                            int64_t xo_sf, int64_t xo_sc, float *v_out, int64_t vo_sb, int64_t vo_sf, int64_t NB,
                            int64_t NF, int64_t C, int64_t H, int64_t W, int align_corners, int v_mode, void *stream) {
  return tcl_warp_fwd_fused(x, x_sb, x_sf, x_sc, v, v_sb, v_sf, 0, grid, theta, x_out, xo_sb, xo_sf, xo_sc, v_out, vo_sb, vo_sf, nullptr, 0, 0,
                            nullptr, 0, 0, NB, NF, C, H, W, align_corners, v_mode, stream);
}

extern "C" int tcl_warp_fwd_fused(const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc, const float *v, int64_t v_sb,
                                  int64_t v_sf, int v_is_hole_mask, const float *grid, const float *theta, float *x_out, int64_t xo_sb,
                                  int64_t xo_sf, int64_t xo_sc, float *v_out, int64_t vo_sb, int64_t vo_sf,
                                  const float *v_target, int64_t vt_sb, int vt_is_hole_mask, float *v_map, int64_t vm_sb, int64_t vm_sf,
                                  int64_t NB, int64_t NF, int64_t C, int64_t H, int64_t W, int align_corners, int v_mode, void *stream) {
  TCL_REQUIRE(NB > 0 && NF > 0 && H > 0 && W > 0 && C >= 0, TCL_ERR_SHAPE, "tcl_warp_fwd: bad shape NB=%ld NF=%ld C=%ld H=%ld W=%ld",
              (long)NB, (long)NF, (long)C, (long)H, (long)W);
  TCL_REQUIRE((grid != nullptr) != (theta != nullptr), TCL_ERR_ARG, "tcl_warp_fwd: exactly one of grid / theta must be given");
  TCL_REQUIRE(v_mode >= TCL_VMODE_NONE && v_mode <= TCL_VMODE_BILINEAR_THRESHOLD, TCL_ERR_ARG, "tcl_warp_fwd: unknown v_mode %d", v_mode);
  TCL_REQUIRE(v_mode == TCL_VMODE_NONE || (v && v_out), TCL_ERR_ARG, "tcl_warp_fwd: v / v_out missing for v_mode %d", v_mode);
  TCL_REQUIRE((x_out == nullptr) || (x != nullptr && C > 0), TCL_ERR_ARG, "tcl_warp_fwd: x_out given without x");
  TCL_REQUIRE(x_out || v_mode != TCL_VMODE_NONE, TCL_ERR_ARG, "tcl_warp_fwd: nothing to compute");
  TCL_REQUIRE(!grid || aligned(grid, 8), TCL_ERR_ARG, "tcl_warp_fwd: grid must be 8-byte aligned");
  // frames per XCD unit: NF only when every frame of a batch item samples ONE shared source frame (frame stride 0)
  const int64_t gf = (x_out && x_sf == 0 && (!v || v_sf == 0)) ? NF : 1;
  TCL_REQUIRE(H * W < (1LL << 30) && NB * NF < 65536 && cdiv(H * W, 256) * gf < 65536, TCL_ERR_SHAPE,
              "tcl_warp_fwd: too many images for one launch (%ld) or image too large", (long)(NB * NF));
  WarpArgs a{x, x_sb, x_sf, x_sc, v, v_sb, v_sf, grid, theta, x_out, xo_sb, xo_sf, xo_sc, v_out, vo_sb, vo_sf,
             (int)NB, (int)NF, (int)C, (int)H, (int)W, align_corners, v_mode, lin_step(W), lin_step(H), (int)gf,
             make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)NF), v_is_hole_mask != 0, vt_is_hole_mask != 0, v_target, vt_sb, v_map, vm_sb, vm_sf};
  TCL_REQUIRE((v_target != nullptr) == (v_map != nullptr), TCL_ERR_ARG, "tcl_warp_fwd_fused: v_target and v_map go together");
  TCL_REQUIRE(!v_map || (v_mode != TCL_VMODE_NONE && x_out != nullptr && C == 3 && W >= 2), TCL_ERR_UNSUPPORTED,
              "tcl_warp_fwd_fused: the fused v_map needs a visibility mode, 3-channel frames and W >= 2");
  const bool vmap = v_map != nullptr;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grd = warp_grid(cdiv(cdiv(H * W, 256), theta ? FwdIters<true>::value : FwdIters<false>::value), a.gf, NB * NF), blk(256);
  const bool c3 = x_out != nullptr && C == 3;
#define TCL_LAUNCH_WARP(TH, W2, VM)                                                                  \
  do {                                                                                                \
    if (vmap && W2 && VM != TCL_VMODE_NONE) hipLaunchKernelGGL((warp_fwd_kernel<TH, W2, VM, true, W2 && VM != TCL_VMODE_NONE>), grd, blk, 0, s, a); \
    else if (c3) hipLaunchKernelGGL((warp_fwd_kernel<TH, W2, VM, true>), grd, blk, 0, s, a);          \
    else hipLaunchKernelGGL((warp_fwd_kernel<TH, W2, VM, false>), grd, blk, 0, s, a);                 \
  } while (0)
#define TCL_LAUNCH_WARP_V(TH, W2)                                                                    \
  do {                                                                                                \
    if (v_mode == TCL_VMODE_NEAREST) TCL_LAUNCH_WARP(TH, W2, TCL_VMODE_NEAREST);                      \
    else if (v_mode == TCL_VMODE_BILINEAR_THRESHOLD) TCL_LAUNCH_WARP(TH, W2, TCL_VMODE_BILINEAR_THRESHOLD); \
    else TCL_LAUNCH_WARP(TH, W2, TCL_VMODE_NONE);                                                     \
  } while (0)
  if (theta) {
    if (W >= 2) TCL_LAUNCH_WARP_V(true, true);
    else TCL_LAUNCH_WARP_V(true, false);
  } else {
    if (W >= 2) TCL_LAUNCH_WARP_V(false, true);
    else TCL_LAUNCH_WARP_V(false, false);
  }
#undef TCL_LAUNCH_WARP_V
#undef TCL_LAUNCH_WARP
  return check_launch("tcl_warp_fwd");
}

extern "C" int tcl_affine_grid(const float *theta, float *grid, int64_t N, int64_t H, int64_t W, int align_corners, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && H * W < (1LL << 30) && N < 65536, TCL_ERR_SHAPE, "tcl_affine_grid: bad shape N=%ld H=%ld W=%ld", (long)N,
              (long)H, (long)W);
  TCL_REQUIRE(theta && grid && aligned(grid, 8), TCL_ERR_ARG, "tcl_affine_grid: NULL / misaligned pointer");
  hipLaunchKernelGGL(affine_grid_kernel, dim3((unsigned)cdiv(H * W, 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, theta, (float2 *)grid,
                     (int)H, (int)W, lin_step(W), lin_step(H), align_corners);
  return check_launch("tcl_affine_grid");
}

extern "C" size_t tcl_warp_bwd_workspace_bytes(int64_t NB, int64_t NF, int64_t H, int64_t W, int have_theta) {
  if (!have_theta || NB <= 0 || NF <= 0 || H <= 0 || W <= 0) return 0;
  // six partial sums per block and image: the larger of the two kernels' block counts (256-pixel chunks / tiles of TILE_BW x TILE_BH)
  const int64_t nblk = cdiv(H * W, 256) > cdiv(W, TILE_BW) * cdiv(H, TILE_BH) ? cdiv(H * W, 256) : cdiv(W, TILE_BW) * cdiv(H, TILE_BH);
  return round_up((size_t)(NB * NF) * nblk * 6 * sizeof(float), 256);
}

extern "C" int tcl_warp_bwd(const float *grad_x_out, int64_t go_sb, int64_t go_sf, int64_t go_sc, const float *x,
                            int64_t x_sb, int64_t x_sf, int64_t x_sc, const float *grid, const float *theta, float *g_x,
                            int64_t gx_sb, int64_t gx_sf, int64_t gx_sc, float *g_grid, float *g_theta, int64_t NB,
                            int64_t NF, int64_t C, int64_t H, int64_t W, int align_corners, void *workspace,
                            size_t workspace_bytes, void *stream) {
  TCL_REQUIRE(NB > 0 && NF > 0 && H > 0 && W > 0 && C > 0, TCL_ERR_SHAPE, "tcl_warp_bwd: bad shape");
  TCL_REQUIRE(grad_x_out && x, TCL_ERR_ARG, "tcl_warp_bwd: grad_x_out / x missing");
  TCL_REQUIRE((grid != nullptr) != (theta != nullptr), TCL_ERR_ARG, "tcl_warp_bwd: exactly one of grid / theta must be given");
  TCL_REQUIRE(!(grid && g_theta) && !(theta && g_grid), TCL_ERR_ARG, "tcl_warp_bwd: gradient kind does not match the forward's source");
  TCL_REQUIRE(!grid || aligned(grid, 8), TCL_ERR_ARG, "tcl_warp_bwd: grid must be 8-byte aligned");
  TCL_REQUIRE(!g_grid || aligned(g_grid, 8), TCL_ERR_ARG, "tcl_warp_bwd: g_grid must be 8-byte aligned");
  TCL_REQUIRE(H * W < (1LL << 30) && NB * NF < 65536 && cdiv(H * W, 256) < 65536, TCL_ERR_SHAPE,
              "tcl_warp_bwd: too many images for one launch (%ld) or image too large", (long)(NB * NF));
  if (!g_x && !g_grid && !g_theta) return TCL_OK;
  const int N = (int)(NB * NF);
  const int per = (int)cdiv(H * W, 256);
  float *partial = nullptr;
  if (g_theta) {
    const size_t need = tcl_warp_bwd_workspace_bytes(NB, NF, H, W, 1);
    TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
                "tcl_warp_bwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
    partial = (float *)workspace;
  }
  WarpBwdArgs a{grad_x_out, go_sb, go_sf, go_sc, x, x_sb, x_sf, x_sc, grid, theta, g_x, gx_sb, gx_sf, gx_sc,
                g_grid, partial, (int)NB, (int)NF, (int)C, (int)H, (int)W, align_corners, lin_step(W), lin_step(H),
                make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)NF)};
  hipStream_t s = (hipStream_t)stream;
  const dim3 grd = warp_grid(per, 1, N), blk(256);
#define TCL_LAUNCH_WBWD(TH, GX)                                                                   \
  do {                                                                                             \
    if (W >= 2 && C == 3) hipLaunchKernelGGL((warp_bwd_kernel<TH, GX, true, true>), grd, blk, 0, s, a);   \
    else if (W >= 2) hipLaunchKernelGGL((warp_bwd_kernel<TH, GX, true, false>), grd, blk, 0, s, a);       \
    else hipLaunchKernelGGL((warp_bwd_kernel<TH, GX, false, false>), grd, blk, 0, s, a);                  \
  } while (0)
  // images of at least one tile: the tiled kernel for the input gradient (LDS-staged scatter-add) and for the theta gradient
  // (one set of block sums per 2048 pixels); the grid-gradient-only case stays on the one-pixel-per-lane kernel
  const bool tiled = W >= TILE_MIN_W && H >= TILE_MIN_H && W < 16384 && H < 16384 && (g_x || theta);
  int nblk = per;
  if (tiled) {
    const int tx = (int)cdiv(W, TILE_BW), ty = (int)cdiv(H, TILE_BH);
    nblk = tx * ty;   // tcl_warp_bwd_workspace_bytes sizes the theta workspace for the larger block count
    const dim3 tg = warp_grid(nblk, 1, N);
#define TCL_LAUNCH_WTILE(TH, GX)                                                                          \
  do {                                                                                                     \
    if (C == 3) hipLaunchKernelGGL((warp_bwd_tile_kernel<TH, GX, true>), tg, blk, 0, s, a, tx, ty);        \
    else hipLaunchKernelGGL((warp_bwd_tile_kernel<TH, GX, false>), tg, blk, 0, s, a, tx, ty);              \
  } while (0)
    if (theta) {
      if (g_x) TCL_LAUNCH_WTILE(true, true);
      else TCL_LAUNCH_WTILE(true, false);
    } else if (g_x) TCL_LAUNCH_WTILE(false, true);
    else TCL_LAUNCH_WTILE(false, false);
#undef TCL_LAUNCH_WTILE
  } else if (theta) {
    if (g_x) TCL_LAUNCH_WBWD(true, true);
    else TCL_LAUNCH_WBWD(true, false);
  } else {
    if (g_x) TCL_LAUNCH_WBWD(false, true);
    else TCL_LAUNCH_WBWD(false, false);
  }
#undef TCL_LAUNCH_WBWD
  int rc = check_launch("tcl_warp_bwd");
  if (rc != TCL_OK) return rc;
  if (g_theta) {
    hipLaunchKernelGGL(warp_theta_reduce_kernel, dim3((unsigned)cdiv(N * 6, 256)), dim3(256), 0, s, partial, g_theta, N, nblk);
    rc = check_launch("tcl_warp_bwd(theta reduce)");
  }
  return rc;
}

extern "C" int tcl_vmap(const float *v_aligned, const float *v_target, float *out, int64_t B, int64_t F, int64_t P, void *stream) {
  TCL_REQUIRE(B > 0 && F > 0 && P > 0, TCL_ERR_SHAPE, "tcl_vmap: bad shape");
  TCL_REQUIRE(v_aligned && v_target && out, TCL_ERR_ARG, "tcl_vmap: NULL pointer");
  int64_t blocks = cdiv(B * F * P, 256);
  if (blocks > kMaxStreamBlocks) blocks = kMaxStreamBlocks;
  hipLaunchKernelGGL(vmap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, v_aligned, v_target, out, B, F, P);
  return check_launch("tcl_vmap");
}
