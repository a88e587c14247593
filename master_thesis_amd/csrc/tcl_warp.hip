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
};

__device__ __forceinline__ float unnormalise(float g, int size, bool align_corners) {
  if (align_corners) return __fmul_rn(__fmul_rn(__fadd_rn(g, 1.f), 0.5f), (float)(size - 1));
  return __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(g, 1.f), (float)size), 1.f), 0.5f);
}

struct Taps {
  float nw, ne, sw, se;      // weights
  float wx, ex, ny, sy;      // distances
  int o_nw, o_ne, o_sw, o_se;  // plane offsets (valid only where the flag is set)
  bool b_nw, b_ne, b_sw, b_se;
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
  const bool xw = (x0 >= 0.f) && (x0 <= (float)(W - 1));
  const bool xe = (x0 >= -1.f) && (x0 <= (float)(W - 2));
  const bool yn = (y0 >= 0.f) && (y0 <= (float)(H - 1));
  const bool ys = (y0 >= -1.f) && (y0 <= (float)(H - 2));
  const int xi = (xw || xe) ? (int)x0 : 0, yi = (yn || ys) ? (int)y0 : 0;
  t.b_nw = xw && yn; t.b_ne = xe && yn; t.b_sw = xw && ys; t.b_se = xe && ys;
  t.o_nw = yi * W + xi; t.o_ne = t.o_nw + 1; t.o_sw = t.o_nw + W; t.o_se = t.o_sw + 1;
  return t;
}

__device__ __forceinline__ float bilinear(const float *plane, const Taps &t) {
  const float a = t.b_nw ? plane[t.o_nw] : 0.f, b = t.b_ne ? plane[t.o_ne] : 0.f;
  const float c = t.b_sw ? plane[t.o_sw] : 0.f, d = t.b_se ? plane[t.o_se] : 0.f;
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(a, t.nw), __fmul_rn(b, t.ne)), __fmul_rn(c, t.sw)), __fmul_rn(d, t.se));
}

template <bool HAS_THETA>
__device__ __forceinline__ void sample_pos(const float *grid, const float *theta, int n, int y, int x, int H, int W,
                                           bool ac, float &ix, float &iy, float &xs, float &ys) {
  float gx, gy;
  if (HAS_THETA) {
    const float *th = theta + (int64_t)n * 6;
    xs = base_coord(x, W, ac);
    ys = base_coord(y, H, ac);
    gx = __fadd_rn(__fadd_rn(__fmul_rn(th[0], xs), __fmul_rn(th[1], ys)), th[2]);
    gy = __fadd_rn(__fadd_rn(__fmul_rn(th[3], xs), __fmul_rn(th[4], ys)), th[5]);
  } else {
    const float2 g = reinterpret_cast<const float2 *>(grid)[((int64_t)n * H + y) * W + x];
    gx = g.x; gy = g.y; xs = 0.f; ys = 0.f;
  }
  ix = unnormalise(gx, W, ac);
  iy = unnormalise(gy, H, ac);
}

template <bool HAS_THETA>
__global__ void __launch_bounds__(256) warp_fwd_kernel(WarpArgs a) {
  const int HW = a.H * a.W;
  const int64_t total = (int64_t)a.NB * a.NF * HW;
  const bool ac = a.align_corners != 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), pix = (int)(i - (int64_t)n * HW);
    const int y = pix / a.W, x = pix - y * a.W;
    const int b = n / a.NF, f = n - b * a.NF;
    float ix, iy, xs, ys;
    sample_pos<HAS_THETA>(a.grid, a.theta, n, y, x, a.H, a.W, ac, ix, iy, xs, ys);
    const Taps t = make_taps(ix, iy, a.H, a.W);
    if (a.xo) {
      const float *xp = a.x + b * a.x_sb + f * a.x_sf;
      float *op = a.xo + b * a.xo_sb + f * a.xo_sf + pix;
      for (int c = 0; c < a.C; ++c) op[c * a.xo_sc] = bilinear(xp + c * a.x_sc, t);
    }
    if (a.v_mode != TCL_VMODE_NONE) {
      const float *vp = a.v + b * a.v_sb + f * a.v_sf;
      float r;
      if (a.v_mode == TCL_VMODE_NEAREST) {
        const float xn = rintf(ix), yn = rintf(iy);  // v_rndne: half to even, as std::nearbyint
        const bool inb = (xn >= 0.f) && (xn <= (float)(a.W - 1)) && (yn >= 0.f) && (yn <= (float)(a.H - 1));
        r = inb ? vp[(int)yn * a.W + (int)xn] : 0.f;
      } else {
        r = bilinear(vp, t) > 0.5f ? 1.f : 0.f;
      }
      a.vo[b * a.vo_sb + f * a.vo_sf + pix] = r;
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
};

// grid = (blocks_per_image, NB*NF); every block owns a contiguous pixel range of one image.
template <bool HAS_THETA>
__global__ void __launch_bounds__(256) warp_bwd_kernel(WarpBwdArgs a) {
  const int HW = a.H * a.W;
  const int n = blockIdx.y;
  const int b = n / a.NF, f = n - b * a.NF;
  const bool ac = a.align_corners != 0;
  const float mx = ac ? 0.5f * (float)(a.W - 1) : 0.5f * (float)a.W;
  const float my = ac ? 0.5f * (float)(a.H - 1) : 0.5f * (float)a.H;
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float *xp = a.x + b * a.x_sb + f * a.x_sf;
  const float *gp = a.go + b * a.go_sb + f * a.go_sf;
  float *gxp = a.gx ? a.gx + b * a.gx_sb + f * a.gx_sf : nullptr;
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < HW; pix += gridDim.x * blockDim.x) {
    const int y = pix / a.W, x = pix - y * a.W;
    float ix, iy, xs, ys;
    sample_pos<HAS_THETA>(a.grid, a.theta, n, y, x, a.H, a.W, ac, ix, iy, xs, ys);
    const Taps t = make_taps(ix, iy, a.H, a.W);
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float g = gp[c * a.go_sc + pix];
      const float *plane = xp + c * a.x_sc;
      const float vnw = t.b_nw ? plane[t.o_nw] : 0.f, vne = t.b_ne ? plane[t.o_ne] : 0.f;
      const float vsw = t.b_sw ? plane[t.o_sw] : 0.f, vse = t.b_se ? plane[t.o_se] : 0.f;
      gix += g * ((vne - vnw) * t.sy + (vse - vsw) * t.ny);
      giy += g * ((vsw - vnw) * t.ex + (vse - vne) * t.wx);
      if (gxp) {
        float *gplane = gxp + c * a.gx_sc;
        if (t.b_nw) atomicAdd(gplane + t.o_nw, g * t.nw);
        if (t.b_ne) atomicAdd(gplane + t.o_ne, g * t.ne);
        if (t.b_sw) atomicAdd(gplane + t.o_sw, g * t.sw);
        if (t.b_se) atomicAdd(gplane + t.o_se, g * t.se);
      }
    }
    gix *= mx;
    giy *= my;
    if (HAS_THETA) {
      acc[0] += gix * xs; acc[1] += gix * ys; acc[2] += gix;
      acc[3] += giy * xs; acc[4] += giy * ys; acc[5] += giy;
    } else if (a.g_grid) {
      reinterpret_cast<float2 *>(a.g_grid)[(int64_t)n * HW + pix] = make_float2(gix, giy);
    }
  }
  if (HAS_THETA && a.partial) {
    __shared__ float red[4][6];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float s = wave_sum(acc[k]);
      if (lane == 0) red[wid][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      a.partial[((int64_t)n * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = s;
    }
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

static int bwd_blocks_per_image(int64_t HW, int64_t N) {
  int64_t per = cdiv(HW, 256);
  const int64_t cap = cdiv((int64_t)kMaxStreamBlocks, N);  // keep the whole grid near 8 blocks per CU
  if (per > cap) per = cap;
  if (per < 1) per = 1;
  return (int)per;
}

}  // namespace tcl

using namespace tcl;

extern "C" int tcl_warp_fwd(const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc, const float *v, int64_t v_sb,
                            int64_t v_sf, const float *grid, const float *theta, float *x_out, int64_t xo_sb,
                            int64_t xo_sf, int64_t xo_sc, float *v_out, int64_t vo_sb, int64_t vo_sf, int64_t NB,
                            int64_t NF, int64_t C, int64_t H, int64_t W, int align_corners, int v_mode, void *stream) {
  TCL_REQUIRE(NB > 0 && NF > 0 && H > 0 && W > 0 && C >= 0, TCL_ERR_SHAPE, "tcl_warp_fwd: bad shape NB=%ld NF=%ld C=%ld H=%ld W=%ld",
              (long)NB, (long)NF, (long)C, (long)H, (long)W);
  TCL_REQUIRE((grid != nullptr) != (theta != nullptr), TCL_ERR_ARG, "tcl_warp_fwd: exactly one of grid / theta must be given");
  TCL_REQUIRE(v_mode >= TCL_VMODE_NONE && v_mode <= TCL_VMODE_BILINEAR_THRESHOLD, TCL_ERR_ARG, "tcl_warp_fwd: unknown v_mode %d", v_mode);
  TCL_REQUIRE(v_mode == TCL_VMODE_NONE || (v && v_out), TCL_ERR_ARG, "tcl_warp_fwd: v / v_out missing for v_mode %d", v_mode);
  TCL_REQUIRE((x_out == nullptr) || (x != nullptr && C > 0), TCL_ERR_ARG, "tcl_warp_fwd: x_out given without x");
  TCL_REQUIRE(x_out || v_mode != TCL_VMODE_NONE, TCL_ERR_ARG, "tcl_warp_fwd: nothing to compute");
  TCL_REQUIRE(!grid || aligned(grid, 8), TCL_ERR_ARG, "tcl_warp_fwd: grid must be 8-byte aligned");
  TCL_REQUIRE(H * W < (1LL << 30) && NB * NF < (1LL << 30), TCL_ERR_SHAPE, "tcl_warp_fwd: image too large");
  WarpArgs a{x, x_sb, x_sf, x_sc, v, v_sb, v_sf, grid, theta, x_out, xo_sb, xo_sf, xo_sc, v_out, vo_sb, vo_sf,
             (int)NB, (int)NF, (int)C, (int)H, (int)W, align_corners, v_mode};
  const int64_t total = NB * NF * H * W;
  int64_t blocks = cdiv(total, 256);
  if (blocks > kMaxStreamBlocks * 4) blocks = kMaxStreamBlocks * 4;
  hipStream_t s = (hipStream_t)stream;
  if (theta)
    hipLaunchKernelGGL(warp_fwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(warp_fwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  return check_launch("tcl_warp_fwd");
}

extern "C" size_t tcl_warp_bwd_workspace_bytes(int64_t NB, int64_t NF, int64_t H, int64_t W, int have_theta) {
  if (!have_theta || NB <= 0 || NF <= 0 || H <= 0 || W <= 0) return 0;
  return round_up((size_t)(NB * NF) * bwd_blocks_per_image(H * W, NB * NF) * 6 * sizeof(float), 256);
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
  TCL_REQUIRE(H * W < (1LL << 30) && NB * NF < 65536, TCL_ERR_SHAPE, "tcl_warp_bwd: too many images for one launch (%ld)", (long)(NB * NF));
  if (!g_x && !g_grid && !g_theta) return TCL_OK;
  const int N = (int)(NB * NF);
  const int per = bwd_blocks_per_image(H * W, N);
  float *partial = nullptr;
  if (g_theta) {
    const size_t need = tcl_warp_bwd_workspace_bytes(NB, NF, H, W, 1);
    TCL_REQUIRE(workspace && aligned(workspace, 256) && workspace_bytes >= need, TCL_ERR_WORKSPACE,
                "tcl_warp_bwd: workspace of %zu bytes (256-aligned) needed, got %zu", need, workspace_bytes);
    partial = (float *)workspace;
  }
  WarpBwdArgs a{grad_x_out, go_sb, go_sf, go_sc, x, x_sb, x_sf, x_sc, grid, theta, g_x, gx_sb, gx_sf, gx_sc,
                g_grid, partial, (int)NB, (int)NF, (int)C, (int)H, (int)W, align_corners};
  hipStream_t s = (hipStream_t)stream;
  if (theta)
    hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3(per, N), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3(per, N), dim3(256), 0, s, a);
  int rc = check_launch("tcl_warp_bwd");
  if (rc != TCL_OK) return rc;
  if (g_theta) {
    hipLaunchKernelGGL(warp_theta_reduce_kernel, dim3((unsigned)cdiv(N * 6, 256)), dim3(256), 0, s, partial, g_theta, N, per);
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
