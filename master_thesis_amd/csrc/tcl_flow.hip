// Flow utilities of utils/flow.py:29-90 for gfx950: abs <-> relative grids, crop, legacy-nearest resize.
// Pure streaming kernels over (N,H,W,2) float2 grids; see include/tcl.h.
#include "tcl_common.h"

namespace tcl {

// out = flow + sign * identity, identity = affine_grid(I, align_corners=True) = (linspace_W[x], linspace_H[y])
__global__ void __launch_bounds__(256) flow_add_identity_kernel(const float2 *__restrict__ flow, float2 *__restrict__ out, int64_t N,
                                                                int H, int W, float sign) {
  const int64_t total = N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % ((int64_t)H * W)), y = r / W, x = r - y * W;
    const float2 f = flow[i];
    out[i] = make_float2(__fadd_rn(f.x, sign * base_coord(x, W, true)), __fadd_rn(f.y, sign * base_coord(y, H, true)));
  }
}

// crop_flow: rel = flow - id(H,W); window; rel.x *= W/cw, rel.y *= H/ch; out = rel + id(ch,cw)
__global__ void __launch_bounds__(256) crop_flow_kernel(const float2 *__restrict__ flow, float2 *__restrict__ out, int64_t N, int H, int W,
                                                        int ch, int cw, int py, int px, float fx, float fy) {
  const int64_t total = N * ch * cw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / ((int64_t)ch * cw);
    const int r = (int)(i - n * ch * cw), y = r / cw, x = r - y * cw;
    const float2 f = flow[(n * H + (py + y)) * W + (px + x)];
    const float rx = __fmul_rn(__fsub_rn(f.x, base_coord(px + x, W, true)), fx);
    const float ry = __fmul_rn(__fsub_rn(f.y, base_coord(py + y, H, true)), fy);
    out[i] = make_float2(__fadd_rn(rx, base_coord(x, cw, true)), __fadd_rn(ry, base_coord(y, ch, true)));
  }
}

// F.interpolate default (legacy nearest): src = min(floor(dst * in/out), in - 1)
__global__ void __launch_bounds__(256) resize_flow_kernel(const float2 *__restrict__ flow, float2 *__restrict__ out, int64_t N, int H, int W,
                                                          int oh, int ow) {
  const float sh = __fdiv_rn((float)H, (float)oh), sw = __fdiv_rn((float)W, (float)ow);
  const int64_t total = N * oh * ow;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / ((int64_t)oh * ow);
    const int r = (int)(i - n * oh * ow), y = r / ow, x = r - y * ow;
    const int sy = min((int)floorf(__fmul_rn((float)y, sh)), H - 1), sx = min((int)floorf(__fmul_rn((float)x, sw)), W - 1);
    out[i] = flow[(n * H + sy) * W + sx];
  }
}

static unsigned flow_blocks(int64_t total) {
  int64_t blocks = cdiv(total, 256);
  if (blocks > kMaxStreamBlocks * 2) blocks = kMaxStreamBlocks * 2;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace tcl

using namespace tcl;

extern "C" int tcl_flow_add_identity(const float *flow, float *out, int64_t N, int64_t H, int64_t W, int sign, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && H * W < (1LL << 31), TCL_ERR_SHAPE, "tcl_flow_add_identity: bad shape");
  TCL_REQUIRE(flow && out && aligned(flow, 8) && aligned(out, 8), TCL_ERR_ARG, "tcl_flow_add_identity: NULL / misaligned pointer");
  TCL_REQUIRE(sign == 1 || sign == -1, TCL_ERR_ARG, "tcl_flow_add_identity: sign must be +1 or -1");
  hipLaunchKernelGGL(flow_add_identity_kernel, dim3(flow_blocks(N * H * W)), dim3(256), 0, (hipStream_t)stream, (const float2 *)flow,
                     (float2 *)out, N, (int)H, (int)W, (float)sign);
  return check_launch("tcl_flow_add_identity");
}

extern "C" int tcl_crop_flow(const float *flow, float *out, int64_t N, int64_t H, int64_t W, int64_t ch, int64_t cw, int64_t py, int64_t px, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && ch > 0 && cw > 0 && py >= 0 && px >= 0 && py + ch <= H && px + cw <= W && H * W < (1LL << 31), TCL_ERR_SHAPE,
              "tcl_crop_flow: crop (%ld,%ld)+(%ld,%ld) outside (%ld,%ld)", (long)py, (long)px, (long)ch, (long)cw, (long)H, (long)W);
  TCL_REQUIRE(flow && out && aligned(flow, 8) && aligned(out, 8), TCL_ERR_ARG, "tcl_crop_flow: NULL / misaligned pointer");
  const float fx = (float)((double)W / (double)cw), fy = (float)((double)H / (double)ch);
  hipLaunchKernelGGL(crop_flow_kernel, dim3(flow_blocks(N * ch * cw)), dim3(256), 0, (hipStream_t)stream, (const float2 *)flow, (float2 *)out, N,
                     (int)H, (int)W, (int)ch, (int)cw, (int)py, (int)px, fx, fy);
  return check_launch("tcl_crop_flow");
}

extern "C" int tcl_resize_flow(const float *flow, float *out, int64_t N, int64_t H, int64_t W, int64_t oh, int64_t ow, void *stream) {
  TCL_REQUIRE(N > 0 && H > 0 && W > 0 && oh > 0 && ow > 0 && H * W < (1LL << 31) && oh * ow < (1LL << 31), TCL_ERR_SHAPE, "tcl_resize_flow: bad shape");
  TCL_REQUIRE(flow && out && aligned(flow, 8) && aligned(out, 8), TCL_ERR_ARG, "tcl_resize_flow: NULL / misaligned pointer");
  hipLaunchKernelGGL(resize_flow_kernel, dim3(flow_blocks(N * oh * ow)), dim3(256), 0, (hipStream_t)stream, (const float2 *)flow, (float2 *)out, N,
                     (int)H, (int)W, (int)oh, (int)ow);
  return check_launch("tcl_resize_flow");
}
