/*
 * tcl.h - C ABI of libtcl_hip.so: the MI355X (gfx950) implementation of the frame-alignment and
 * temporal-copying hot path of AlomdaElmasry/master_thesis.
 *
 * The reference has no FFI / plugin registry: its boundary for this path is a set of Python functions
 * (utils/correlation.py, utils/flow.py, models/cpn.py, models/cpn_original.py).  Each entry point below
 * names the reference function (file:line, relative to the reference root) whose arithmetic it replaces;
 * the Python shim in master_thesis_amd/ keeps those functions' names and signatures and calls through
 * this ABI with raw device pointers (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to float32 unless stated; tensors are dense row-major in the
 *    shape written next to them; "nullable" pointers may be NULL;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call only enqueues
 *    work on `stream`: it never synchronises, never allocates or frees device memory and keeps no
 *    mutable state that results depend on (re-entrant across streams and across the per-GPU processes).
 *    The only process-wide state is two per-device lookup caches, each filled once under a mutex: the
 *    kernel attribute + CU count of the wide contraction kernels (tcl_corr4d_bwd) and the device address
 *    of the static scratch array of tcl_ctx_fwd_static;
 *  - scratch space is a caller-provided `workspace` whose size comes from the matching
 *    *_workspace_bytes function; it must be 256-byte aligned;
 *  - return value: TCL_OK (0) or a TCL_ERR_* code; tcl_last_error() then returns a thread-local,
 *    human-readable message.  The Python shim turns non-zero into RuntimeError.
 *  - masks / visibility maps are float32 holding {0,1}; index and mask results are bit-exact with the
 *    reference, float results match to 1e-4 relative (TCL_PREC_F32), 1e-2 (TCL_PREC_BF16).
 */
#ifndef TCL_H_
#define TCL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TCL_VERSION 110

enum {
  TCL_OK = 0,
  TCL_ERR_SHAPE = 1,     /* a dimension is <= 0 or inconsistent                */
  TCL_ERR_ARG = 2,       /* NULL / misaligned pointer, unknown enum value      */
  TCL_ERR_WORKSPACE = 3, /* workspace missing, misaligned or too small         */
  TCL_ERR_LAUNCH = 4,    /* hipGetLastError() reported a launch failure        */
  TCL_ERR_UNSUPPORTED = 5
};

/* arithmetic of the all-pairs contraction (tcl_corr4d_*) */
enum {
  TCL_PREC_F32 = 0,    /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain                       */
  TCL_PREC_BF16 = 1,   /* operands rounded to bf16, fp32 accumulate (1e-2 bar)               */
  TCL_PREC_BF16X3 = 2  /* bf16 hi/lo split, 3 bf16 MFMAs per product (fp32-class, ~1e-5 abs) */
};

/* context-matching variants (tcl_ctx_*) */
enum {
  TCL_CTX_DEFAULT = 0,   /* models/cpn.py:94-135  CPNContextMatchingDefault   */
  TCL_CTX_COMPLETE = 1,  /* models/cpn.py:142-195 CPNContextMatchingComplete  */
  TCL_CTX_CM_MODULE = 2  /* models/cpn_original.py:201-243 CM_Module          */
};

/* how the 1-channel visibility map is sampled by tcl_warp_fwd */
enum {
  TCL_VMODE_NONE = 0,
  TCL_VMODE_NEAREST = 1,           /* utils/flow.py:98-100 (align_set)                  */
  TCL_VMODE_BILINEAR_THRESHOLD = 2 /* models/cpn.py:288-290: bilinear then (> 0.5)      */
};

int tcl_version(void);
const char *tcl_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * A1  all-pairs masked feature correlation.  Replaces compute_masked_4d_correlation,
 *     utils/correlation.py:28-54.
 *   x_t (B,C,P)  v_t (B,P) nullable   x_r (B,C,F,P)  v_r (B,F,P) nullable   P = H*W
 *   out (B,F,P,P): out[b,f,p,q] = <a^[b,:,p], r^[b,f,:,q]>, a^ = x v / (||x v||_2 + 1e-9).
 *   The forward leaves the normalised operands and the norms in `workspace` (TCL_PREC_F32: fp32 images; the bf16 variants:
 *   the bf16 hi (and lo) images that their forward AND backward products read, and no fp32 image); pass the same buffer,
 *   unmodified and with the same `precision`, to tcl_corr4d_bwd as `fwd_workspace`.
 * ------------------------------------------------------------------------------------------- */
size_t tcl_corr4d_fwd_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision);
int tcl_corr4d_fwd(const float *x_t, const float *v_t, const float *x_r, const float *v_r, float *out,
                   int64_t B, int64_t C, int64_t F, int64_t P, int precision,
                   void *workspace, size_t workspace_bytes, void *stream);

/*   The contraction step alone: out = a^T r^ from the normalised operands a previous tcl_corr4d_fwd left in
 *   `fwd_workspace` (same B,C,F,P,precision).  Lets a caller re-materialise the volume without re-normalising. */
int tcl_corr4d_contract(const void *fwd_workspace, float *out, int64_t B, int64_t C, int64_t F, int64_t P,
                        int precision, void *stream);

/*   grad_out (B,F,P,P) -> g_x_t (B,C,P), g_x_r (B,C,F,P) (either may be NULL to skip it).
 *   Autograd of the same reference function (the reference never differentiates it; the oracle is
 *   torch.autograd over it, see tests/golden/make_golden.py). */
size_t tcl_corr4d_bwd_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision);
int tcl_corr4d_bwd(const float *grad_out, const float *x_t, const float *v_t, const float *x_r, const float *v_r,
                   const void *fwd_workspace, float *g_x_t, float *g_x_r,
                   int64_t B, int64_t C, int64_t F, int64_t P, int precision,
                   void *workspace, size_t workspace_bytes, void *stream);

/* A1 + A10 fused: corr_to_flow(compute_masked_4d_correlation(...)) without the volume in HBM.
 *   flow (B,F,P,2), utils/flow.py:8-26 on top of utils/correlation.py:28-54.  S = side of the square map (P = S*S).
 *   `precision` as in tcl_corr4d_fwd: with TCL_PREC_BF16 / BF16X3 the contraction runs on the bf16 matrix pipe and, with no
 *   fp32 volume to store, is bound by that pipe and not by HBM; the indices then are the argmax of THAT precision's volume
 *   (bit-identical to tcl_corr_to_flow over tcl_corr4d_fwd at the same precision). */
size_t tcl_corr4d_argmax_flow_workspace_bytes(int64_t B, int64_t C, int64_t F, int64_t P, int precision);
int tcl_corr4d_argmax_flow(const float *x_t, const float *v_t, const float *x_r, const float *v_r, float *flow,
                           int64_t B, int64_t C, int64_t F, int64_t S, int precision,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A2  per-pixel masked correlation.  Replaces compute_masked_correlation, utils/correlation.py:4-25.
 *   f1,f2 (B,C,P)  m1,m2 (B,P)  -> out (B,P).  Backward: g_f1,g_f2 (B,C,P), either nullable.
 * ------------------------------------------------------------------------------------------- */
int tcl_corr1x1_fwd(const float *f1, const float *f2, const float *m1, const float *m2, float *out,
                    int64_t B, int64_t C, int64_t P, void *stream);
int tcl_corr1x1_bwd(const float *grad_out, const float *f1, const float *f2, const float *m1, const float *m2,
                    float *g_f1, float *g_f2, int64_t B, int64_t C, int64_t P, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A3/A4  grid warp.  Replaces the F.grid_sample pairs of align_set (utils/flow.py:93-101,
 *   align_corners=1, grid given, v nearest) and of CPNet.align / CPNOriginal.align
 *   (models/cpn.py:280-295, models/cpn_original.py:276-284: align_corners=0, grid = affine_grid(theta),
 *   never materialised, v bilinear then > 0.5).
 *   Images are addressed as [nb][nf] with explicit element strides so that the reference's
 *   (B,C,F,H,W) tensors are read and written in place (no transposed copies):
 *     x     : element (b,f,c,y,x) at x + b*x_sb + f*x_sf + c*x_sc + y*W + x       (C channels, bilinear)
 *     v     : element (b,f,y,x)   at v + b*v_sb + f*v_sf + y*W + x               (nullable)
 *     x_out / v_out likewise with their own strides.
 *   Exactly one of grid (NB*NF,H,W,2 dense) / theta (NB*NF,2,3 dense) is non-NULL.
 * ------------------------------------------------------------------------------------------- */
int tcl_warp_fwd(const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc,
                 const float *v, int64_t v_sb, int64_t v_sf,
                 const float *grid, const float *theta,
                 float *x_out, int64_t xo_sb, int64_t xo_sf, int64_t xo_sc,
                 float *v_out, int64_t vo_sb, int64_t vo_sf,
                 int64_t NB, int64_t NF, int64_t C, int64_t H, int64_t W,
                 int align_corners, int v_mode, void *stream);

/*   tcl_warp_fwd with the elementwise kernels around it folded in (the one-pair inference unit of the sequence schedulers,
 *   thesis_dfpn/runner.py:153-159, and CPNet.align, models/cpn.py:283-290, issue them per call):
 *     v_is_hole_mask != 0 : `v` holds HOLE masks m; the sampled visibility is 1 - m, flipped per tap in the kernel (the
 *                           `1 - m` of models/cpn.py:288, thesis_dfpn/runner.py:157 never becomes a tensor);
 *     v_target (nullable) : (NB,H,W) at v_target + b*vt_sb [hole mask when vt_is_hole_mask]; then v_map (strided like v_out)
 *                           = clamp(v_out - v_target, 0, 1): T1 (thesis_dfpn/runner.py:158, models/cpn_original.py:287) in the
 *                           warp's epilogue.  Needs v_mode != NONE, 3-channel frames, W >= 2.
 *   All other arguments and every result as tcl_warp_fwd (which is this call with the extras off). */
int tcl_warp_fwd_fused(const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc,
                       const float *v, int64_t v_sb, int64_t v_sf, int v_is_hole_mask,
                       const float *grid, const float *theta,
                       float *x_out, int64_t xo_sb, int64_t xo_sf, int64_t xo_sc,
                       float *v_out, int64_t vo_sb, int64_t vo_sf,
                       const float *v_target, int64_t vt_sb, int vt_is_hole_mask,
                       float *v_map, int64_t vm_sb, int64_t vm_sf,
                       int64_t NB, int64_t NF, int64_t C, int64_t H, int64_t W,
                       int align_corners, int v_mode, void *stream);

/*   Backward of the bilinear branch: grad_x_out (strided like x_out) ->
 *     g_grid (NB*NF,H,W,2) or g_theta (NB*NF,2,3)  [whichever matches the forward; nullable]
 *     g_x    (strided like x; nullable; accumulated with float atomics into a buffer the CALLER zeroed).
 *   g_theta needs `workspace` of tcl_warp_bwd_workspace_bytes (per-block partial sums). */
size_t tcl_warp_bwd_workspace_bytes(int64_t NB, int64_t NF, int64_t H, int64_t W, int have_theta);
int tcl_warp_bwd(const float *grad_x_out, int64_t go_sb, int64_t go_sf, int64_t go_sc,
                 const float *x, int64_t x_sb, int64_t x_sf, int64_t x_sc,
                 const float *grid, const float *theta,
                 float *g_x, int64_t gx_sb, int64_t gx_sf, int64_t gx_sc,
                 float *g_grid, float *g_theta,
                 int64_t NB, int64_t NF, int64_t C, int64_t H, int64_t W, int align_corners,
                 void *workspace, size_t workspace_bytes, void *stream);

/*   F.affine_grid(theta, [N,C,H,W], align_corners) materialised: theta (N,2,3) -> grid (N,H,W,2).  The grid of
 *   MovementSimulator.simulate_movement (utils/movement.py:54-55, align_corners=1) and of CPNet.align
 *   (models/cpn.py:286, align_corners=0); tcl_warp_fwd(theta=...) samples through the same positions without storing them. */
int tcl_affine_grid(const float *theta, float *grid, int64_t N, int64_t H, int64_t W, int align_corners, void *stream);

/* T1  v_map = clamp(v_ref_aligned - v_target, 0, 1); thesis_dfpn/runner.py:137-138,158; models/cpn_original.py:287.
 *   v_aligned (B,F,P) dense, v_target (B,P) -> out (B,F,P). */
int tcl_vmap(const float *v_aligned, const float *v_target, float *out, int64_t B, int64_t F, int64_t P, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A5  masked softmax over the reference axis.  Replaces masked_softmax, models/cpn.py:7-15
 *   (= CM_Module.masked_softmax, models/cpn_original.py:191-199), for dim = the R axis of
 *   vec (O,R,I) (outer, refs, inner) with mask (Om,R,I) broadcast over the outer axis in groups:
 *   outer index o uses mask row o / (O/Om).
 * ------------------------------------------------------------------------------------------- */
int tcl_masked_softmax_fwd(const float *vec, const float *mask, float *out,
                           int64_t O, int64_t Om, int64_t R, int64_t I, void *stream);
int tcl_masked_softmax_bwd(const float *grad_out, const float *vec, const float *mask, float *g_vec,
                           int64_t O, int64_t Om, int64_t R, int64_t I, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A6/A7/A8  context matching + aggregation.  Replaces CPNContextMatchingDefault.forward
 *   (models/cpn.py:94-135), CPNContextMatchingComplete.forward (:142-195) and CM_Module.forward
 *   (models/cpn_original.py:201-243), selected by `variant`.
 *   c_feats (B,C,R+1,h,w) (frame 0 = target)   v_t (B,H,W)   v_aligned (B,R,H,W)
 *   p_in   (B,2C+1,h,w) [default, cm_module] or (B,C+1,h,w) [complete]
 *   c_mask (B,h,w)      c_match (B,R,h,w): the channel-independent softmax weights (after the hole mask
 *   of the complete variant); kept for the backward pass and for ref_importance.
 *   The workspace also receives what the backward pass needs (resized masks, similarities): pass it on.
 * ------------------------------------------------------------------------------------------- */
size_t tcl_ctx_workspace_bytes(int64_t B, int64_t C, int64_t R, int64_t h, int64_t w);
int tcl_ctx_fwd(const float *c_feats, const float *v_t, const float *v_aligned,
                float *p_in, float *c_mask, float *c_match,
                int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int64_t H, int64_t W,
                int variant, int normalize, void *workspace, size_t workspace_bytes, void *stream);
/*   g_p_in like p_in, g_c_mask (B,h,w) nullable -> g_c_feats (B,C,R+1,h,w).  fwd_workspace = the forward's. */
size_t tcl_ctx_bwd_workspace_bytes(int64_t B, int64_t C, int64_t R, int64_t h, int64_t w);
int tcl_ctx_bwd(const float *g_p_in, const float *g_c_mask, const float *c_feats, const float *c_match,
                const void *fwd_workspace, float *g_c_feats,
                int64_t B, int64_t C, int64_t R, int64_t h, int64_t w,
                int variant, int normalize, void *workspace, size_t workspace_bytes, void *stream);

/*   The one entry point that ALLOCATES NOTHING and needs no workspace.  It has no counterpart in the reference; it exists for
 *   callers that cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses that forbid
 *   hipMalloc.  Same arguments, kernels and results as tcl_ctx_fwd (inference: the backward pass still needs tcl_ctx_fwd's
 *   workspace), with the forward's scratch in a 4 MiB device array inside the library (one copy per GPU, held until the
 *   process ends).  Shapes whose tcl_ctx_workspace_bytes exceeds 4 MiB return TCL_ERR_WORKSPACE (C4's 32 targets x 10
 *   references at 30 x 53 need 2.7 MiB).  All calls on one GPU share that array: calls that overlap on one GPU (two streams,
 *   two threads) race; serialise them.  The array's address is looked up on the first call on each device: make that call
 *   outside graph capture. */
int tcl_ctx_fwd_static(const float *c_feats, const float *v_t, const float *v_aligned,
                       float *p_in, float *c_mask, float *c_match,
                       int64_t B, int64_t C, int64_t R, int64_t h, int64_t w, int64_t H, int64_t W,
                       int variant, int normalize, void *stream);

/* A9  ref_importance.  Replaces add_match_border (models/cpn.py:23-36) / the ref_importance block of the
 *   complete variant (models/cpn.py:178-186) incl. add_border (:39-44) and the 3x3-cross dilation of
 *   ImageTransforms.dilatate (utils/transforms.py:87-105), without the host round trip.
 *   c_match (B,R,h,w), v_t (B,H,W) -> ref_importance (B,3,R+1,H,W).  middle_is_vt: 0 = zero plane (default
 *   variant), 1 = v_t plane (complete variant). */
int tcl_ref_importance(const float *c_match, const float *v_t, float *ref_importance,
                       int64_t B, int64_t R, int64_t h, int64_t w, int64_t H, int64_t W, int middle_is_vt, void *stream);
/*   (bilinear(v, align_corners=0) > 0.5) resize of a mask, models/cpn.py:98,104.  v (N,H,W) -> out (N,h,w). */
int tcl_resize_mask(const float *v, float *out, int64_t N, int64_t H, int64_t W, int64_t h, int64_t w, void *stream);
/*   3x3-cross dilation, utils/transforms.py:87-105 (cv2.dilate MORPH_CROSS, 1 iteration).  img (N,H,W). */
int tcl_dilate_cross3(const float *img, float *out, int64_t N, int64_t H, int64_t W, void *stream);

/* ---------------------------------------------------------------------------------------------
 * A10-A13  flow utilities, utils/flow.py.
 * ------------------------------------------------------------------------------------------- */
/* corr_to_flow, utils/flow.py:8-26: corr (N,S*S [target], S*S [ref]) -> flow (N,S*S,2); first-max tie-break.
 *   NaN similarities are UNSUPPORTED here and in tcl_corr4d_argmax_flow: torch.argmax treats NaN as the maximum, these
 *   kernels compare with `>` and skip NaN entries (a row of finite values is bit-exact with the reference). */
int tcl_corr_to_flow(const float *corr, float *flow, int64_t N, int64_t S, void *stream);
/* flow_abs_to_relative (sign=-1) / flow_relative_to_abs (sign=+1), utils/flow.py:29-66.  flow (N,H,W,2). */
int tcl_flow_add_identity(const float *flow, float *out, int64_t N, int64_t H, int64_t W, int sign, void *stream);
/* crop_flow, utils/flow.py:69-84.  flow (N,H,W,2) -> out (N,ch,cw,2) cropped at (py,px). */
int tcl_crop_flow(const float *flow, float *out, int64_t N, int64_t H, int64_t W,
                  int64_t ch, int64_t cw, int64_t py, int64_t px, void *stream);
/* resize_flow, utils/flow.py:87-90 (legacy nearest).  flow (N,H,W,2) -> out (N,oh,ow,2). */
int tcl_resize_flow(const float *flow, float *out, int64_t N, int64_t H, int64_t W, int64_t oh, int64_t ow, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TCL_H_ */
