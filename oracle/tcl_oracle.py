"""CPU oracle for the frame-alignment / temporal-copying hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy (float32) restatement of the algorithms on the reference's hot path.  It is
the checker: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  The product (`master_thesis_amd/`) never imports, links or calls anything in `oracle/`.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py` against golden
vectors produced by running the real reference (imported from /root/reference in the build container by
`tests/golden/make_golden.py`) on seeded inputs.  The one exception is the 3x3-cross dilation used for
the `ref_importance` border (cv2 is absent from the image): it is pinned by the scipy restatement of
`cv2.dilate` inside the golden generator, see the header of that script.

Each function cites the reference file:line (relative to the reference root) that it restates.  The
restatement is formula-level: nothing here calls torch / ATen, and gradients are hand-derived.

Conventions: x = frames (B,C,F,H,W); v = visibility (1 = visible); flows are absolute
sampling grids in [-1,1] with last dim (x,y).
"""
import numpy as np

F32 = np.float32
EPS = F32(1e-9)

# Accumulator of the LONG reductions (the 203,520-term similarity of context matching and the sums of its backward, the
# per-image theta gradient).  float32 restates the reference's own arithmetic; tests that hold the HIP path to the fp32
# bar at full size switch to float64 (`with long_sums(np.float64)`), so that the yardstick is the exact sum of the same
# terms rather than another float32 summation order.
_ACC = F32


class long_sums:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _ACC
        self.prev, _ACC = _ACC, self.dtype
        return self

    def __exit__(self, *exc):
        global _ACC
        _ACC = self.prev
        return False


def _f(a):
    return np.ascontiguousarray(a, dtype=F32)


# =============================================================================================
# A1  all-pairs masked correlation                                  utils/correlation.py:28-54
# =============================================================================================
def _normalise_over_c(t):
    """t (..., C, P) -> t / (||t||_2 over C + 1e-9); also returns the norm (..., 1, P).  correlation.py:47-50"""
    n = np.sqrt(np.sum(t * t, axis=-2, keepdims=True, dtype=F32)).astype(F32)
    return (t / (n + EPS)).astype(F32), n


def compute_masked_4d_correlation(x_target_feats, v_target, x_ref_feats, v_ref):
    """(B,C,H,W), (B,1,H,W)|None, (B,C,F,H,W), (B,1,F,H,W)|None -> (B,F,H,W,H,W).  utils/correlation.py:28-54.
    First (H,W) pair indexes the target pixel, last pair the reference pixel."""
    xt, xr = _f(x_target_feats), _f(x_ref_feats)
    b, c, f, h, w = xr.shape
    if v_target is not None:
        xt = xt * _f(v_target)
    if v_ref is not None:
        xr = xr * _f(v_ref)
    a_hat, _ = _normalise_over_c(xt.reshape(b, c, h * w))                     # (B,C,P)
    r_hat, _ = _normalise_over_c(xr.reshape(b, c, f, h * w).transpose(0, 2, 1, 3))  # (B,F,C,Q)
    corr = np.matmul(a_hat.transpose(0, 2, 1)[:, None], r_hat)               # (B,F,P,Q)
    return corr.reshape(b, f, h, w, h, w).astype(F32)


def _normalise_bwd(t, n, d_hat):
    """Backward of t/(||t||+eps) over axis -2.  At ||t|| == 0 the norm's subgradient is 0 (torch's norm
    backward), leaving d_hat/eps."""
    ne = n + EPS
    dot = np.sum(d_hat * t, axis=-2, keepdims=True, dtype=F32)
    safe_n = np.where(n > 0, n, F32(1))
    corr = np.where(n > 0, dot / (ne * ne * safe_n), F32(0)).astype(F32)
    return (d_hat / ne - t * corr).astype(F32)


def compute_masked_4d_correlation_bwd(x_target_feats, v_target, x_ref_feats, v_ref, grad_out):
    """Hand-derived gradient of A1 w.r.t. the two feature tensors (masks are data)."""
    xt, xr = _f(x_target_feats), _f(x_ref_feats)
    b, c, f, h, w = xr.shape
    p = h * w
    vt = _f(v_target).reshape(b, 1, p) if v_target is not None else np.ones((b, 1, p), F32)
    vr = _f(v_ref).reshape(b, 1, f, p).transpose(0, 2, 1, 3) if v_ref is not None else np.ones((b, f, 1, p), F32)
    t = xt.reshape(b, c, p) * vt                                   # (B,C,P)
    r = xr.reshape(b, c, f, p).transpose(0, 2, 1, 3) * vr          # (B,F,C,Q)
    a_hat, n_t = _normalise_over_c(t)
    r_hat, n_r = _normalise_over_c(r)
    g = _f(grad_out).reshape(b, f, p, p)
    d_a = np.matmul(r_hat, g.transpose(0, 1, 3, 2)).sum(axis=1, dtype=F32)   # (B,F,C,Q)@(B,F,Q,P) -> sum over refs -> (B,C,P)
    d_r = np.matmul(a_hat[:, None], g).astype(F32)                            # (B,1,C,P)@(B,F,P,Q) -> (B,F,C,Q)
    g_xt = _normalise_bwd(t, n_t, d_a) * vt
    g_xr = _normalise_bwd(r, n_r, d_r) * vr
    return g_xt.reshape(b, c, h, w).astype(F32), g_xr.transpose(0, 2, 1, 3).reshape(b, c, f, h, w).astype(F32)


# =============================================================================================
# A2  per-pixel masked correlation                                   utils/correlation.py:4-25
# =============================================================================================
def compute_masked_correlation(feats_1, feats_2, mask_1, mask_2):
    """(B,C,H,W) x2, masks (B,1,H,W) -> (B,1,H,W).  utils/correlation.py:4-25"""
    f1, f2 = _f(feats_1) * _f(mask_1), _f(feats_2) * _f(mask_2)
    b, c, h, w = f1.shape
    a1, _ = _normalise_over_c(f1.reshape(b, c, h * w))
    a2, _ = _normalise_over_c(f2.reshape(b, c, h * w))
    return np.sum(a1 * a2, axis=1, dtype=F32).reshape(b, 1, h, w).astype(F32)


def compute_masked_correlation_bwd(feats_1, feats_2, mask_1, mask_2, grad_out):
    b, c, h, w = feats_1.shape
    m1, m2 = _f(mask_1).reshape(b, 1, -1), _f(mask_2).reshape(b, 1, -1)
    t1, t2 = _f(feats_1).reshape(b, c, -1) * m1, _f(feats_2).reshape(b, c, -1) * m2
    a1, n1 = _normalise_over_c(t1)
    a2, n2 = _normalise_over_c(t2)
    g = _f(grad_out).reshape(b, 1, -1)
    g1 = _normalise_bwd(t1, n1, g * a2) * m1
    g2 = _normalise_bwd(t2, n2, g * a1) * m2
    return g1.reshape(b, c, h, w).astype(F32), g2.reshape(b, c, h, w).astype(F32)


# =============================================================================================
# grid sampling primitives (ATen semantics used by utils/flow.py:93-101 and models/cpn.py:280-295)
# =============================================================================================
def _unnormalise(g, size, align_corners):
    g = g.astype(F32)
    if align_corners:
        return ((g + F32(1)) / F32(2) * F32(size - 1)).astype(F32)
    return (((g + F32(1)) * F32(size) - F32(1)) / F32(2)).astype(F32)


def _gather(x, iy, ix):
    """x (N,C,H,W); iy, ix (N,Ho,Wo) integer -> (N,C,Ho,Wo), zero outside."""
    n, c, h, w = x.shape
    inb = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    iyc, ixc = np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)
    nn = np.arange(n)[:, None, None]
    val = x[nn, :, iyc, ixc]                     # (N,Ho,Wo,C)
    val = np.where(inb[..., None], val, F32(0))
    return np.moveaxis(val, 3, 1).astype(F32)


def _bilinear_params(grid, h, w, align_corners):
    ix = _unnormalise(grid[..., 0], w, align_corners)
    iy = _unnormalise(grid[..., 1], h, align_corners)
    x0, y0 = np.floor(ix), np.floor(iy)
    wx = (ix - x0).astype(F32)    # distance to the west neighbour
    ex = (F32(1) - wx).astype(F32)
    ny = (iy - y0).astype(F32)    # distance to the north neighbour
    sy = (F32(1) - ny).astype(F32)
    return x0.astype(np.int64), y0.astype(np.int64), wx, ex, ny, sy


def grid_sample_bilinear(x, grid, align_corners):
    """F.grid_sample(mode='bilinear', padding_mode='zeros').  x (N,C,H,W), grid (N,Ho,Wo,2)."""
    x, grid = _f(x), _f(grid)
    n, c, h, w = x.shape
    x0, y0, wx, ex, ny, sy = _bilinear_params(grid, h, w, align_corners)
    nw, ne, sw, se = (sy * ex)[:, None], (sy * wx)[:, None], (ny * ex)[:, None], (ny * wx)[:, None]
    out = _gather(x, y0, x0) * nw + _gather(x, y0, x0 + 1) * ne
    out = out + _gather(x, y0 + 1, x0) * sw
    out = out + _gather(x, y0 + 1, x0 + 1) * se
    return out.astype(F32)


def grid_sample_nearest(x, grid, align_corners):
    """F.grid_sample(mode='nearest', padding_mode='zeros'): index = nearbyint (half to even)."""
    x, grid = _f(x), _f(grid)
    n, c, h, w = x.shape
    ix = np.rint(_unnormalise(grid[..., 0], w, align_corners)).astype(np.int64)
    iy = np.rint(_unnormalise(grid[..., 1], h, align_corners)).astype(np.int64)
    return _gather(x, iy, ix)


def _scatter_add(gx, iy, ix, val):
    """gx (N,C,H,W) += val (N,C,Ho,Wo) at (iy,ix) where in bounds."""
    n, c, h, w = gx.shape
    inb = (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    nn = np.broadcast_to(np.arange(n)[:, None, None], ix.shape)
    for ch in range(c):
        np.add.at(gx[:, ch], (nn[inb], iy[inb], ix[inb]), val[:, ch][inb])


def grid_sample_bilinear_bwd(x, grid, align_corners, grad_out):
    """-> (grad_x, grad_grid).  grad_x is a scatter-add of grad_out * weight; grad_grid the per-pixel sum over C."""
    x, grid, g = _f(x), _f(grid), _f(grad_out)
    n, c, h, w = x.shape
    x0, y0, wx, ex, ny, sy = _bilinear_params(grid, h, w, align_corners)
    gx = np.zeros_like(x)
    _scatter_add(gx, y0, x0, g * (sy * ex)[:, None])
    _scatter_add(gx, y0, x0 + 1, g * (sy * wx)[:, None])
    _scatter_add(gx, y0 + 1, x0, g * (ny * ex)[:, None])
    _scatter_add(gx, y0 + 1, x0 + 1, g * (ny * wx)[:, None])
    v_nw, v_ne = _gather(x, y0, x0), _gather(x, y0, x0 + 1)
    v_sw, v_se = _gather(x, y0 + 1, x0), _gather(x, y0 + 1, x0 + 1)
    d_ix = np.sum(g * ((v_ne - v_nw) * sy[:, None] + (v_se - v_sw) * ny[:, None]), axis=1, dtype=F32)
    d_iy = np.sum(g * ((v_sw - v_nw) * ex[:, None] + (v_se - v_ne) * wx[:, None]), axis=1, dtype=F32)
    mx = F32(w - 1) / F32(2) if align_corners else F32(w) / F32(2)
    my = F32(h - 1) / F32(2) if align_corners else F32(h) / F32(2)
    return gx.astype(F32), np.stack([d_ix * mx, d_iy * my], axis=-1).astype(F32)


def _linspace(steps):
    """torch.linspace(-1, 1, steps) in float32 (symmetric evaluation from both ends)."""
    if steps <= 1:
        return np.zeros((steps,), F32)
    step = F32(2) / F32(steps - 1)
    idx = np.arange(steps)
    lo = (F32(-1) + step * idx.astype(F32)).astype(F32)
    hi = (F32(1) - step * (steps - 1 - idx).astype(F32)).astype(F32)
    return np.where(idx < steps // 2, lo, hi).astype(F32)


def _base_coords(steps, align_corners):
    r = _linspace(steps)
    if not align_corners and steps > 1:
        r = (r * F32(steps - 1) / F32(steps)).astype(F32)
    return r


def affine_grid(theta, h, w, align_corners):
    """F.affine_grid(theta (N,2,3), (N,C,h,w)) -> (N,h,w,2)."""
    theta = _f(theta)
    xs, ys = _base_coords(w, align_corners), _base_coords(h, align_corners)
    X, Y = np.broadcast_to(xs[None, :], (h, w)), np.broadcast_to(ys[:, None], (h, w))
    gx = theta[:, 0, 0, None, None] * X + theta[:, 0, 1, None, None] * Y + theta[:, 0, 2, None, None]
    gy = theta[:, 1, 0, None, None] * X + theta[:, 1, 1, None, None] * Y + theta[:, 1, 2, None, None]
    return np.stack([gx, gy], axis=-1).astype(F32)


def affine_grid_bwd(grad_grid, h, w, align_corners):
    """grad_grid (N,h,w,2) -> grad_theta (N,2,3)."""
    g = _f(grad_grid)
    xs, ys = _base_coords(w, align_corners), _base_coords(h, align_corners)
    base = np.stack([np.broadcast_to(xs[None, :], (h, w)), np.broadcast_to(ys[:, None], (h, w)), np.ones((h, w), F32)], axis=-1)
    return np.einsum("nhwi,hwj->nij", g.astype(np.float64), base.astype(np.float64)).astype(F32)


# =============================================================================================
# A3  align_set                                                          utils/flow.py:93-101
# =============================================================================================
def align_set(x, v, flow):
    """x (B,3,F,H,W), v (B,1,F,H,W), flow (B,F,H,W,2) -> (x_aligned, v_aligned), align_corners=True;
    x bilinear, v nearest.  utils/flow.py:93-101"""
    x, v, flow = _f(x), _f(v), _f(flow)
    b, c, f, h, w = x.shape
    grid = flow.reshape(b * f, h, w, 2)
    xa = grid_sample_bilinear(x.transpose(0, 2, 1, 3, 4).reshape(b * f, c, h, w), grid, True)
    va = grid_sample_nearest(v.transpose(0, 2, 1, 3, 4).reshape(b * f, 1, h, w), grid, True)
    return (xa.reshape(b, f, c, h, w).transpose(0, 2, 1, 3, 4).copy(), va.reshape(b, f, 1, h, w).transpose(0, 2, 1, 3, 4).copy())


def align_set_bwd(x, v, flow, grad_x_aligned):
    """-> (grad_x, grad_flow); the nearest branch contributes no gradient to the flow."""
    x, flow, g = _f(x), _f(flow), _f(grad_x_aligned)
    b, c, f, h, w = x.shape
    gx, gg = grid_sample_bilinear_bwd(x.transpose(0, 2, 1, 3, 4).reshape(b * f, c, h, w), flow.reshape(b * f, h, w, 2), True,
                                      g.transpose(0, 2, 1, 3, 4).reshape(b * f, c, h, w))
    return gx.reshape(b, f, c, h, w).transpose(0, 2, 1, 3, 4).copy(), gg.reshape(b, f, h, w, 2)


def v_map(v_ref_aligned, v_target):
    """clamp(v_ref_aligned - v_target[:, :, None], 0, 1).  thesis_dfpn/runner.py:137-138,158; models/cpn_original.py:287"""
    return np.clip(_f(v_ref_aligned) - _f(v_target)[:, :, None], 0, 1).astype(F32)


# =============================================================================================
# A4  CPN affine alignment (the affine_grid + grid_sample part)         models/cpn.py:280-295
# =============================================================================================
def cpn_align(x, m, y, theta, r_list):
    """x,y (B,3,F,H,W), m (B,1,F,H,W), theta (B*R,2,3) -> x_aligned (B,3,R,H,W), v_aligned (B,1,R,H,W) in {0,1},
    y_aligned.  align_corners=False everywhere; v = bilinear(1-m) > 0.5.  models/cpn.py:280-295"""
    x, m = _f(x), _f(m)
    b, c, f, h, w = x.shape
    r = len(r_list)
    grid = affine_grid(theta, h, w, False)

    def warp(t, ch):
        src = t[:, :, r_list].transpose(0, 2, 1, 3, 4).reshape(b * r, ch, h, w)
        return grid_sample_bilinear(src, grid, False).reshape(b, r, ch, h, w).transpose(0, 2, 1, 3, 4).copy()

    xa = warp(x, c)
    va = (warp(F32(1) - m, 1) > F32(0.5)).astype(F32)
    ya = warp(_f(y), c) if y is not None else None
    return xa, va, ya


def cpn_align_bwd(x, theta, r_list, grad_x_aligned):
    """-> (grad_x (scattered into the r_list frames), grad_theta)."""
    x, g = _f(x), _f(grad_x_aligned)
    b, c, f, h, w = x.shape
    r = len(r_list)
    grid = affine_grid(theta, h, w, False)
    src = x[:, :, r_list].transpose(0, 2, 1, 3, 4).reshape(b * r, c, h, w)
    gsrc, ggrid = grid_sample_bilinear_bwd(src, grid, False, g.transpose(0, 2, 1, 3, 4).reshape(b * r, c, h, w))
    gx = np.zeros_like(x)
    gx[:, :, r_list] = gsrc.reshape(b, r, c, h, w).transpose(0, 2, 1, 3, 4)
    return gx, affine_grid_bwd(ggrid, h, w, False)


# =============================================================================================
# A5  masked softmax                      models/cpn.py:7-15 (= models/cpn_original.py:191-199)
# =============================================================================================
def masked_softmax(vec, mask, dim):
    """NB the max is taken over the MASKED values, zeros of masked-out entries included.  models/cpn.py:7-15"""
    vec, mask = _f(vec), _f(mask)
    mv = vec * mask
    mx = np.max(mv, axis=dim, keepdims=True)
    me = (np.exp(mv - mx).astype(F32) * mask).astype(F32)
    s = np.sum(me, axis=dim, keepdims=True, dtype=F32)
    s = s + (s < F32(1e-4)).astype(F32)
    return (me / s).astype(F32)


def masked_softmax_bwd(vec, mask, dim, grad_out):
    """General gradient, including the `sums < 1e-4` branch where the max shift does not cancel; the max
    routes its gradient to the first maximal index."""
    vec, mask, g = _f(vec), _f(mask), _f(grad_out)
    mask_b = np.broadcast_to(mask, vec.shape)
    mv = vec * mask_b
    mx = np.max(mv, axis=dim, keepdims=True)
    me = np.exp(mv - mx).astype(F32) * mask_b
    s = np.sum(me, axis=dim, keepdims=True, dtype=F32)
    sp = s + (s < F32(1e-4)).astype(F32)
    d_me = g / sp - np.sum(g * me, axis=dim, keepdims=True, dtype=F32) / (sp * sp)
    d_arg = d_me * me                       # gradient w.r.t. (mv - mx)
    am = np.expand_dims(np.argmax(mv, axis=dim), dim)
    onehot = np.zeros_like(mv)
    np.put_along_axis(onehot, am, F32(1), axis=dim)
    d_mv = d_arg - onehot * np.sum(d_arg, axis=dim, keepdims=True, dtype=F32)
    return (d_mv * mask_b).astype(F32)


# =============================================================================================
# bilinear resizing (align_corners=False) as used on masks and on c_match       models/cpn.py:98,104,27-29
# =============================================================================================
def _linear_taps(in_size, out_size):
    scale = F32(in_size) / F32(out_size)
    dst = np.arange(out_size, dtype=F32)
    src = (scale * (dst + F32(0.5)) - F32(0.5)).astype(F32)
    src = np.where(src < 0, F32(0), src).astype(F32)
    i0 = src.astype(np.int64)
    i1 = i0 + (i0 < in_size - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    l0 = (F32(1) - l1).astype(F32)
    return i0, i1, l0, l1


def interpolate_bilinear(x, size):
    """F.interpolate(x (N,C,H,W), size, mode='bilinear', align_corners=False)."""
    x = _f(x)
    y0, y1, ly0, ly1 = _linear_taps(x.shape[2], size[0])
    x0, x1, lx0, lx1 = _linear_taps(x.shape[3], size[1])
    top = x[:, :, y0][:, :, :, x0] * lx0 + x[:, :, y0][:, :, :, x1] * lx1
    bot = x[:, :, y1][:, :, :, x0] * lx0 + x[:, :, y1][:, :, :, x1] * lx1
    return (top * ly0[:, None] + bot * ly1[:, None]).astype(F32)


def resize_mask(v, size):
    """(F.interpolate(v, size, 'bilinear', align_corners=False) > 0.5).float()   models/cpn.py:98,104"""
    return (interpolate_bilinear(v, size) > F32(0.5)).astype(F32)


# =============================================================================================
# A9  3x3-cross dilation and the ref_importance border      utils/transforms.py:87-105, models/cpn.py:23-44
# =============================================================================================
def dilate_cross3(img):
    """cv2.dilate(img, MORPH_CROSS 3x3, iterations=1) on the last two axes; pixels outside do not contribute."""
    img = _f(img)
    out = img.copy()
    out[..., 1:, :] = np.maximum(out[..., 1:, :], img[..., :-1, :])
    out[..., :-1, :] = np.maximum(out[..., :-1, :], img[..., 1:, :])
    out[..., :, 1:] = np.maximum(out[..., :, 1:], img[..., :, :-1])
    out[..., :, :-1] = np.maximum(out[..., :, :-1], img[..., :, 1:])
    return out


def _paint_border(planes, hole):
    """planes (B,R+1,H,W) grey importance; hole = 1 - v_t (B,1,H,W).  Border pixels become red in every plane.
    models/cpn.py:23-26,34-36 and :39-44"""
    border = ((dilate_cross3(hole) - hole) > 0).astype(F32)           # (B,1,H,W)
    rgb = np.repeat(planes[:, None], 3, axis=1)                        # (B,3,R+1,H,W)
    out = rgb * (F32(1) - border[:, :, None])
    out[:, 0] += border
    return out.astype(F32)


# =============================================================================================
# A6/A7/A8  context matching + aggregation     models/cpn.py:94-135, :142-195; models/cpn_original.py:201-243
# =============================================================================================
def _context_common(c_feats, v_t, v_aligned, variant, normalize):
    cf = _f(c_feats)
    b, c, f, h, w = cf.shape
    r = f - 1
    vt = resize_mask(v_t, (h, w))                                       # (B,1,h,w)
    vr = np.stack([resize_mask(_f(v_aligned)[:, :, i], (h, w)) for i in range(r)], axis=2)  # (B,1,R,h,w)
    vmap = vt[:, :, None] * vr                                          # (B,1,R,h,w)
    v_sum = vmap[:, 0].sum(axis=(-1, -2), dtype=F32)                    # (B,R)
    zeros = v_sum < F32(1e-4)
    v_sum = v_sum + zeros.astype(F32)
    ft, fr = cf[:, :, 0], cf[:, :, 1:]
    gs = np.sum((vmap * ft[:, :, None]).astype(_ACC) * fr.astype(_ACC), axis=(1, 3, 4), dtype=_ACC).astype(F32)  # (B,R)
    if variant == "cm_module":
        norm = v_sum * F32(c)              # cpn_original.py:226 always divides by v_sum and C
    else:
        norm = v_sum * F32(c) if normalize else np.ones_like(v_sum)
    gs = np.where(zeros, F32(0), gs / norm).astype(F32)
    return cf, vt, vr, vmap, zeros, norm, gs


def context_matching(c_feats, v_t, v_aligned, normalize=True, variant="default"):
    """variant in {'default' (models/cpn.py:94-135), 'complete' (:142-195), 'cm_module' (cpn_original.py:201-243)}.
    Returns (p_in, c_mask, ref_importance|None)."""
    cf, vt, vr, vmap, zeros, norm, gs = _context_common(c_feats, v_t, v_aligned, variant, normalize)
    b, c, f, h, w = cf.shape
    r = f - 1
    cos = np.broadcast_to(gs[:, None, :, None, None], (b, 1, r, h, w))
    c_match = masked_softmax(cos, vr, 2)                                # (B,1,R,h,w): identical for every channel
    if variant == "complete":
        c_match = c_match * (F32(1) - vt)[:, :, None]
    c_out = np.sum(cf[:, :, 1:] * c_match, axis=2, dtype=F32)           # (B,C,h,w)
    c_mask = (F32(1) - np.sum(c_match * vr, axis=2, dtype=F32)).astype(F32)   # (B,1,h,w)
    ref_importance = None
    if variant != "cm_module":
        v_t = _f(v_t)
        H, W = v_t.shape[2], v_t.shape[3]
        imp = interpolate_bilinear(c_match[:, 0], (H, W))               # (B,R,H,W)
        mid = np.zeros((b, 1, H, W), F32) if variant == "default" else v_t
        planes = np.concatenate([imp[:, :r // 2], mid, imp[:, r // 2:]], axis=1)
        ref_importance = _paint_border(planes, F32(1) - v_t)
    if variant == "complete":
        p_in = np.concatenate([cf[:, :, 0] * vt + c_out, c_mask], axis=1)
    else:
        p_in = np.concatenate([cf[:, :, 0], c_out, c_mask], axis=1)
    return p_in.astype(F32), c_mask, ref_importance


def context_matching_bwd(c_feats, v_t, v_aligned, grad_p_in, grad_c_mask, normalize=True, variant="default"):
    """Hand-derived gradient w.r.t. c_feats of (p_in, c_mask); ref_importance is treated as a visualisation output."""
    cf, vt, vr, vmap, zeros, norm, gs = _context_common(c_feats, v_t, v_aligned, variant, normalize)
    b, c, f, h, w = cf.shape
    r = f - 1
    gp = _f(grad_p_in)
    ft, fr = cf[:, :, 0], cf[:, :, 1:]
    cos = np.ascontiguousarray(np.broadcast_to(gs[:, None, :, None, None], (b, 1, r, h, w)))
    sm = masked_softmax(cos, vr, 2)
    hole = (F32(1) - vt)[:, :, None] if variant == "complete" else F32(1)
    c_match = sm * hole
    if variant == "complete":
        g_dec, g_mask = gp[:, :c], gp[:, c:c + 1] + _f(grad_c_mask)
        g_ft_direct, g_out = g_dec * vt, g_dec
    else:
        g_ft_direct, g_out, g_mask = gp[:, :c], gp[:, c:2 * c], gp[:, 2 * c:2 * c + 1] + _f(grad_c_mask)
    g_fr = g_out[:, :, None] * c_match                                    # direct path through c_out
    d_match = np.sum(g_out[:, :, None].astype(_ACC) * fr.astype(_ACC), axis=1, keepdims=True, dtype=_ACC).astype(F32) - g_mask[:, :, None] * vr
    d_sm = d_match * hole
    d_cos = masked_softmax_bwd(cos, vr, 2, d_sm)                          # (B,1,R,h,w)
    d_gs = np.sum(d_cos, axis=(1, 3, 4), dtype=_ACC).astype(F32)           # (B,R)
    d_gs = np.where(zeros, F32(0), d_gs / norm).astype(F32)
    coef = d_gs[:, None, :, None, None] * vmap                            # (B,1,R,h,w)
    g_ft = g_ft_direct + np.sum(coef * fr, axis=2, dtype=F32)
    g_fr = g_fr + coef * ft[:, :, None]
    return np.concatenate([g_ft[:, :, None], g_fr], axis=2).astype(F32)


# =============================================================================================
# A10-A13  flow utilities                                                    utils/flow.py:8-90
# =============================================================================================
def corr_to_flow(corr):
    """(B,F,S,S,S,S) -> (B,F,S,S,2): first-max argmax over the last S*S axis -> (x,y) in [-1,1].  utils/flow.py:8-26"""
    corr = _f(corr)
    b, f, s = corr.shape[:3]
    idx = np.argmax(corr.reshape(b, f, s, s, s * s), axis=-1)
    pos = np.stack([idx % s, idx // s], axis=-1).astype(F32)
    return (pos * F32(2 / (s - 1)) - F32(1)).astype(F32)


def _identity_grid(h, w):
    xs, ys = _linspace(w), _linspace(h)
    return np.stack([np.broadcast_to(xs[None, :], (h, w)), np.broadcast_to(ys[:, None], (h, w))], axis=-1).astype(F32)


def flow_abs_to_relative(flow):
    """utils/flow.py:29-46"""
    flow = _f(flow)
    return (flow - _identity_grid(flow.shape[2], flow.shape[3])).astype(F32)


def flow_relative_to_abs(flow_rel):
    """utils/flow.py:49-66"""
    flow_rel = _f(flow_rel)
    return (flow_rel + _identity_grid(flow_rel.shape[2], flow_rel.shape[3])).astype(F32)


def crop_flow(flow, crop_size, crop_position):
    """utils/flow.py:69-84 (x scaled by w/crop_w, y by h/crop_h, python-float factors)."""
    flow = _f(flow)
    h, w = flow.shape[2], flow.shape[3]
    rel = flow_abs_to_relative(flow)
    cut = rel[:, :, crop_position[0]:crop_position[0] + crop_size[0], crop_position[1]:crop_position[1] + crop_size[1]].copy()
    cut[..., 0] *= F32(w / crop_size[1])
    cut[..., 1] *= F32(h / crop_size[0])
    return flow_relative_to_abs(cut)


def resize_flow(flow, size):
    """utils/flow.py:87-90: F.interpolate default mode = legacy nearest, src = min(floor(dst * in/out), in-1)."""
    flow = _f(flow)
    h, w = flow.shape[2], flow.shape[3]
    sy = np.minimum(np.floor(np.arange(size[0], dtype=F32) * (F32(h) / F32(size[0]))).astype(np.int64), h - 1)
    sx = np.minimum(np.floor(np.arange(size[1], dtype=F32) * (F32(w) / F32(size[1]))).astype(np.int64), w - 1)
    return np.ascontiguousarray(flow[:, :, sy][:, :, :, sx])
