"""CPU oracle for the rows around the hot path: runner step, sequence schedulers, movement simulator.
TEST INFRASTRUCTURE ONLY (same rules as tcl_oracle.py: imported by tests/, smoke() and bench.py's cpu_baseline only).

Plain numpy, ONE sequence at a time, loops written the way the reference runs them (one (target, reference) pair per
step, batch size 1), on top of the pinned ops of tcl_oracle.py.  The networks are arguments (numpy callables).

Parity status: PINNED by tests/golden/{schedules,sequences,inpaint_step,movement}.npz, captured from the real reference
by tests/golden/make_golden_sequences.py and checked in tests/test_sequences.py.
"""
import numpy as np

from . import tcl_oracle as O

F32 = np.float32
FILL = np.array([0.485, 0.456, 0.406], dtype=F32).reshape(1, 3, 1, 1)


# =============================================================================================
# N2  step functions                 thesis_dfpn/runner.py:153-159, thesis_inpainting/runner.py:99-111
# =============================================================================================
def dfpn_infer_step(flow_fn, x_target, m_target, x_ref, m_ref):
    """flow_fn(x_ref) -> flow_256 (B,F,H,W,2).  -> (x_ref_aligned, v_ref_aligned, v_map).  thesis_dfpn/runner.py:153-159"""
    x_ref_aligned, v_ref_aligned = O.align_set(x_ref, 1 - m_ref, flow_fn(x_ref))
    return x_ref_aligned, v_ref_aligned, O.v_map(v_ref_aligned, 1 - m_target)


def cpn_align_step(theta_fn, x_target, m_target, x_refs, m_refs):
    """theta_fn(n) -> (n,2,3).  The warp half of CPNOriginal.align, models/cpn_original.py:276-290."""
    b, c, r, h, w = x_refs.shape
    xa, va, _ = O.cpn_align(x_refs, m_refs, None, theta_fn(b * r), list(range(r)))
    return xa, va, O.v_map(va, 1 - m_target)


def inpaint_infer_step(align_step, inpaint_fn, x_target, m_target, x_ref, m_ref):
    """align_step(x_target, m_target, x_ref, m_ref) -> (x_ref_aligned, v_ref_aligned, v_map); then the inpainting
    network.  thesis_inpainting/runner.py:99-111"""
    x_ref_aligned, v_ref_aligned, v_map = align_step(x_target, m_target, x_ref, m_ref)
    y_hat, y_hat_comp = inpaint_fn(x_target, 1 - m_target, x_ref_aligned, v_ref_aligned, v_map)
    return x_ref_aligned, v_ref_aligned, v_map, y_hat, y_hat_comp


# =============================================================================================
# N3  schedules                                  thesis_inpainting/runner.py:189-197, 222-227
# =============================================================================================
def ff_indexes(t, max_t, s, D):
    cands = [r for r in range(max_t) if r != t]
    order = sorted(zip([abs(r - t) for r in cands], cands))
    return [r for _, r in order if abs(r - t) <= D and abs(r - t) % s == 0]


def ip_indexes(t, t_list, s, D):
    done = list(reversed(t_list[:t_list.index(t)]))
    return done + [r for r in ff_indexes(t, len(t_list), s, D) if r not in done]


def cpn_reference_indexes(t, n_frames, p=2, max_len=120):
    """thesis_cpn/runner.py:154-171"""
    start, end = t - max_len, t + max_len
    if t - max_len < 0:
        end = (t + max_len) - (t - max_len)
        if end > n_frames:
            end = n_frames - 1
        start = 0
    elif t + max_len > n_frames:
        start = max((t - max_len) - (t + max_len - n_frames), 0)
        end = n_frames - 1
    return [i for i in range(start, end, p) if i != t]


def _cov(m):
    return float(np.sum(m, dtype=np.float64)) * 100 / m.size


def _pair(step, x_t, m_t, x_all, m_all, r):
    """One (target, reference r) step at batch 1 -> (v_map (1,1,H,W), y_hat_comp (1,3,H,W))."""
    _, _, v_map, _, y_hat_comp = step(x_t, m_t, x_all[None, :, [r]], m_all[None, :, [r]])
    return v_map[:, :, 0], y_hat_comp[:, :, 0]


def algorithm_ff(x, m, step, s=1, D=20, e=1):
    """x (C,F,H,W), m (1,F,H,W) -> (C,F,H,W).  thesis_inpainting/runner.py:170-187"""
    x, m = O._f(x), O._f(m)
    out = np.zeros_like(x)
    for t in range(x.shape[1]):
        x_t, m_t, comp = x[None, :, t].copy(), m[None, :, t].copy(), None
        cands = ff_indexes(t, x.shape[1], s, D)
        while comp is None or (cands and _cov(m_t) > e):
            v_map, comp = _pair(step, x_t, m_t, x, m, cands.pop(0))
            m_t = m_t - v_map
            x_t = ((1 - m_t) * comp + np.repeat(m_t, 3, axis=1) * FILL).astype(F32)
        out[:, t] = comp[0]
    return out


def algorithm_ip(x, m, step, s=1, D=20, e=1):
    """thesis_inpainting/runner.py:199-220"""
    y, mm = O._f(x).copy(), O._f(m).copy()
    f = y.shape[1]
    t_list = sorted(range(f), key=lambda k: abs(k - f // 2))
    for t in t_list:
        cands, comp = ip_indexes(t, t_list, s, D), None
        while comp is None or (cands and _cov(mm[:, t]) > e):
            v_map, comp = _pair(step, y[None, :, t], mm[None, :, t], y, mm, cands.pop(0))
            mm[:, t] = mm[:, t] - v_map[0]
            y[:, t] = (1 - mm[:, t]) * comp[0] + np.repeat(mm[:, t], 3, axis=0) * FILL[0]
        mm[:, t] = 0
        y[:, t] = comp[0]
    return y


def algorithm_cp(x, m, step, N=20, s=1, e=1):
    """thesis_inpainting/runner.py:229-253"""
    y, mm = O._f(x).copy(), O._f(m).copy()
    f = y.shape[1]
    for i in range(N):
        for t in [k for k in range(f) if (k // s) % (s if s > 1 else 2) == i % 2]:
            if mm[:, t].sum() == 0:
                continue
            for r in (t - s, t + s):
                if not 0 <= r < f:
                    continue
                v_map, comp = _pair(step, y[None, :, t], mm[None, :, t], y, mm, r)
                mm[:, t] = mm[:, t] - v_map[0]
                y[:, t] = (1 - mm[:, t]) * comp[0] + np.repeat(mm[:, t], 3, axis=0) * FILL[0]
                if _cov(mm[:, t]) < e or i >= N - 2:
                    mm[:, t] = 0
                    y[:, t] = comp[0]
    return y


def algorithm_cpn(x, m, model):
    """Forward + backward pass blended linearly in time.  thesis_cpn/runner.py:121-152"""
    x, m = O._f(x), O._f(m)
    c, f, h, w = x.shape
    both = np.zeros((2, c, f, h, w), F32)
    for d in range(2):
        xc, mc = x[None].copy(), m[None].copy()
        for t in (range(f) if d == 0 else reversed(range(f))):
            r_list = cpn_reference_indexes(t, f)
            _, comp = model(xc[:, :, t], mc[:, :, t], xc[:, :, r_list], mc[:, :, r_list])
            xc[:, :, t] = comp
            mc[:, :, t] = 0
            both[d, :, t] = comp[0]
    k = np.arange(f, dtype=np.float64).reshape(1, f, 1, 1)
    return (both[0] * (k / f) + both[1] * ((f - k) / f)).astype(F32)


# =============================================================================================
# N4  movement simulator                                                utils/movement.py:23-107
# =============================================================================================
def stack_transformations(mats, t):
    """(n,3,3) float32 per-frame transforms -> transforms relative to frame t.  utils/movement.py:83-96"""
    mats = O._f(mats)
    out = np.zeros_like(mats)
    out[t] = mats[t]
    for i in range(t - 1, -1, -1):
        out[i] = np.linalg.inv(mats[i]).astype(F32) @ out[i + 1]
    for i in range(t + 1, mats.shape[0]):
        out[i] = mats[i] @ out[i - 1]
    return out


def affine2theta(p, h, w):
    """utils/movement.py:98-107 in float32, left to right."""
    p = O._f(p)
    H, W, one, two = F32(h), F32(w), F32(1), F32(2)
    return np.array([[p[0, 0], p[0, 1] * H / W, p[0, 2] * two / W + p[0, 0] + p[0, 1] - one],
                     [p[1, 0] * W / H, p[1, 1], p[1, 2] * two / H + p[1, 0] + p[1, 1] - one]], dtype=F32)


def simulate_movement(x, mats):
    """x (C,H,W), mats (n,3,3) -> (y (C,n,H,W), flow_inv (n,H,W,2)); align_corners=True.  utils/movement.py:23-61"""
    x, mats = O._f(x), O._f(mats)
    c, h, w = x.shape
    n = mats.shape[0]
    inv = np.stack([np.linalg.inv(a).astype(F32) for a in mats])
    theta = np.stack([affine2theta(a, h, w) for a in stack_transformations(mats, n // 2)])
    theta_inv = np.stack([affine2theta(a, h, w) for a in stack_transformations(inv, n // 2)])
    y = O.grid_sample_bilinear(np.repeat(x[None], n, axis=0), O.affine_grid(theta, h, w, True), True)
    return y.transpose(1, 0, 2, 3).copy(), O.affine_grid(theta_inv, h, w, True)


def transform_single(image, mat):
    """align_corners=False (torch's default).  utils/movement.py:76-81"""
    image = O._f(image)
    c, h, w = image.shape
    return O.grid_sample_bilinear(image[None], O.affine_grid(affine2theta(mat, h, w)[None], h, w, False), False)[0]
