"""torch.autograd wrappers around the C ABI of libtcl_hip.so.

Everything here is plumbing: tensors are checked (HIP device, float32), made dense where the kernels need
it, outputs and workspaces come from torch's caching allocator, and the call is enqueued on torch's current
stream of the tensor's device.  There is no CPU path: a CPU tensor raises.
"""
import ctypes

import torch

from . import _lib
from ._lib import TclError

# Under torch.autocast the surrounding convolutions hand the ops bf16 / fp16 tensors.  The kernels compute in fp32, so every
# autograd Function casts its floating-point GPU inputs to fp32 at the op boundary while autocast is active (and runs its
# body with autocast off), exactly as a stock fp32-only op behaves; outside autocast a non-fp32 tensor is an error.
_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


# ----------------------------------------------------------------------------------------------- helpers
def _require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise TclError("tcl ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise TclError("tcl ops take float32 tensors (got %s)" % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise TclError("tensors on different devices: %s vs %s" % (dev, t.device))
    return dev


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev):
    """Make `dev` the current HIP device for the launches inside: a no-op when it already is (the one-GPU-per-process case: the
    torch.cuda.device context manager costs several microseconds per call, a third of the host time of a one-pair inference call)."""
    return _NO_SWITCH if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_WS_CACHE = {}   # (device, stream, bytes) -> buffer, for calls no backward pass will follow


def _workspace(nbytes, dev, cache=False):
    """Scratch for one entry-point call.  With `cache` (the caller runs under no_grad: nothing will read the buffer after the
    call's own kernels) the buffer of the last call of that size on this device and stream is handed out again: work on one
    stream is ordered, so two calls never use it at once, and the one-pair inference loops of the sequence schedulers
    (thousands of identical calls) stop going through the allocator."""
    n = max(int(nbytes), 256)
    if not cache or torch.cuda.is_current_stream_capturing():   # a buffer made during graph capture belongs to that graph's pool
        return torch.empty(n, dtype=torch.uint8, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, n)
    buf = _WS_CACHE.get(key)
    if buf is None:
        if len(_WS_CACHE) >= 32:
            _WS_CACHE.clear()
        buf = _WS_CACHE[key] = torch.empty(n, dtype=torch.uint8, device=dev)
    return buf


def _planes(t, h, w):
    """Return t itself if its last two dims are a dense (h,w) plane, else a contiguous copy."""
    if t.stride(-1) == 1 and t.stride(-2) == w:
        return t
    return t.contiguous()


# ----------------------------------------------------------------------------------------------- A1 corr4d
class Corr4dFn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, x_t, v_t, x_r, v_r, precision):
        dev = _require_device(x_t, v_t, x_r, v_r)
        b, c, f, h, w = x_r.shape
        if x_t.shape != (b, c, h, w):
            raise ValueError("x_target_feats %s does not match x_ref_feats %s" % (tuple(x_t.shape), tuple(x_r.shape)))
        p = h * w
        xt, xr = x_t.contiguous(), x_r.contiguous()
        vt = None if v_t is None else v_t.expand(b, 1, h, w).contiguous()
        vr = None if v_r is None else v_r.expand(b, 1, f, h, w).contiguous()
        out = torch.empty((b, f, h, w, h, w), dtype=torch.float32, device=dev)
        lib = _lib.lib()
        ws = _workspace(lib.tcl_corr4d_fwd_workspace_bytes(b, c, f, p, precision), dev, cache=not any(ctx.needs_input_grad))
        with _on(dev):
            _lib.call("tcl_corr4d_fwd", _p(xt), _p(vt), _p(xr), _p(vr), _p(out), b, c, f, p, precision,
                      _p(ws), ws.numel(), _stream(dev))
        ctx.save_for_backward(xt, vt, xr, vr, ws)
        ctx.dims = (b, c, f, h, w, precision)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        xt, vt, xr, vr, ws = ctx.saved_tensors
        b, c, f, h, w, precision = ctx.dims
        p = h * w
        dev = xt.device
        g = grad_out.contiguous()
        need_t, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        g_t = torch.empty_like(xt) if need_t else None
        g_r = torch.empty_like(xr) if need_r else None
        if need_t or need_r:
            lib = _lib.lib()
            ws2 = _workspace(lib.tcl_corr4d_bwd_workspace_bytes(b, c, f, p, precision), dev)
            with _on(dev):
                _lib.call("tcl_corr4d_bwd", _p(g), _p(xt), _p(vt), _p(xr), _p(vr), _p(ws), _p(g_t), _p(g_r),
                          b, c, f, p, precision, _p(ws2), ws2.numel(), _stream(dev))
        return g_t, None, g_r, None, None


def corr4d(x_t, v_t, x_r, v_r, precision=None):
    """precision None: selected by the input dtype: fp32 tensors -> exact fp32 MFMA; bf16 / fp16 tensors ("bf16 is
    a build extension selected by input dtype") -> the bf16-MFMA variant with fp32 accumulation, result in the input dtype
    (what torch.matmul of the reference returns for such inputs)."""
    low = x_r.dtype in (torch.bfloat16, torch.float16) and not torch.is_autocast_enabled("cuda")
    if not torch.is_autocast_enabled("cuda") and x_t.dtype != x_r.dtype:
        raise TclError("x_target_feats (%s) and x_ref_feats (%s) must share a dtype" % (x_t.dtype, x_r.dtype))
    if low:
        # fp16 tensors take the same bf16-operand path (8-bit mantissa, 1e-2 bar: coarser than an fp16 matmul, documented in
        # DESIGN.md section 1); the fp32 accumulator is rounded to the input dtype on the way out
        in_dtype = x_r.dtype
        f32 = lambda t: None if t is None else t.float()
        out = Corr4dFn.apply(f32(x_t), f32(v_t), f32(x_r), f32(v_r), _lib.PREC_BF16 if precision is None else precision)
        return out.to(in_dtype)
    return Corr4dFn.apply(x_t, v_t, x_r, v_r, _lib.PREC_F32 if precision is None else precision)


def corr4d_argmax_flow(x_t, v_t, x_r, v_r, precision=None):
    """corr_to_flow(compute_masked_4d_correlation(...)) without materialising the volume (square maps only).  `precision`
    (default fp32) selects the contraction's arithmetic as in corr4d: with bf16 / bf16x3 the fused kernel runs on the bf16
    matrix pipe, unbound from the fp32 volume store."""
    dev = _require_device(x_t, v_t, x_r, v_r)
    b, c, f, h, w = x_r.shape
    if h != w:
        raise ValueError("corr_to_flow needs square feature maps (got %dx%d)" % (h, w))
    p = h * w
    prec = _lib.PREC_F32 if precision is None else precision
    xt, xr = x_t.detach().contiguous(), x_r.detach().contiguous()
    vt = None if v_t is None else v_t.expand(b, 1, h, w).contiguous()
    vr = None if v_r is None else v_r.expand(b, 1, f, h, w).contiguous()
    flow = torch.empty((b, f, h, w, 2), dtype=torch.float32, device=dev)
    ws = _workspace(_lib.lib().tcl_corr4d_argmax_flow_workspace_bytes(b, c, f, p, prec), dev, cache=True)
    with _on(dev):
        _lib.call("tcl_corr4d_argmax_flow", _p(xt), _p(vt), _p(xr), _p(vr), _p(flow), b, c, f, h, prec, _p(ws), ws.numel(), _stream(dev))
    return flow


# ----------------------------------------------------------------------------------------------- A2 corr1x1
class Corr1x1Fn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, f1, f2, m1, m2):
        dev = _require_device(f1, f2, m1, m2)
        b, c, h, w = f1.shape
        f1c, f2c = f1.contiguous(), f2.contiguous()
        m1c, m2c = m1.expand(b, 1, h, w).contiguous(), m2.expand(b, 1, h, w).contiguous()
        out = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("tcl_corr1x1_fwd", _p(f1c), _p(f2c), _p(m1c), _p(m2c), _p(out), b, c, h * w, _stream(dev))
        ctx.save_for_backward(f1c, f2c, m1c, m2c)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        f1, f2, m1, m2 = ctx.saved_tensors
        b, c, h, w = f1.shape
        dev = f1.device
        g = grad_out.contiguous()
        g1 = torch.empty_like(f1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(f2) if ctx.needs_input_grad[1] else None
        with _on(dev):
            _lib.call("tcl_corr1x1_bwd", _p(g), _p(f1), _p(f2), _p(m1), _p(m2), _p(g1), _p(g2), b, c, h * w, _stream(dev))
        return g1, g2, None, None


# ----------------------------------------------------------------------------------------------- A3/A4 warp
def _frame_runs(frames, f_total):
    """[source frame of output frame 0, 1, ...] -> [(source start, output start, count)] of consecutive runs."""
    runs, o = [], 0
    while o < len(frames):
        n = 1
        while o + n < len(frames) and frames[o + n] == frames[o] + n:
            n += 1
        if frames[o] < 0 or frames[o] + n > f_total:
            raise IndexError("frame index %d outside [0, %d)" % (frames[o], f_total))
        runs.append((int(frames[o]), o, n))
        o += n
    return runs


_MAX_FRAME_RUNS = 4   # more runs than this: gather the frames once (one copy) instead of one launch per run


class WarpFn(torch.autograd.Function):
    """x (B,C,F,H,W) [any strides with dense planes], v (B,1,F,H,W) | None, and either grid (B*F',H,W,2) or
    theta (B*F',2,3) -> x_out as a (B,C,F',H,W) view of a (B,F',C,H,W) buffer (what the reference returns),
    v_out likewise.  Gradients flow to x and to the grid / theta; v is data.

    `frames` (a list of F' source frame indices, default: all F in order) selects which frame of x / v each output frame
    samples: `WarpFn(x, ..., frames=r_list)` == `WarpFn(x[:, :, r_list], ...)` without the gathered copy of the frames (the
    `x[:, :, r_list]` of models/cpn.py:283-290 is an index kernel as long as the warp itself): every run of consecutive
    indices is one launch on a strided view of x that writes its frames of the shared output (the C ABI takes strides)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, v, grid, theta, align_corners, v_mode, frames=None, v_is_hole=False):
        """`v_is_hole`: v holds the reference's HOLE masks m; the kernel samples 1 - m (flipped per tap), so the `1 - m`
        tensor of models/cpn.py:288 / thesis_dfpn/runner.py:157 is never made."""
        dev = _require_device(x, v, grid, theta)
        b, c, f_src, h, w = x.shape
        runs = [(0, 0, f_src)] if frames is None else _frame_runs([int(i) for i in frames], f_src)
        ctx.gathered = None
        if len(runs) > _MAX_FRAME_RUNS:
            ctx.gathered = (torch.as_tensor([int(i) for i in frames], device=dev), f_src)   # backward scatters g_x back
            x, v = x[:, :, list(frames)], (None if v is None else v[:, :, list(frames)])
            f_src, runs = x.shape[2], [(0, 0, x.shape[2])]
        f = sum(n for _, _, n in runs)
        xs = _planes(x, h, w)
        vs = None if v is None else _planes(v, h, w)
        gridc = None if grid is None else grid.contiguous()
        thetac = None if theta is None else theta.contiguous()
        x_out = torch.empty((b, f, c, h, w), dtype=torch.float32, device=dev)
        v_out = torch.empty((b, f, 1, h, w), dtype=torch.float32, device=dev) if vs is not None else None
        whole = len(runs) == 1 and runs[0][2] == f
        with _on(dev):
            for src, dst, n in runs:
                # the run's positions: rows b * F' + dst .. + n of grid / theta (a small dense copy unless the run is everything)
                g_run = None if gridc is None else (gridc if whole else gridc.view(b, f, h, w, 2)[:, dst:dst + n].contiguous())
                t_run = None if thetac is None else (thetac if whole else thetac.view(b, f, 2, 3)[:, dst:dst + n].contiguous())
                xr, xo = xs[:, :, src:src + n], x_out[:, dst:dst + n]
                vr = None if vs is None else vs[:, :, src:src + n]
                vo = None if v_out is None else v_out[:, dst:dst + n]
                _lib.call("tcl_warp_fwd_fused",
                          _p(xr), xs.stride(0), xs.stride(2), xs.stride(1),
                          _p(vr), vs.stride(0) if vs is not None else 0, vs.stride(2) if vs is not None else 0, int(bool(v_is_hole)),
                          _p(g_run), _p(t_run),
                          _p(xo), x_out.stride(0), x_out.stride(1), x_out.stride(2),
                          _p(vo), v_out.stride(0) if v_out is not None else 0, v_out.stride(1) if v_out is not None else 0,
                          None, 0, 0, None, 0, 0,
                          b, n, c, h, w, int(align_corners), v_mode if vs is not None else _lib.VMODE_NONE, _stream(dev))
        ctx.save_for_backward(xs, gridc, thetac)
        ctx.align_corners = int(align_corners)
        ctx.x_shape = tuple(xs.shape)
        ctx.runs, ctx.f_out = runs, f
        if v_out is not None:
            v_ret = v_out.transpose(1, 2)
            ctx.mark_non_differentiable(v_ret)
        else:
            v_ret = None
        return x_out.transpose(1, 2), v_ret

    @staticmethod
    @_bwd
    def backward(ctx, g_x_out, _g_v_out):
        xs, gridc, thetac = ctx.saved_tensors
        b, c, f_src, h, w = ctx.x_shape
        runs, f = ctx.runs, ctx.f_out
        dev = xs.device
        go = _planes(g_x_out, h, w)
        need_x, need_grid, need_theta = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        g_x = torch.zeros((b, c, f_src, h, w), dtype=torch.float32, device=dev) if need_x else None
        g_grid = torch.empty_like(gridc) if (need_grid and gridc is not None) else None
        g_theta = torch.empty_like(thetac) if (need_theta and thetac is not None) else None
        whole = len(runs) == 1 and runs[0][2] == f
        if need_x or g_grid is not None or g_theta is not None:
            with _on(dev):
                for src, dst, n in runs:
                    g_run = None if gridc is None else (gridc if whole else gridc.view(b, f, h, w, 2)[:, dst:dst + n].contiguous())
                    t_run = None if thetac is None else (thetac if whole else thetac.view(b, f, 2, 3)[:, dst:dst + n].contiguous())
                    gg_run = None if g_grid is None else (g_grid if whole else torch.empty_like(g_run))
                    gt_run = None if g_theta is None else (g_theta if whole else torch.empty_like(t_run))
                    ws = _workspace(_lib.lib().tcl_warp_bwd_workspace_bytes(b, n, h, w, 1 if g_theta is not None else 0), dev)
                    _lib.call("tcl_warp_bwd",
                              _p(go[:, :, dst:dst + n]), go.stride(0), go.stride(2), go.stride(1),
                              _p(xs[:, :, src:src + n]), xs.stride(0), xs.stride(2), xs.stride(1),
                              _p(g_run), _p(t_run),
                              _p(g_x[:, :, src:src + n]) if need_x else None, g_x.stride(0) if need_x else 0, g_x.stride(2) if need_x else 0,
                              g_x.stride(1) if need_x else 0,
                              _p(gg_run), _p(gt_run),
                              b, n, c, h, w, ctx.align_corners, _p(ws), ws.numel(), _stream(dev))
                    if not whole:
                        if g_grid is not None:
                            g_grid.view(b, f, h, w, 2)[:, dst:dst + n] = gg_run
                        if g_theta is not None:
                            g_theta.view(b, f, 2, 3)[:, dst:dst + n] = gt_run
        if g_x is not None and ctx.gathered is not None:
            idx, f_all = ctx.gathered
            g_x = torch.zeros((b, c, f_all, h, w), dtype=torch.float32, device=dev).index_add_(2, idx, g_x)
        return g_x, None, g_grid, g_theta, None, None, None, None


def warp_infer(x_ref, m_ref, m_target, grid=None, theta=None, align_corners=True, v_mode=None):
    """Inference-only warp of reference frames with everything around it in ONE launch (tcl_warp_fwd_fused):
    x_ref (B,3,F,H,W), m_ref (B,1,F,H,W) HOLE masks, m_target (B,1,H,W) hole mask, positions from `grid` (B*F,H,W,2) or
    `theta` (B*F,2,3) -> x_aligned (B,3,F,H,W), v_aligned = warp(1 - m_ref) (B,1,F,H,W), v_map = clamp(v_aligned - (1 - m_target), 0, 1).
    Not differentiable (the callers run it under no_grad); three-channel frames."""
    dev = _require_device(x_ref, m_ref, grid, theta, m_target)
    b, c, f, h, w = x_ref.shape
    if c != 3 or w < 2:
        raise ValueError("warp_infer takes three-channel frames at least two pixels wide")
    if (grid is None) == (theta is None):
        raise ValueError("exactly one of grid / theta")
    xs, ms = _planes(x_ref.detach(), h, w), _planes(m_ref.detach(), h, w)
    mt = m_target.detach().reshape(b, h, w)
    mt = mt if (mt.stride(2) == 1 and mt.stride(1) == w) else mt.contiguous()
    gridc = None if grid is None else grid.detach().reshape(b * f, h, w, 2).contiguous()
    thetac = None if theta is None else theta.detach().reshape(b * f, 2, 3).contiguous()
    # one allocation, planes [frames (3), visibility, v_map] per batch item: the three results are plain channel slices of it (the C
    # ABI takes strides, so no transposed views are needed on the way out)
    out = torch.empty((b, 5, f, h, w), dtype=torch.float32, device=dev)
    sb, sc, sf = 5 * f * h * w, f * h * w, h * w
    base = out.data_ptr()
    with _on(dev):
        _lib.call("tcl_warp_fwd_fused",
                  _p(xs), xs.stride(0), xs.stride(2), xs.stride(1), _p(ms), ms.stride(0), ms.stride(2), 1,
                  _p(gridc), _p(thetac),
                  ctypes.c_void_p(base), sb, sf, sc, ctypes.c_void_p(base + 3 * sc * 4), sb, sf,
                  _p(mt), mt.stride(0), 1, ctypes.c_void_p(base + 4 * sc * 4), sb, sf,
                  b, f, 3, h, w, int(bool(align_corners)), _lib.VMODE_NEAREST if v_mode is None else v_mode, _stream(dev))
    return out[:, 0:3], out[:, 3:4], out[:, 4:5]


def align_infer(x_ref, m_ref, flow, m_target):
    """The one-pair inference unit after the alignment network (thesis_dfpn/runner.py:157-158):
    align_set(x_ref, 1 - m_ref, flow) and v_map = clamp(v_aligned - (1 - m_target), 0, 1), one launch."""
    return warp_infer(x_ref, m_ref, m_target, grid=flow, align_corners=True, v_mode=_lib.VMODE_NEAREST)


class AffineGridFn(torch.autograd.Function):
    """theta (N,2,3) -> grid (N,H,W,2) = F.affine_grid(theta, (N,*,H,W), align_corners)."""

    @staticmethod
    @_fwd
    def forward(ctx, theta, h, w, align_corners):
        dev = _require_device(theta)
        n = theta.shape[0]
        if tuple(theta.shape[1:]) != (2, 3):
            raise ValueError("affine_grid expects theta (N,2,3), got %s" % (tuple(theta.shape),))
        tc = theta.contiguous()
        grid = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("tcl_affine_grid", _p(tc), _p(grid), n, h, w, int(bool(align_corners)), _stream(dev))
        ctx.meta = (h, w, bool(align_corners))
        return grid

    @staticmethod
    @_bwd
    def backward(ctx, g):
        # grid = base @ theta^T with a constant base grid: a (6 x HW) reduction per image; data-side op, torch suffices
        h, w, ac = ctx.meta
        base = torch.nn.functional.affine_grid(torch.eye(2, 3, device=g.device).unsqueeze(0), (1, 1, h, w), align_corners=ac)
        base = torch.cat([base, torch.ones_like(base[..., :1])], -1).reshape(1, h * w, 3)
        return torch.matmul(g.reshape(g.shape[0], h * w, 2).transpose(1, 2), base), None, None, None


def vmap(v_aligned, v_target):
    """clamp(v_aligned - v_target.unsqueeze(2), 0, 1): (B,1,F,H,W), (B,1,H,W) -> (B,1,F,H,W)."""
    dev = _require_device(v_aligned, v_target)
    b, _, f, h, w = v_aligned.shape
    va = v_aligned.transpose(1, 2).contiguous()   # (B,F,1,H,W)
    vt = v_target.contiguous()
    out = torch.empty_like(va)
    with _on(dev):
        _lib.call("tcl_vmap", _p(va), _p(vt), _p(out), b, f, h * w, _stream(dev))
    return out.transpose(1, 2)


# ----------------------------------------------------------------------------------------------- A5 masked softmax
def _softmax_layout(vec, mask, dim):
    dim = dim % vec.dim()
    shape = tuple(vec.shape)
    outer = 1
    for s in shape[:dim]:
        outer *= s
    inner = 1
    for s in shape[dim + 1:]:
        inner *= s
    mask = mask.float()
    while mask.dim() < vec.dim():
        mask = mask.unsqueeze(0)
    mshape = tuple(mask.shape)
    # fast layout: mask = vec.shape[:k] + (1,)*(dim-k) + vec.shape[dim:]  -> one mask row serves O/Om outer rows
    k = 0
    while k < dim and mshape[k] == shape[k]:
        k += 1
    if all(s == 1 for s in mshape[k:dim]) and mshape[dim:] == shape[dim:]:
        om = 1
        for s in shape[:k]:
            om *= s
        return dim, outer, om, shape[dim], inner, mask.contiguous()
    return dim, outer, outer, shape[dim], inner, mask.expand(shape).contiguous()


class MaskedSoftmaxFn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, vec, mask, dim):
        dev = _require_device(vec)
        if not mask.is_cuda:
            raise TclError("tcl ops run on the GPU only (mask on %s)" % mask.device)
        dim, o, om, r, i, maskc = _softmax_layout(vec, mask, dim)
        vc = vec.contiguous()
        out = torch.empty_like(vc)
        with _on(dev):
            _lib.call("tcl_masked_softmax_fwd", _p(vc), _p(maskc), _p(out), o, om, r, i, _stream(dev))
        ctx.save_for_backward(vc, maskc)
        ctx.layout = (o, om, r, i)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        vc, maskc = ctx.saved_tensors
        o, om, r, i = ctx.layout
        dev = vc.device
        g = grad_out.contiguous()
        gv = torch.empty_like(vc)
        with _on(dev):
            _lib.call("tcl_masked_softmax_bwd", _p(g), _p(vc), _p(maskc), _p(gv), o, om, r, i, _stream(dev))
        return gv, None, None


# ----------------------------------------------------------------------------------------------- A6-A9 context matching
class ContextMatchingFn(torch.autograd.Function):
    """c_feats (B,C,R+1,h,w), v_t (B,1,H,W), v_aligned (B,1,R,H,W) -> p_in, c_mask (B,1,h,w), c_match (B,R,h,w)."""

    @staticmethod
    @_fwd
    def forward(ctx, c_feats, v_t, v_aligned, normalize, variant):
        dev = _require_device(c_feats, v_t, v_aligned)
        b, c, f, h, w = c_feats.shape
        r = f - 1
        if r < 1:
            raise ValueError("context matching needs at least one reference frame")
        H, W = v_t.shape[-2], v_t.shape[-1]
        cf = c_feats.contiguous()
        vt = v_t.reshape(b, H, W).contiguous()
        va = v_aligned.reshape(b, r, H, W).contiguous()
        cch = c if variant == _lib.CTX_COMPLETE else 2 * c
        p_in = torch.empty((b, cch + 1, h, w), dtype=torch.float32, device=dev)
        c_mask = torch.empty((b, 1, h, w), dtype=torch.float32, device=dev)
        c_match = torch.empty((b, r, h, w), dtype=torch.float32, device=dev)
        ws = _workspace(_lib.lib().tcl_ctx_workspace_bytes(b, c, r, h, w), dev, cache=not any(ctx.needs_input_grad))
        with _on(dev):
            _lib.call("tcl_ctx_fwd", _p(cf), _p(vt), _p(va), _p(p_in), _p(c_mask), _p(c_match), b, c, r, h, w, H, W,
                      variant, int(bool(normalize)), _p(ws), ws.numel(), _stream(dev))
        ctx.save_for_backward(cf, c_match, ws)
        ctx.meta = (b, c, r, h, w, variant, int(bool(normalize)))
        ctx.mark_non_differentiable(c_match)
        return p_in, c_mask, c_match

    @staticmethod
    @_bwd
    def backward(ctx, g_p_in, g_c_mask, _g_c_match):
        cf, c_match, ws = ctx.saved_tensors
        b, c, r, h, w, variant, normalize = ctx.meta
        dev = cf.device
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        gp = g_p_in.contiguous()
        gm = None if g_c_mask is None else g_c_mask.contiguous()
        g_cf = torch.empty_like(cf)
        ws2 = _workspace(_lib.lib().tcl_ctx_bwd_workspace_bytes(b, c, r, h, w), dev)
        with _on(dev):
            _lib.call("tcl_ctx_bwd", _p(gp), _p(gm), _p(cf), _p(c_match), _p(ws), _p(g_cf), b, c, r, h, w, variant, normalize,
                      _p(ws2), ws2.numel(), _stream(dev))
        return g_cf, None, None, None, None


def ref_importance(c_match, v_t, middle_is_vt):
    """c_match (B,R,h,w), v_t (B,1,H,W) -> (B,3,R+1,H,W) visualisation tensor (not differentiable)."""
    dev = _require_device(c_match, v_t)
    b, r, h, w = c_match.shape
    H, W = v_t.shape[-2], v_t.shape[-1]
    cm = c_match.detach().contiguous()
    vt = v_t.detach().reshape(b, H, W).contiguous()
    out = torch.empty((b, 3, r + 1, H, W), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("tcl_ref_importance", _p(cm), _p(vt), _p(out), b, r, h, w, H, W, int(bool(middle_is_vt)), _stream(dev))
    return out


def resize_mask(v, size):
    """(F.interpolate(v (N,1,H,W), size, 'bilinear', align_corners=False) > 0.5).float()"""
    dev = _require_device(v)
    n, ch, H, W = v.shape
    vc = v.contiguous()
    out = torch.empty((n, ch, size[0], size[1]), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("tcl_resize_mask", _p(vc), _p(out), n * ch, H, W, size[0], size[1], _stream(dev))
    return out


def dilate_cross3(img):
    """3x3-cross dilation over the last two axes."""
    dev = _require_device(img)
    H, W = img.shape[-2], img.shape[-1]
    ic = img.contiguous()
    out = torch.empty_like(ic)
    with _on(dev):
        _lib.call("tcl_dilate_cross3", _p(ic), _p(out), ic.numel() // (H * W), H, W, _stream(dev))
    return out


# ----------------------------------------------------------------------------------------------- A10-A13 flow utilities
def corr_to_flow(corr):
    dev = _require_device(corr)
    b, f, s = corr.shape[0], corr.shape[1], corr.shape[2]
    if tuple(corr.shape[2:]) != (s, s, s, s):
        raise ValueError("corr_to_flow expects (B,F,S,S,S,S), got %s" % (tuple(corr.shape),))
    cc = corr.detach().contiguous()
    flow = torch.empty((b, f, s, s, 2), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("tcl_corr_to_flow", _p(cc), _p(flow), b * f, s, _stream(dev))
    return flow


class FlowAddIdentityFn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, flow, sign):
        dev = _require_device(flow)
        b, f, h, w, two = flow.shape
        fc = flow.contiguous()
        out = torch.empty_like(fc)
        with _on(dev):
            _lib.call("tcl_flow_add_identity", _p(fc), _p(out), b * f, h, w, sign, _stream(dev))
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        return g, None


class CropFlowFn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, flow, crop_size, crop_position):
        dev = _require_device(flow)
        b, f, h, w, two = flow.shape
        ch, cw = int(crop_size[0]), int(crop_size[1])
        py, px = int(crop_position[0]), int(crop_position[1])
        fc = flow.contiguous()
        out = torch.empty((b, f, ch, cw, 2), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("tcl_crop_flow", _p(fc), _p(out), b * f, h, w, ch, cw, py, px, _stream(dev))
        ctx.meta = (b, f, h, w, ch, cw, py, px)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        b, f, h, w, ch, cw, py, px = ctx.meta
        gf = g.new_zeros((b, f, h, w, 2))
        scale = g.new_tensor([w / cw, h / ch])
        gf[:, :, py:py + ch, px:px + cw] = g * scale
        return gf, None, None


class ResizeFlowFn(torch.autograd.Function):
    @staticmethod
    @_fwd
    def forward(ctx, flow, size):
        dev = _require_device(flow)
        b, f, h, w, two = flow.shape
        oh, ow = int(size[0]), int(size[1])
        fc = flow.contiguous()
        out = torch.empty((b, f, oh, ow, 2), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("tcl_resize_flow", _p(fc), _p(out), b * f, h, w, oh, ow, _stream(dev))
        ctx.meta = (b, f, h, w, oh, ow)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        # gradient of a nearest gather: scatter-add back (flows of the hot path are data here; torch ops suffice)
        b, f, h, w, oh, ow = ctx.meta
        sy = torch.clamp((torch.arange(oh, device=g.device, dtype=torch.float32) * (torch.tensor(float(h)) / oh).float()).floor().long(), max=h - 1)
        sx = torch.clamp((torch.arange(ow, device=g.device, dtype=torch.float32) * (torch.tensor(float(w)) / ow).float()).floor().long(), max=w - 1)
        gf = g.new_zeros((b, f, h, w, 2))
        idx = (sy[:, None] * w + sx[None, :]).reshape(-1)
        gf.view(b, f, h * w, 2).index_add_(2, idx, g.reshape(b, f, oh * ow, 2))
        return gf, None
