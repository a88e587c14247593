"""Drop-in for the reference's utils/flow.py: same function names, arguments and return shapes, computed by the
HIP kernels of csrc/tcl_warp.hip, tcl_flow.hip and tcl_corr.hip through the C ABI (include/tcl.h)."""
from .. import _lib, _ops


def corr_to_flow(corr):
    """(B,F,S,S,S,S) volume -> (B,F,S,S,2) absolute flow in [-1,1]: arg-max reference pixel of every target pixel,
    first maximum on ties.  Mirrors utils/flow.py:8-26 (tcl_corr_to_flow)."""
    b, f, s, *_ = corr.size()
    return _ops.corr_to_flow(corr)


def flow_abs_to_relative(flow):
    """(B,F,H,W,2) absolute grid -> displacement from the identity grid.  Mirrors utils/flow.py:29-46."""
    b, f, h, w, _ = flow.size()
    return _ops.FlowAddIdentityFn.apply(flow, -1)


def flow_relative_to_abs(flow_rel):
    """(B,F,H,W,2) displacement -> absolute grid.  Mirrors utils/flow.py:49-66."""
    b, f, h, w, _ = flow_rel.size()
    return _ops.FlowAddIdentityFn.apply(flow_rel, 1)


def crop_flow(flow, crop_size, crop_position):
    """Window of an absolute flow, re-expressed in the window's own [-1,1] frame.  Mirrors utils/flow.py:69-84."""
    b, f, h, w, _ = flow.size()
    return _ops.CropFlowFn.apply(flow, tuple(crop_size), tuple(crop_position))


def resize_flow(flow, size):
    """Legacy-nearest resize of a (B,F,H,W,2) grid to `size`.  Mirrors utils/flow.py:87-90."""
    b, f, h, w, _ = flow.size()
    return _ops.ResizeFlowFn.apply(flow, tuple(size))


def align_set(x, v, flow):
    """Warp frames x (B,C,F,H,W) bilinearly and visibility maps v (B,1,F,H,W) by nearest neighbour with the absolute
    grid `flow` (B,F,H,W,2), align_corners=True, zero padding -> (x_aligned, v_aligned).  One fused kernel reads the
    grid once for both.  Mirrors utils/flow.py:93-101 (tcl_warp_fwd / _bwd)."""
    b, c, f, h, w = x.size()
    return _ops.WarpFn.apply(x, v, flow.reshape(-1, h, w, 2), None, True, _lib.VMODE_NEAREST)
