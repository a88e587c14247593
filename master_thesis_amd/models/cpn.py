"""Drop-in for the hot-path parts of the reference's models/cpn.py: masked_softmax, the two context-matching
modules, the border helpers and the affine alignment warp of CPNet.align.  Same names and signatures; the conv
encoders / decoder of CPNet are not part of this package (they stay stock PyTorch in the reference)."""
import torch
import torch.nn as nn

from .. import _lib, _ops


def masked_softmax(vec, mask, dim):
    """Softmax of `vec` over `dim` restricted to mask == 1 (zero-safe).  Mirrors models/cpn.py:7-15."""
    return _ops.MaskedSoftmaxFn.apply(vec, mask, dim)


def add_match_border(v_t, c_match):
    """ref_importance of the default variant: c_match[:, 0] (B,R,h,w) upsampled to v_t's size, a zero plane inserted in
    the middle, the hole border painted red -> (B,3,R+1,H,W).  Mirrors models/cpn.py:23-36, on the device."""
    return _ops.ref_importance(c_match[:, 0], v_t, middle_is_vt=False)


def add_border(x, mask):
    """Paint the outer border of `mask` (B,1,H,W) red on x (B,3,F,H,W).  Mirrors models/cpn.py:39-44, on the device."""
    border = ((_ops.dilate_cross3(mask) - mask) > 0).float().unsqueeze(2)
    red = torch.tensor([1.0, 0.0, 0.0], device=x.device).view(1, 3, 1, 1, 1)
    return x * (1 - border) + border * red


class _ContextMatching(nn.Module):
    _variant = _lib.CTX_DEFAULT
    _middle_is_vt = False

    def forward(self, c_feats, v_t, v_aligned, normalize=True):
        b, c_c, f, h, w = c_feats.size()
        p_in, c_mask, c_match = _ops.ContextMatchingFn.apply(c_feats, v_t, v_aligned, normalize, self._variant)
        ref_importance = _ops.ref_importance(c_match, v_t, self._middle_is_vt)
        return p_in, c_mask, ref_importance


class CPNContextMatchingDefault(_ContextMatching):
    """(c_feats (B,C,R+1,h,w), v_t (B,1,H,W), v_aligned (B,1,R,H,W), normalize) ->
    (cat[target feats, aggregated refs, c_mask] (B,2C+1,h,w), c_mask (B,1,h,w), ref_importance (B,3,R+1,H,W)).
    Mirrors models/cpn.py:90-135."""
    _variant = _lib.CTX_DEFAULT
    _middle_is_vt = False


class CPNContextMatchingComplete(_ContextMatching):
    """Hole-only variant -> (cat[target*v_t + aggregated refs, c_mask] (B,C+1,h,w), c_mask, ref_importance).
    Mirrors models/cpn.py:138-195."""
    _variant = _lib.CTX_COMPLETE
    _middle_is_vt = True


def affine_align(x, m, y, theta_rt, r_list, training=False):
    """The affine_grid + grid_sample block of CPNet.align (models/cpn.py:280-295) / CPNOriginal.align
    (models/cpn_original.py:276-284): theta_rt (B*R,2,3) -> x_aligned (B,C,R,H,W), v_aligned (B,1,R,H,W) in {0,1},
    y_aligned | None.  The sampling grid is derived from theta inside the kernel and never stored."""
    x_refs = x[:, :, r_list]
    v_refs = 1 - m[:, :, r_list]
    x_aligned, v_aligned = _ops.WarpFn.apply(x_refs, v_refs, None, theta_rt, False, _lib.VMODE_BILINEAR_THRESHOLD)
    y_aligned = None
    if not training and y is not None:
        y_aligned, _ = _ops.WarpFn.apply(y[:, :, r_list], None, None, theta_rt, False, _lib.VMODE_NONE)
    return x_aligned, v_aligned, y_aligned
