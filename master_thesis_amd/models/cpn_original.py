"""Drop-in for the context-matching module of the reference's models/cpn_original.py."""
import torch.nn as nn

from .. import _lib, _ops


class CM_Module(nn.Module):
    """(values (B,C,T+1,h,w), tvmap (B,1,H,W), rvmaps (B,1,T,H,W)) -> (cat[t_feat, c_out, c_mask], c_mask).
    Mirrors models/cpn_original.py:187-243 (similarity always divided by the overlap area and by C)."""

    def masked_softmax(self, vec, mask, dim):
        return _ops.MaskedSoftmaxFn.apply(vec, mask, dim)

    def forward(self, values, tvmap, rvmaps):
        B, C, T, H, W = values.size()
        p_in, c_mask, _ = _ops.ContextMatchingFn.apply(values, tvmap, rvmaps, True, _lib.CTX_CM_MODULE)
        return p_in, c_mask
