"""Drop-in for the reference's utils/correlation.py: same function names, arguments and return shapes, computed by
the HIP kernels of csrc/tcl_corr.hip through the C ABI (include/tcl.h)."""
from .. import _lib, _ops


def compute_masked_correlation(feats_1, feats_2, mask_1, mask_2):
    """Per-pixel masked, L2-normalised correlation.  (B,C,H,W) x2 and masks (B,1,H,W) -> (B,1,H,W).
    Mirrors utils/correlation.py:4-25 (tcl_corr1x1_fwd / _bwd)."""
    b, c, h, w = feats_1.size()
    return _ops.Corr1x1Fn.apply(feats_1, feats_2, mask_1, mask_2)


def compute_masked_4d_correlation(x_target_feats, v_target, x_ref_feats, v_ref, precision=None):
    """All-pairs masked correlation volume.  x_target_feats (B,C,H,W), v_target (B,1,H,W) | None,
    x_ref_feats (B,C,F,H,W), v_ref (B,1,F,H,W) | None -> (B,F,H,W,H,W), target pixel first.
    Mirrors utils/correlation.py:28-54 (tcl_corr4d_fwd / _bwd).  `precision` is an extension: None (default) follows the
    input dtype (fp32 tensors: exact fp32 MFMA; bf16 / fp16 tensors: bf16 MFMA, result in the input dtype); 'f32' / 'fp32',
    'bf16' or 'bf16x3' force a variant."""
    b, c, ref_n, h, w = x_ref_feats.size()
    prec = {None: None, "f32": _lib.PREC_F32, "fp32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "bf16x3": _lib.PREC_BF16X3}[precision]
    return _ops.corr4d(x_target_feats, v_target, x_ref_feats, v_ref, prec)
