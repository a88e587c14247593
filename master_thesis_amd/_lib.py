"""ctypes binding of libtcl_hip.so (the C ABI declared in include/tcl.h).

The library is built in-tree by `python -m master_thesis_amd.build` (or `__graft_entry__.build()`) into
master_thesis_amd/csrc/libtcl_hip.so.  There is NO fallback: if the library is missing or an entry point
fails, the caller gets an exception.  torch is imported first so that the library binds to the HIP runtime
torch already loaded (same SONAME), which makes torch's streams and device pointers valid inside it.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before the library: shares its libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCL_LIB_PATH") or os.path.join(_HERE, "csrc", "libtcl_hip.so")  # override: A/B experiments only

_c = ctypes
_P, _I64, _INT, _SZ = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t

# name -> (restype, argtypes); mirrors include/tcl.h one to one
SIGNATURES = {
    "tcl_version": (_INT, []),
    "tcl_last_error": (_c.c_char_p, []),
    "tcl_corr4d_fwd_workspace_bytes": (_SZ, [_I64] * 4 + [_INT]),
    "tcl_corr4d_fwd": (_INT, [_P] * 5 + [_I64] * 4 + [_INT, _P, _SZ, _P]),
    "tcl_corr4d_contract": (_INT, [_P] * 2 + [_I64] * 4 + [_INT, _P]),
    "tcl_corr4d_bwd_workspace_bytes": (_SZ, [_I64] * 4 + [_INT]),
    "tcl_corr4d_bwd": (_INT, [_P] * 8 + [_I64] * 4 + [_INT, _P, _SZ, _P]),
    "tcl_corr4d_argmax_flow_workspace_bytes": (_SZ, [_I64] * 4 + [_INT]),
    "tcl_corr4d_argmax_flow": (_INT, [_P] * 5 + [_I64] * 4 + [_INT, _P, _SZ, _P]),
    "tcl_corr1x1_fwd": (_INT, [_P] * 5 + [_I64] * 3 + [_P]),
    "tcl_corr1x1_bwd": (_INT, [_P] * 7 + [_I64] * 3 + [_P]),
    "tcl_warp_fwd": (_INT, [_P, _I64, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _I64]
                     + [_I64] * 5 + [_INT, _INT, _P]),
    "tcl_warp_fwd_fused": (_INT, [_P, _I64, _I64, _I64, _P, _I64, _I64, _INT, _P, _P, _P, _I64, _I64, _I64, _P, _I64, _I64,
                                 _P, _I64, _INT, _P, _I64, _I64] + [_I64] * 5 + [_INT, _INT, _P]),
    "tcl_warp_bwd_workspace_bytes": (_SZ, [_I64] * 4 + [_INT]),
    "tcl_warp_bwd": (_INT, [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _P, _P]
                     + [_I64] * 5 + [_INT, _P, _SZ, _P]),
    "tcl_affine_grid": (_INT, [_P] * 2 + [_I64] * 3 + [_INT, _P]),
    "tcl_vmap": (_INT, [_P] * 3 + [_I64] * 3 + [_P]),
    "tcl_masked_softmax_fwd": (_INT, [_P] * 3 + [_I64] * 4 + [_P]),
    "tcl_masked_softmax_bwd": (_INT, [_P] * 4 + [_I64] * 4 + [_P]),
    "tcl_ctx_workspace_bytes": (_SZ, [_I64] * 5),
    "tcl_ctx_fwd": (_INT, [_P] * 6 + [_I64] * 7 + [_INT, _INT, _P, _SZ, _P]),
    "tcl_ctx_fwd_static": (_INT, [_P] * 6 + [_I64] * 7 + [_INT, _INT, _P]),
    "tcl_ctx_bwd_workspace_bytes": (_SZ, [_I64] * 5),
    "tcl_ctx_bwd": (_INT, [_P] * 6 + [_I64] * 5 + [_INT, _INT, _P, _SZ, _P]),
    "tcl_ref_importance": (_INT, [_P] * 3 + [_I64] * 6 + [_INT, _P]),
    "tcl_resize_mask": (_INT, [_P] * 2 + [_I64] * 5 + [_P]),
    "tcl_dilate_cross3": (_INT, [_P] * 2 + [_I64] * 3 + [_P]),
    "tcl_corr_to_flow": (_INT, [_P] * 2 + [_I64] * 2 + [_P]),
    "tcl_flow_add_identity": (_INT, [_P] * 2 + [_I64] * 3 + [_INT, _P]),
    "tcl_crop_flow": (_INT, [_P] * 2 + [_I64] * 7 + [_P]),
    "tcl_resize_flow": (_INT, [_P] * 2 + [_I64] * 5 + [_P]),
}

PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2
CTX_DEFAULT, CTX_COMPLETE, CTX_CM_MODULE = 0, 1, 2
VMODE_NONE, VMODE_NEAREST, VMODE_BILINEAR_THRESHOLD = 0, 1, 2

_lib = None


class TclError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TclError(
                "libtcl_hip.so is missing (%s): build it with `python -m master_thesis_amd.build`; "
                "there is no CPU fallback" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI and the header disagree
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise TclError("%s failed (code %d): %s" % (what, rc, lib().tcl_last_error().decode()))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
