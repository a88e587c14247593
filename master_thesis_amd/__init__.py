"""master_thesis_amd - MI355X (gfx950) implementation of the frame-alignment / temporal-copying hot path of
AlomdaElmasry/master_thesis: all-pairs feature correlation, flow / affine grid warp and masked cross-frame
context matching, as hand-written HIP kernels behind a C ABI (include/tcl.h, csrc/libtcl_hip.so).

The sub-modules mirror the reference's module and function names for this path:

    master_thesis_amd.utils.correlation   <- utils/correlation.py
    master_thesis_amd.utils.flow          <- utils/flow.py
    master_thesis_amd.models.cpn          <- models/cpn.py (context matching, helpers, CPNet around the HIP warp + matching)
    master_thesis_amd.models.thesis_alignment <- models/thesis_alignment.py (the alignment network around the HIP cost volume)
    master_thesis_amd.models.cpn_original <- models/cpn_original.py (CM_Module; CPNOriginal around the HIP warp + matching)
    master_thesis_amd.utils.movement      <- utils/movement.py (on the device)
    master_thesis_amd.runners / schedulers <- the static step functions and sequence schedulers of the three runners

`install_dropin()` registers them under the reference's own names (`utils.correlation`, `utils.flow`, and the
hot-path symbols of `models.cpn` / `models.cpn_original`), see INTEGRATION.md.
"""
from . import _lib  # noqa: F401
from ._lib import LIB_PATH, TclError, lib  # noqa: F401

__version__ = "0.1.0"


# functions the reference's DataLoader workers call on CPU tensors (thesis/dataset.py:281-289: crop_flow / resize_flow on
# the flows of utils/movement.py, in forked worker processes that must never touch the GPU)
_CPU_ROUTED = ("crop_flow", "resize_flow", "flow_abs_to_relative", "flow_relative_to_abs")


def _route_cpu_to(original, ours):
    """The drop-in for a function the reference also calls on CPU tensors: GPU tensors go to the HIP op, CPU tensors to the
    function that was replaced (the reference's own code, untouched).  This package has no CPU arithmetic of its own."""
    import functools

    @functools.wraps(ours)
    def routed(flow, *args, **kwargs):
        if getattr(flow, "is_cuda", False):
            return ours(flow, *args, **kwargs)
        return original(flow, *args, **kwargs)
    routed.__tcl_original__ = original
    return routed


def install_dropin(reference_modules=None):
    """Patch the reference's hot-path symbols in place with this package's implementations.

    reference_modules: optional dict with already-imported reference modules
        {'utils.correlation': mod, 'utils.flow': mod, 'models.cpn': mod, 'models.cpn_original': mod};
    by default they are looked up in sys.modules.  Only functions/classes on the hot path are replaced; the conv
    stacks, runners and data code of the reference stay untouched.

    CPU tensors: the reference's dataset code calls `crop_flow` / `resize_flow` (and through them the abs<->rel helpers) on
    CPU tensors inside DataLoader workers.  For those four functions the installed symbol keeps a handle to the function it
    replaces and hands CPU tensors to it, so the data pipeline runs as before; GPU tensors take the HIP kernels.  Every
    other op has no CPU route and raises TclError on a CPU tensor.
    """
    import sys

    from .models import cpn as _cpn
    from .models import cpn_original as _cpo
    from .utils import correlation as _corr
    from .utils import flow as _flow

    mods = reference_modules or sys.modules
    table = {
        "utils.correlation": (_corr, ["compute_masked_correlation", "compute_masked_4d_correlation"]),
        "utils.flow": (_flow, ["corr_to_flow", "flow_abs_to_relative", "flow_relative_to_abs", "crop_flow", "resize_flow", "align_set"]),
        "models.cpn": (_cpn, ["masked_softmax", "add_match_border", "add_border", "CPNContextMatchingDefault", "CPNContextMatchingComplete"]),
        "models.cpn_original": (_cpo, ["CM_Module"]),
    }
    patched = []
    for name, (src, symbols) in table.items():
        target = mods.get(name)
        if target is None:
            continue
        for sym in symbols:
            ours = getattr(src, sym)
            if sym in _CPU_ROUTED:
                original = getattr(target, sym, None)
                original = getattr(original, "__tcl_original__", original)      # installing twice keeps the first original
                if callable(original) and original is not ours:
                    ours = _route_cpu_to(original, ours)
            setattr(target, sym, ours)
            patched.append("%s.%s" % (name, sym))
    # CPNet keeps its own conv stacks and weights; its two hot methods are rebound so that the affine warp and (through the
    # patched CPNContextMatchingDefault of new instances) the context matching run on the HIP kernels
    ref_cpn = mods.get("models.cpn")
    if ref_cpn is not None and hasattr(ref_cpn, "CPNet"):
        for meth in ("align", "copy_and_paste"):
            setattr(ref_cpn.CPNet, meth, getattr(_cpn.CPNet, meth))
            patched.append("models.cpn.CPNet.%s" % meth)
    ref_cpo = mods.get("models.cpn_original")
    if ref_cpo is not None and hasattr(ref_cpo, "CPNOriginal"):   # same for the original network (what thesis_cpn runs)
        for meth in ("align", "copy_and_paste"):
            setattr(ref_cpo.CPNOriginal, meth, getattr(_cpo.CPNOriginal, meth))
            patched.append("models.cpn_original.CPNOriginal.%s" % meth)
    # the sequence schedulers (static methods of the runner classes), if those modules are loaded: same signatures,
    # and `schedulers.*_batched` is then available to callers that hold several sequences
    from . import schedulers as _sched
    runner_table = {
        "thesis_inpainting.runner": ("ThesisInpaintingRunner", {
            "inpainting_algorithm_ff": _sched.inpainting_algorithm_ff, "inpainting_algorithm_ip": _sched.inpainting_algorithm_ip,
            "inpainting_algorithm_cp": _sched.inpainting_algorithm_cp, "inpainting_algorithm_ff_indexes": _sched.inpainting_algorithm_ff_indexes,
            "inpainting_algorithm_ip_indexes": _sched.inpainting_algorithm_ip_indexes}),
        "thesis_cpn.runner": ("ThesisCPNRunner", {
            "inpainting_algorithm": _sched.inpainting_algorithm_cpn, "get_reference_frame_indexes": _sched.get_reference_frame_indexes}),
    }
    for name, (cls_name, methods) in runner_table.items():
        cls = getattr(mods.get(name), cls_name, None) if mods.get(name) is not None else None
        if cls is None:
            continue
        for meth, fn in methods.items():
            setattr(cls, meth, staticmethod(fn))
            patched.append("%s.%s.%s" % (name, cls_name, meth))
    return patched
