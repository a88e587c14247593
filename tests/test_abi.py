"""CPU-only checks of the drop-in boundary: the library loads, exports every symbol include/tcl.h declares, the ctypes
table matches the header, argument validation works without a GPU, and the product never imports the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tcl.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tcl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ["tcl_corr4d_fwd", "tcl_corr4d_bwd", "tcl_warp_fwd", "tcl_warp_bwd", "tcl_ctx_fwd", "tcl_ctx_bwd",
                 "tcl_masked_softmax_fwd", "tcl_corr_to_flow", "tcl_last_error", "tcl_version"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from master_thesis_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared_symbols():
        assert hasattr(handle, sym), "libtcl_hip.so does not export %s" % sym


def test_ctypes_table_matches_header():
    from master_thesis_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_res, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = [p.strip() for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), "%s: header has %d parameters, ctypes table %d" % (name, len(params), len(args))


def test_version_and_validation_without_gpu():
    from master_thesis_amd import _lib
    lib = _lib.lib()
    assert lib.tcl_version() == 110
    # shape / argument errors are detected on the host before anything is launched
    assert lib.tcl_corr4d_fwd(None, None, None, None, None, 0, 1, 1, 1, 0, None, 0, None) == 1
    assert b"bad shape" in lib.tcl_last_error()
    assert lib.tcl_corr4d_fwd(None, None, None, None, None, 1, 1, 1, 1, 0, None, 0, None) == 2
    assert lib.tcl_corr4d_fwd(None, None, None, None, None, 1, 1, 1, 1, 7, None, 0, None) == 5
    assert lib.tcl_corr4d_fwd_workspace_bytes(16, 256, 1, 4096, 0) >= 2 * 16 * 256 * 4096 * 4
    assert lib.tcl_crop_flow(None, None, 1, 4, 4, 3, 3, 2, 2, None) == 1
    with pytest.raises(_lib.TclError):
        _lib.call("tcl_vmap", None, None, None, 1, 1, 1, None)


def test_cpu_tensors_are_rejected():
    import torch
    from master_thesis_amd import TclError
    from master_thesis_amd.utils import correlation, flow
    with pytest.raises(TclError):
        correlation.compute_masked_4d_correlation(torch.zeros(1, 2, 2, 2), None, torch.zeros(1, 2, 1, 2, 2), None)
    with pytest.raises(TclError):
        flow.align_set(torch.zeros(1, 3, 1, 4, 4), torch.zeros(1, 1, 1, 4, 4), torch.zeros(1, 1, 4, 4, 2))


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "master_thesis_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("the oracle is", ""), "%s mentions the oracle" % fn


def test_install_dropin_patches_reference_names():
    import types
    import master_thesis_amd as m
    fake = {k: types.ModuleType(k) for k in ["utils.correlation", "utils.flow", "models.cpn", "models.cpn_original"]}
    patched = m.install_dropin(fake)
    assert "utils.flow.align_set" in patched and "models.cpn_original.CM_Module" in patched
    assert fake["utils.correlation"].compute_masked_4d_correlation is m.utils.correlation.compute_masked_4d_correlation
    # runner classes, when loaded, get the schedulers as static methods
    fake["thesis_inpainting.runner"] = types.ModuleType("thesis_inpainting.runner")
    fake["thesis_inpainting.runner"].ThesisInpaintingRunner = type("ThesisInpaintingRunner", (), {})
    fake["thesis_cpn.runner"] = types.ModuleType("thesis_cpn.runner")
    fake["thesis_cpn.runner"].ThesisCPNRunner = type("ThesisCPNRunner", (), {})
    patched = m.install_dropin(fake)
    assert "thesis_inpainting.runner.ThesisInpaintingRunner.inpainting_algorithm_cp" in patched
    assert fake["thesis_inpainting.runner"].ThesisInpaintingRunner.inpainting_algorithm_ff_indexes(2, 5, 1, 20) == [1, 3, 0, 4]
    assert fake["thesis_cpn.runner"].ThesisCPNRunner.get_reference_frame_indexes(0, 5) == [2]
    # CPNet of the reference, when present, gets the two hot methods; its conv stacks and weights stay its own
    fake["models.cpn_original"].CPNOriginal = type("CPNOriginal", (), {})
    fake["models.cpn"].CPNet = type("CPNet", (), {})
    patched = m.install_dropin(fake)
    assert "models.cpn.CPNet.align" in patched and "models.cpn.CPNet.copy_and_paste" in patched
    assert fake["models.cpn"].CPNet.align is m.models.cpn.CPNet.align
    assert fake["models.cpn_original"].CPNOriginal.copy_and_paste is m.models.cpn_original.CPNOriginal.copy_and_paste


def test_install_dropin_routes_cpu_tensors_to_the_replaced_original():
    """thesis/dataset.py:281-289 calls crop_flow / resize_flow on CPU tensors in DataLoader workers: the installed symbols hand
    CPU tensors to the function they replaced, and keep raising where nothing was replaced."""
    import types
    import torch
    import master_thesis_amd as m
    calls = []
    fake = {"utils.flow": types.ModuleType("utils.flow")}

    def original_crop_flow(flow, crop_size, crop_position):
        calls.append(("crop_flow", tuple(flow.shape), tuple(crop_size), tuple(crop_position)))
        return "original result"

    def original_resize_flow(flow, size):
        calls.append(("resize_flow", tuple(flow.shape), tuple(size)))
        return "original resize"

    fake["utils.flow"].crop_flow = original_crop_flow
    fake["utils.flow"].resize_flow = original_resize_flow
    m.install_dropin(fake)
    cpu_flow = torch.zeros(1, 2, 8, 8, 2)
    assert fake["utils.flow"].crop_flow(cpu_flow, (4, 4), (1, 2)) == "original result"
    assert fake["utils.flow"].resize_flow(cpu_flow, (4, 4)) == "original resize"
    assert calls == [("crop_flow", (1, 2, 8, 8, 2), (4, 4), (1, 2)), ("resize_flow", (1, 2, 8, 8, 2), (4, 4))]
    # installing again keeps the FIRST original (no wrapper chains)
    m.install_dropin(fake)
    assert fake["utils.flow"].crop_flow.__tcl_original__ is original_crop_flow
    # no original to fall back to (the abs<->rel helpers were absent from the fake module): CPU tensors raise
    with pytest.raises(m.TclError):
        fake["utils.flow"].flow_abs_to_relative(cpu_flow)
    # align_set has no CPU route at all
    with pytest.raises(m.TclError):
        fake["utils.flow"].align_set(torch.zeros(1, 3, 1, 4, 4), torch.zeros(1, 1, 1, 4, 4), torch.zeros(1, 1, 4, 4, 2))


def test_shipped_library_reads_no_environment_switches():
    """A/B switches of the kernels are compile-time -D flags of side builds; the shipped sources must not read the environment."""
    csrc = os.path.join(ROOT, "master_thesis_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, fn)).read(), fn


def test_movement_simulator_names_its_gpu_only_contract():
    """a CPU frame must fail with an error that says what to do, before any kernel or library call."""
    import torch
    from master_thesis_amd import _lib
    from master_thesis_amd.utils.movement import MovementSimulator
    with pytest.raises(_lib.TclError, match="GPU only"):
        MovementSimulator(10, 0.1, 0.1).simulate_movement(torch.zeros(3, 8, 8), 3)
