"""tcl_ctx_fwd_static (the entry point that allocates nothing, include/tcl.h) against tcl_ctx_fwd, through the C ABI.

Same kernels, same pixel-range split (a function of the shape only), so the outputs must be bit-identical.  Two calls in a
row, the second with a smaller input (the static scratch keeps the first call's contents), then a third, larger one again;
a shape beyond the 4 MiB buffer must be rejected with TCL_ERR_WORKSPACE before anything is launched.
Collected last (file name) so that a failure here cannot hide the other GPU tests behind `pytest -x`.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(seed, B, C, R, h, w, H, W):
    g = torch.Generator().manual_seed(seed)
    cf = (0.3 * torch.randn(B, C, R + 1, h, w, generator=g)).to(DEV)
    v_t = (torch.rand(B, H // 4, W // 4, generator=g) < 0.6).float().repeat_interleave(4, 1).repeat_interleave(4, 2).to(DEV)
    v_al = (torch.rand(B, R, H // 4, W // 4, generator=g) < 0.7).float().repeat_interleave(4, 2).repeat_interleave(4, 3).to(DEV)
    return cf.contiguous(), v_t.contiguous(), v_al.contiguous()


def _run(entry, cf, v_t, v_al, variant, normalize):
    from master_thesis_amd import _lib
    B, C, F, h, w = cf.shape
    R, H, W = F - 1, v_t.shape[-2], v_t.shape[-1]
    cch = C if variant == _lib.CTX_COMPLETE else 2 * C
    p_in = torch.full((B, cch + 1, h, w), float("nan"), device=DEV)
    c_mask = torch.full((B, h, w), float("nan"), device=DEV)
    c_match = torch.full((B, R, h, w), float("nan"), device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if entry == "static":
        _lib.call("tcl_ctx_fwd_static", p(cf), p(v_t), p(v_al), p(p_in), p(c_mask), p(c_match), B, C, R, h, w, H, W, variant, normalize, stream)
    else:
        ws = torch.empty(_lib.lib().tcl_ctx_workspace_bytes(B, C, R, h, w), dtype=torch.uint8, device=DEV)
        _lib.call("tcl_ctx_fwd", p(cf), p(v_t), p(v_al), p(p_in), p(c_mask), p(c_match), B, C, R, h, w, H, W, variant, normalize,
                  p(ws), ws.numel(), stream)
    torch.cuda.synchronize()
    return p_in, c_mask, c_match


def test_static_entry_matches_workspace_entry_two_calls_in_a_row():
    from master_thesis_amd import _lib
    shapes = [
        (1, 4, 128, 10, 30, 53, 240, 424, _lib.CTX_DEFAULT, 0),     # C4 harness shape
        (2, 2, 16, 3, 6, 8, 24, 32, _lib.CTX_COMPLETE, 1),          # smaller: stale scratch of the first call must not matter
        (3, 32, 128, 10, 30, 53, 240, 424, _lib.CTX_CM_MODULE, 1),  # the full C4 batch (2.7 MiB of scratch)
        (4, 1, 5, 20, 5, 7, 20, 28, _lib.CTX_DEFAULT, 1),           # R > 16: two reference passes
    ]
    for seed, B, C, R, h, w, H, W, variant, normalize in shapes:
        cf, v_t, v_al = _inputs(seed, B, C, R, h, w, H, W)
        got = _run("static", cf, v_t, v_al, variant, normalize)
        want = _run("workspace", cf, v_t, v_al, variant, normalize)
        for a, b, name in zip(got, want, ("p_in", "c_mask", "c_match")):
            assert not torch.isnan(a).any(), name
            assert torch.equal(a, b), "%s differs for shape %s" % (name, (B, C, R, h, w))


def test_static_entry_rejects_shapes_beyond_its_buffer():
    from master_thesis_amd import _lib
    lib = _lib.lib()
    B, C, R, h, w = 64, 128, 10, 60, 106          # 64 targets at CPNOriginal's stride-4 maps: far beyond 4 MiB
    assert lib.tcl_ctx_workspace_bytes(B, C, R, h, w) > (4 << 20)
    one = ctypes.c_void_p(256)                    # never dereferenced: the size check comes first
    rc = lib.tcl_ctx_fwd_static(one, one, one, one, one, one, B, C, R, h, w, 240, 424, 0, 0, None)
    assert rc == 3 and b"static buffer" in lib.tcl_last_error()
