"""GPU parity tests: the HIP path (through the C ABI, via the drop-in modules) against
  (1) the golden vectors captured from the real reference (tests/golden/*.npz) and
  (2) the numpy oracle on seeded inputs at sizes it finishes in seconds,
with the bars of BASELINE.json: fp32 within 1e-4 relative, masks / indices bit-exact.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest
import torch

from oracle import tcl_oracle as O
from tests.conftest import assert_close, assert_exact, load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, grad=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(True) if grad else t


def N(t):
    return None if t is None else t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def mods():
    from master_thesis_amd.models import cpn, cpn_original
    from master_thesis_amd.utils import correlation, flow
    return dict(corr=correlation, flow=flow, cpn=cpn, cpo=cpn_original)


# ------------------------------------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("case", load_golden("corr4d"))
def test_golden_corr4d(mods, case):
    xt, xr = T(case["x_t"], True), T(case["x_r"], True)
    out = mods["corr"].compute_masked_4d_correlation(xt, T(case.get("v_t")), xr, T(case.get("v_r")))
    assert_close(N(out), case["out"], what="corr4d fwd")
    gt, gr = torch.autograd.grad((out * T(case["cot"])).sum(), [xt, xr])
    assert_close(N(gt), case["g_x_t"], what="corr4d g_x_t")
    assert_close(N(gr), case["g_x_r"], what="corr4d g_x_r")


@pytest.mark.parametrize("case", load_golden("corr1x1"))
def test_golden_corr1x1(mods, case):
    f1, f2 = T(case["f1"], True), T(case["f2"], True)
    out = mods["corr"].compute_masked_correlation(f1, f2, T(case["m1"]), T(case["m2"]))
    assert_close(N(out), case["out"], what="corr1x1 fwd")
    g1, g2 = torch.autograd.grad((out * T(case["cot"])).sum(), [f1, f2])
    assert_close(N(g1), case["g_f1"], what="corr1x1 g1")
    assert_close(N(g2), case["g_f2"], what="corr1x1 g2")


@pytest.mark.parametrize("case", load_golden("align_set"))
def test_golden_align_set(mods, case):
    x, flow = T(case["x"], True), T(case["flow"], True)
    xa, va = mods["flow"].align_set(x, T(case["v"]), flow)
    assert xa.shape == case["x_aligned"].shape and va.shape == case["v_aligned"].shape
    assert_close(N(xa), case["x_aligned"], what="align_set x")
    assert_exact(N(va), case["v_aligned"], what="align_set v")
    gx, gf = torch.autograd.grad((xa * T(case["cot"])).sum(), [x, flow])
    assert_close(N(gx), case["g_x"], what="align_set g_x")
    assert_close(N(gf), case["g_flow"], what="align_set g_flow")


@pytest.mark.parametrize("case", load_golden("cpn_align"))
def test_golden_cpn_align(mods, case):
    r_list = [int(i) for i in case["r_list"]]
    x, theta = T(case["x"], True), T(case["theta"], True)
    xa, va, ya = mods["cpn"].affine_align(x, T(case["m"]), T(case["y"]), theta, r_list, training=False)
    assert_close(N(xa), case["x_aligned"], what="cpn_align x")
    assert_close(N(ya), case["y_aligned"], what="cpn_align y")
    assert_exact(N(va), case["v_aligned"], what="cpn_align v")
    gx, gth = torch.autograd.grad((xa * T(case["cot"])).sum(), [x, theta])
    assert_close(N(gx), case["g_x"], what="cpn_align g_x")
    assert_close(N(gth), case["g_theta"], atol=2e-5, what="cpn_align g_theta")


@pytest.mark.parametrize("case", load_golden("masked_softmax"))
def test_golden_masked_softmax(mods, case):
    vec = T(case["vec"], True)
    out = mods["cpn"].masked_softmax(vec, T(case["mask"]), 2)
    assert_close(N(out), case["out"], what="masked_softmax")
    (g,) = torch.autograd.grad((out * T(case["cot"])).sum(), [vec])
    assert_close(N(g), case["g_vec"], what="masked_softmax bwd")
    out2 = mods["cpo"].CM_Module().masked_softmax(vec, T(case["mask"]), 2)
    assert_exact(N(out2), N(out))


@pytest.mark.parametrize("case", load_golden("mask_resize"))
def test_golden_mask_resize(case):
    from master_thesis_amd import _ops
    out = _ops.resize_mask(T(case["v"].astype(np.float32)), tuple(int(s) for s in case["size"]))
    assert_exact(N(out).astype(np.uint8), case["out"], what="mask resize")


@pytest.mark.parametrize("case", load_golden("context_matching"))
def test_golden_context_matching(mods, case):
    variant, normalize = str(case["variant"]), bool(case["normalize"])
    cf = T(case["c_feats"], True)
    v_t, v_al = T(case["v_t"]), T(case["v_aligned"])
    if variant == "default":
        p_in, c_mask, ref_imp = mods["cpn"].CPNContextMatchingDefault()(cf, v_t, v_al, normalize=normalize)
    elif variant == "complete":
        p_in, c_mask, ref_imp = mods["cpn"].CPNContextMatchingComplete()(cf, v_t, v_al, normalize=normalize)
    else:
        p_in, c_mask = mods["cpo"].CM_Module()(cf, v_t, v_al)
        ref_imp = None
    assert_close(N(p_in), case["p_in"], what="%s p_in" % variant)
    assert_close(N(c_mask), case["c_mask"], what="%s c_mask" % variant)
    if ref_imp is not None:
        assert_close(N(ref_imp), case["ref_importance"], what="%s ref_importance" % variant)
    (g,) = torch.autograd.grad((p_in * T(case["cot_p_in"])).sum() + (c_mask * T(case["cot_c_mask"])).sum(), [cf])
    assert_close(N(g), case["g_c_feats"], what="%s g_c_feats" % variant)


@pytest.mark.parametrize("case", load_golden("flow_utils"))
def test_golden_flow_utils(mods, case):
    op, fl = str(case["op"]), mods["flow"]
    if op == "corr_to_flow":
        assert_exact(N(fl.corr_to_flow(T(case["corr"]))), case["out"], what=op)
    elif op == "flow_abs_to_relative":
        assert_close(N(fl.flow_abs_to_relative(T(case["flow"]))), case["out"], atol=1e-6, what=op)
    elif op == "flow_relative_to_abs":
        assert_close(N(fl.flow_relative_to_abs(T(case["flow"]))), case["out"], atol=1e-6, what=op)
    elif op == "crop_flow":
        flow = T(case["flow"])
        out = fl.crop_flow(flow, tuple(int(s) for s in case["crop_size"]), tuple(int(s) for s in case["crop_position"]))
        assert_close(N(out), case["out"], atol=1e-6, what=op)
    elif op == "resize_flow":
        assert_exact(N(fl.resize_flow(T(case["flow"]), tuple(int(s) for s in case["size"]))), case["out"], what=op)


def test_dilate_and_add_border(mods):
    rng = np.random.default_rng(5)
    mask = (rng.random((2, 1, 13, 17)) < 0.3).astype(np.float32)
    from master_thesis_amd import _ops
    assert_exact(N(_ops.dilate_cross3(T(mask))), O.dilate_cross3(mask), what="dilate")
    x = rng.random((2, 3, 4, 13, 17)).astype(np.float32)
    got = N(mods["cpn"].add_border(T(x), T(mask)))
    border = ((O.dilate_cross3(mask) - mask) > 0).astype(np.float32)[:, :, None]
    ref = x * (1 - border)
    ref[:, 0:1] += border
    assert_close(got, ref, what="add_border")


# ------------------------------------------------------------------------------------------------ oracle, seeded, larger
def _masks(rng, shape, p=0.9):
    return (rng.random(shape) < p).astype(np.float32)


@pytest.mark.parametrize("B,C,F,H,W,masked", [
    (2, 64, 2, 16, 16, True),      # aligned fast path
    (1, 512, 4, 16, 16, True),     # the alignment model's shape (C1 / C3 of BASELINE.json)
    (2, 37, 3, 9, 7, True),        # ragged everything (scalar staging path, partial tiles)
    (1, 256, 1, 64, 64, False),    # one pair of C2
    (1, 20, 2, 24, 20, True),      # P % 128 != 0 with vector loads
    # >= 128 whole 256 x 256 tiles in the backward products: the wide kernel (gemm_f32_wide_kernel), which the small cases
    # above never reach: two reference frames = two accumulated segments and 8 k-tiles per segment; 32 k-tiles and 4 column
    # tiles; two row tiles (C = 512)
    (128, 256, 2, 16, 16, True),
    (32, 256, 1, 32, 32, False),
    (16, 512, 1, 32, 32, True),
    (43, 256, 3, 16, 16, True),    # 129 tiles for d_r only (d_a: 43 tiles -> the 256 x 128 kernel), three segments, odd batch
])
def test_corr4d_vs_oracle(mods, B, C, F, H, W, masked):
    rng = np.random.default_rng(B * 1000 + C)
    xt = rng.standard_normal((B, C, H, W)).astype(np.float32)
    xr = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    vt = _masks(rng, (B, 1, H, W)) if masked else None
    vr = _masks(rng, (B, 1, F, H, W)) if masked else None
    cot = rng.standard_normal((B, F, H, W, H, W)).astype(np.float32)
    txt, txr = T(xt, True), T(xr, True)
    out = mods["corr"].compute_masked_4d_correlation(txt, T(vt), txr, T(vr))
    assert_close(N(out), O.compute_masked_4d_correlation(xt, vt, xr, vr), what="corr4d fwd")
    gt, gr = torch.autograd.grad((out * T(cot)).sum(), [txt, txr])
    ogt, ogr = O.compute_masked_4d_correlation_bwd(xt, vt, xr, vr, cot)
    assert_close(N(gt), ogt, what="corr4d g_x_t")
    assert_close(N(gr), ogr, what="corr4d g_x_r")


def test_corr4d_properties_full_size(mods):
    """C2 shape (64x64, C=256): self-correlation has a unit diagonal and is symmetric; masked pixels give zero rows."""
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(2, 256, 64, 64, generator=g).to(DEV)
    v = (torch.rand(2, 1, 64, 64, generator=g) < 0.9).float().to(DEV)
    out = mods["corr"].compute_masked_4d_correlation(x, v, x.unsqueeze(2), v.unsqueeze(2)).reshape(2, 4096, 4096)
    vis = v.reshape(2, 4096)
    diag = torch.diagonal(out, dim1=1, dim2=2)
    assert torch.allclose(diag[vis > 0], torch.ones_like(diag[vis > 0]), atol=1e-5)
    assert torch.equal(diag[vis == 0], torch.zeros_like(diag[vis == 0]))
    assert torch.allclose(out, out.transpose(1, 2), atol=1e-6)
    assert float(out.abs().max()) <= 1.0 + 1e-5
    assert torch.equal(out[0][vis[0] == 0], torch.zeros_like(out[0][vis[0] == 0]))


def test_corr4d_backward_is_deterministic_at_c2(mods):
    """No atomics and a fixed summation order in every contraction kernel: two runs at the C2 batch (the wide backward kernels,
    256 blocks on 256 CUs) give bit-identical volumes and gradients."""
    g = torch.Generator(device="cpu").manual_seed(12)
    B, C, F, S = 16, 256, 1, 64
    xt = torch.randn(B, C, S, S, generator=g).to(DEV).requires_grad_(True)
    xr = torch.randn(B, C, F, S, S, generator=g).to(DEV).requires_grad_(True)
    cot = torch.randn(B, F, S, S, S, S, generator=g).to(DEV)
    runs = []
    for _ in range(2):
        out = mods["corr"].compute_masked_4d_correlation(xt, None, xr, None)
        runs.append((out.detach().clone(),) + tuple(t.clone() for t in torch.autograd.grad(out, [xt, xr], grad_outputs=cot)))
    for a, b, what in zip(runs[0], runs[1], ("volume", "g_x_t", "g_x_r")):
        assert torch.equal(a, b), what


def test_corr4d_c2_full_batch_vs_torch_fp32(mods):
    """BASELINE configs[1] at its full size (16 pairs, C = 256, 64 x 64): the shape bench.py times, i.e. the 128 x 128 forward
    kernel on 16,384 tiles and the wide backward kernels on 256.  Too large for the numpy oracle in a test, so the reference
    here is the same formula (utils/correlation.py:28-54) in stock fp32 torch ops on the GPU (rocBLAS bmm + autograd):
    an independent implementation of a floating-point kernel, held to the fp32 bar."""
    g = torch.Generator(device="cpu").manual_seed(11)
    B, C, F, S = 16, 256, 1, 64
    xt = torch.randn(B, C, S, S, generator=g).to(DEV).requires_grad_(True)
    xr = torch.randn(B, C, F, S, S, generator=g).to(DEV).requires_grad_(True)
    vt = (torch.rand(B, 1, S, S, generator=g) < 0.9).float().to(DEV)
    vr = (torch.rand(B, 1, F, S, S, generator=g) < 0.9).float().to(DEV)
    cot = torch.randn(B, F, S * S, S * S, generator=g).to(DEV)

    out = mods["corr"].compute_masked_4d_correlation(xt, vt, xr, vr)
    gt, gr = torch.autograd.grad(out, [xt, xr], grad_outputs=cot.view_as(out))

    xt2, xr2 = xt.detach().clone().requires_grad_(True), xr.detach().clone().requires_grad_(True)
    a = (xt2 * vt).reshape(B, C, -1)
    a = a / (a.norm(dim=1, keepdim=True) + 1e-9)
    r = (xr2 * vr).reshape(B, C, F, -1)
    r = r / (r.norm(dim=1, keepdim=True) + 1e-9)
    ref = torch.einsum("bcp,bcfq->bfpq", a, r)
    rgt, rgr = torch.autograd.grad(ref, [xt2, xr2], grad_outputs=cot)

    def close(got, want, what):
        got, want = got.detach(), want.detach()
        err = (got - want).abs()
        tol = 1e-5 * float(want.abs().max()) + 1e-4 * want.abs()
        bad = int((err > tol).sum())
        assert bad == 0, "%s: %d / %d off, worst %.3e" % (what, bad, err.numel(), float((err - tol).max()))

    close(out.reshape(B, F, S * S, S * S), ref, "corr4d fwd")
    close(gt, rgt, "corr4d g_x_t")
    close(gr, rgr, "corr4d g_x_r")


@pytest.mark.parametrize("precision,margin", [("f32", 1e-5), ("bf16x3", 1e-4), ("bf16", 5e-3)])
@pytest.mark.parametrize("B,C,F,S", [(2, 32, 2, 8), (1, 64, 1, 16), (1, 256, 1, 64), (2, 19, 3, 5), (2, 256, 2, 32)])
def test_fused_argmax_flow(mods, B, C, F, S, precision, margin):
    """corr4d + corr_to_flow fused == corr_to_flow(corr4d(...)) at the same precision exactly (same arithmetic: the fused kernel
    shares the store kernel's k-loop), and == the oracle wherever the two best similarities are separated by more than that
    precision's noise (fp32: 1e-5; bf16x3: 1e-4; bf16 operands: 5e-3, ~20 sigma of the rounding error of a unit-vector dot)."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(S)
    prec = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16": _lib.PREC_BF16}[precision]
    xt = rng.standard_normal((B, C, S, S)).astype(np.float32)
    xr = rng.standard_normal((B, C, F, S, S)).astype(np.float32)
    vt, vr = _masks(rng, (B, 1, S, S)), _masks(rng, (B, 1, F, S, S))
    fused = N(_ops.corr4d_argmax_flow(T(xt), T(vt), T(xr), T(vr), precision=prec))
    vol = mods["corr"].compute_masked_4d_correlation(T(xt), T(vt), T(xr), T(vr), precision=precision)
    assert_exact(fused, N(mods["flow"].corr_to_flow(vol)), what="fused vs unfused (%s)" % precision)
    ovol = O.compute_masked_4d_correlation(xt, vt, xr, vr)
    oflow = O.corr_to_flow(ovol)
    top2 = np.sort(ovol.reshape(B, F, S, S, S * S), axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > margin
    assert clear.mean() > (0.8 if precision == "f32" else 0.2)
    assert_exact(fused[clear], oflow[clear], what="fused vs oracle (clear maxima, %s)" % precision)
    fused_none = N(_ops.corr4d_argmax_flow(T(xt), None, T(xr), None, precision=prec))
    vol_none = mods["corr"].compute_masked_4d_correlation(T(xt), None, T(xr), None, precision=precision)
    assert_exact(fused_none, N(mods["flow"].corr_to_flow(vol_none)), what="fused vs unfused, no masks (%s)" % precision)


@pytest.mark.parametrize("B,F,H,W,amp", [(4, 4, 64, 64, 0.3), (2, 4, 256, 256, 0.2), (3, 1, 33, 47, 1.0), (2, 2, 16, 16, 0.5)])
def test_align_set_vs_oracle(mods, B, F, H, W, amp):
    rng = np.random.default_rng(H + W)
    x = rng.random((B, 3, F, H, W)).astype(np.float32)
    v = _masks(rng, (B, 1, F, H, W), 0.8)
    ident = np.stack(np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H)), axis=-1).astype(np.float32)
    flow = (ident[None, None] + amp * (rng.random((B, F, H, W, 2)) * 2 - 1)).astype(np.float32)
    cot = rng.standard_normal((B, 3, F, H, W)).astype(np.float32)
    tx, tf = T(x, True), T(flow, True)
    xa, va = mods["flow"].align_set(tx, T(v), tf)
    oxa, ova = O.align_set(x, v, flow)
    assert_close(N(xa), oxa, what="align_set x")
    assert_exact(N(va), ova, what="align_set v")
    gx, gf = torch.autograd.grad((xa * T(cot)).sum(), [tx, tf])
    ogx, ogf = O.align_set_bwd(x, v, flow, cot)
    assert_close(N(gx), ogx, what="align_set g_x")
    assert_close(N(gf), ogf, what="align_set g_flow")
    # T1: v_map
    from master_thesis_amd import _ops
    vt = _masks(rng, (B, 1, H, W), 0.8)
    assert_exact(N(_ops.vmap(va, T(vt))), O.v_map(ova, vt), what="v_map")


def test_align_set_strided_inputs(mods):
    """The runners pass views (x[:, :, r_list], 1 - m[...]): non-contiguous frames must give the same result."""
    rng = np.random.default_rng(11)
    x = rng.random((2, 3, 5, 20, 24)).astype(np.float32)
    v = _masks(rng, (2, 1, 5, 20, 24), 0.8)
    flow = (rng.random((2, 4, 20, 24, 2)) * 2.4 - 1.2).astype(np.float32)
    r_list = [0, 1, 3, 4]
    tx, tv = T(x), T(v)
    xa, va = mods["flow"].align_set(tx[:, :, r_list], tv[:, :, r_list], T(flow))
    oxa, ova = O.align_set(x[:, :, r_list], v[:, :, r_list], flow)
    assert_close(N(xa), oxa)
    assert_exact(N(va), ova)
    xa2, va2 = mods["flow"].align_set(tx[:, :, ::2], tv[:, :, ::2], T(flow[:, :3]))   # strided slice: planes stay dense
    oxa2, ova2 = O.align_set(x[:, :, ::2], v[:, :, ::2], flow[:, :3])
    assert_close(N(xa2), oxa2)
    assert_exact(N(va2), ova2)


def test_cpn_align_c4_shape(mods):
    """C4: 240x424 frames, 10 references, theta = identity + noise."""
    rng = np.random.default_rng(4)
    B, R, H, W = 2, 10, 240, 424
    x = rng.random((B, 3, R + 1, H, W)).astype(np.float32)
    m = 1 - np.repeat(np.repeat(_masks(rng, (B, 1, R + 1, H // 8, W // 8), 0.7), 8, axis=3), 8, axis=4)
    r_list = [i for i in range(R + 1) if i != R // 2]
    theta = (np.array([[1, 0, 0], [0, 1, 0]], np.float32)[None] + 0.05 * rng.standard_normal((B * R, 2, 3))).astype(np.float32)
    tx, tth = T(x, True), T(theta, True)
    xa, va, ya = mods["cpn"].affine_align(tx, T(m), None, tth, r_list, training=True)
    oxa, ova, _ = O.cpn_align(x, m, None, theta, r_list)
    assert ya is None
    assert_close(N(xa), oxa, what="cpn_align x")
    assert_exact(N(va), ova, what="cpn_align v (bilinear > 0.5)")
    cot = rng.standard_normal(oxa.shape).astype(np.float32)
    gx, gth = torch.autograd.grad((xa * T(cot)).sum(), [tx, tth])
    ogx, ogth = O.cpn_align_bwd(x, theta, r_list, cot)
    assert_close(N(gx), ogx, what="cpn_align g_x")
    # the oracle sums the 101,760 per-pixel terms in float64 (affine_grid_bwd); positions are bit-identical on both sides
    assert_close(N(gth), ogth, rtol=1e-4, atol=1e-5, what="cpn_align g_theta (sum over 101,760 pixels)")


@pytest.mark.parametrize("variant", ["default", "complete", "cm_module"])
@pytest.mark.parametrize("normalize", [True, False])
def test_context_matching_c4_shape(mods, variant, normalize):
    """C4: c_feats (B,128,11,30,53), masks 240x424."""
    if variant == "cm_module" and not normalize:
        pytest.skip("CM_Module always normalises")
    rng = np.random.default_rng(40)
    B, C, R, h, w, H, W = 2, 128, 10, 30, 53, 240, 424
    scale = 0.05 if not normalize else 1.0     # keep exp() of un-normalised similarities in a testable range
    cf = (scale * rng.standard_normal((B, C, R + 1, h, w))).astype(np.float32)
    v_t = np.repeat(np.repeat(_masks(rng, (B, 1, H // 4, W // 4), 0.6), 4, axis=2), 4, axis=3)
    v_al = np.repeat(np.repeat(_masks(rng, (B, 1, R, H // 4, W // 4), 0.7), 4, axis=3), 4, axis=4)
    v_al[0, :, 3] = 0
    tcf = T(cf, True)
    if variant == "default":
        p_in, c_mask, ref_imp = mods["cpn"].CPNContextMatchingDefault()(tcf, T(v_t), T(v_al), normalize=normalize)
    elif variant == "complete":
        p_in, c_mask, ref_imp = mods["cpn"].CPNContextMatchingComplete()(tcf, T(v_t), T(v_al), normalize=normalize)
    else:
        p_in, c_mask = mods["cpo"].CM_Module()(tcf, T(v_t), T(v_al))
        ref_imp = None
    with O.long_sums(np.float64):      # the 203,520-term similarity sums exactly: the yardstick is not another fp32 order
        op, om, oimp = O.context_matching(cf, v_t, v_al, normalize, variant)
    assert_close(N(p_in), op, what="p_in")
    assert_close(N(c_mask), om, what="c_mask")
    if ref_imp is not None:
        assert_close(N(ref_imp), oimp, what="ref_importance")
    cot1 = rng.standard_normal(op.shape).astype(np.float32)
    cot2 = rng.standard_normal(om.shape).astype(np.float32)
    (g,) = torch.autograd.grad((p_in * T(cot1)).sum() + (c_mask * T(cot2)).sum(), [tcf])
    with O.long_sums(np.float64):
        og = O.context_matching_bwd(cf, v_t, v_al, cot1, cot2, normalize, variant)
    assert_close(N(g), og, what="g_c_feats (203,520-term sums)")


def test_errors_are_loud(mods):
    from master_thesis_amd import TclError
    with pytest.raises(TclError):
        mods["corr"].compute_masked_4d_correlation(torch.zeros(1, 2, 2, 2), None, torch.zeros(1, 2, 1, 2, 2), None)
    with pytest.raises(ValueError):
        mods["corr"].compute_masked_4d_correlation(torch.zeros(1, 2, 2, 2, device=DEV), None, torch.zeros(1, 2, 2, device=DEV), None)
    with pytest.raises(TclError):
        mods["flow"].crop_flow(torch.zeros(1, 1, 4, 4, 2, device=DEV), (3, 3), (2, 2))


# ------------------------------------------------------------------------------------------------ precision variants of A1
def assert_close_bf16(got, ref, what=""):
    """north_star's bar for the bf16-MFMA variant: 1e-2 (relative to the tensor's scale, plus 1e-2 relative)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = max(1e-6, float(np.max(np.abs(ref))))
    bad = np.abs(got - ref) > 1e-2 * scale + 1e-2 * np.abs(ref)
    assert not bad.any(), "%s: %d/%d outside the bf16 bar" % (what, bad.sum(), bad.size)


@pytest.mark.parametrize("B,C,F,H,W,masked", [
    (2, 64, 2, 16, 16, True), (1, 512, 4, 16, 16, True), (2, 37, 3, 9, 7, True), (1, 256, 1, 64, 64, False), (1, 20, 2, 24, 20, True)])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_corr4d_precision_variants(mods, B, C, F, H, W, masked, precision):
    """TCL_PREC_BF16X3 (hi/lo split, three bf16 MFMAs) must meet the fp32 bar; TCL_PREC_BF16 the 1e-2 bar."""
    rng = np.random.default_rng(B * 1000 + C)
    xt = rng.standard_normal((B, C, H, W)).astype(np.float32)
    xr = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    vt = _masks(rng, (B, 1, H, W)) if masked else None
    vr = _masks(rng, (B, 1, F, H, W)) if masked else None
    cot = rng.standard_normal((B, F, H, W, H, W)).astype(np.float32)
    txt, txr = T(xt, True), T(xr, True)
    out = mods["corr"].compute_masked_4d_correlation(txt, T(vt), txr, T(vr), precision=precision)
    gt, gr = torch.autograd.grad((out * T(cot)).sum(), [txt, txr])
    oout = O.compute_masked_4d_correlation(xt, vt, xr, vr)
    ogt, ogr = O.compute_masked_4d_correlation_bwd(xt, vt, xr, vr, cot)
    check = assert_close if precision == "bf16x3" else assert_close_bf16
    check(N(out), oout, what="corr4d fwd " + precision)
    check(N(gt), ogt, what="corr4d g_x_t " + precision)
    check(N(gr), ogr, what="corr4d g_x_r " + precision)


@pytest.mark.parametrize("case", load_golden("corr4d"))
def test_golden_corr4d_bf16x3(mods, case):
    xt, xr = T(case["x_t"], True), T(case["x_r"], True)
    out = mods["corr"].compute_masked_4d_correlation(xt, T(case.get("v_t")), xr, T(case.get("v_r")), precision="bf16x3")
    assert_close(N(out), case["out"], what="corr4d fwd bf16x3")
    gt, gr = torch.autograd.grad((out * T(case["cot"])).sum(), [xt, xr])
    assert_close(N(gt), case["g_x_t"], what="corr4d g_x_t bf16x3")
    assert_close(N(gr), case["g_x_r"], what="corr4d g_x_r bf16x3")


# ------------------------------------------------------------------------------------------------ shape corners of the rewritten kernels
@pytest.mark.parametrize("C,F,H,W", [(1, 2, 12, 20), (5, 1, 9, 31), (4, 3, 2, 2), (3, 2, 7, 1), (2, 1, 1, 5)])
def test_warp_generic_channel_counts(C, F, H, W):
    """align_set hard-codes 3 channels in the reference; the kernel's generic-C path (and W < 2) must agree with the same arithmetic."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(C * 100 + W)
    B = 2
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    v = _masks(rng, (B, 1, F, H, W), 0.7)
    flow = (rng.random((B, F, H, W, 2)) * 2.6 - 1.3).astype(np.float32)
    cot = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    tx, tf = T(x, True), T(flow, True)
    xa, va = _ops.WarpFn.apply(tx, T(v), tf.reshape(-1, H, W, 2), None, True, _lib.VMODE_NEAREST)
    oxa, ova = O.align_set(x, v, flow)
    assert_close(N(xa), oxa, what="warp x")
    assert_exact(N(va), ova, what="warp v")
    gx, gf = torch.autograd.grad((xa * T(cot)).sum(), [tx, tf])
    ogx, ogf = O.align_set_bwd(x, v, flow, cot)
    assert_close(N(gx), ogx, what="warp g_x")
    assert_close(N(gf), ogf, what="warp g_flow")


@pytest.mark.parametrize("H,W", [(7, 1), (1, 5), (1, 1), (6, 9)])
@pytest.mark.parametrize("align_corners", [False, True])
def test_warp_thin_images_both_corner_conventions(H, W, align_corners):
    """One-pixel-wide / -high images: with align_corners = False the sampled column is g / 2, so every negative grid value puts the WEST tap
    outside and the east tap on column 0 (the kernel's clamped two-wide load must then hand out its first element).  Checked against
    stock ATen on the CPU, forward and both gradients."""
    from master_thesis_amd import _lib, _ops
    import torch.nn.functional as Fn
    rng = np.random.default_rng(H * 10 + W)
    B, C, F = 2, 3, 2
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    grid = (rng.random((B * F, H, W, 2)) * 2.4 - 1.2).astype(np.float32)
    cot = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    tx, tg = T(x, True), T(grid, True)
    xa, _ = _ops.WarpFn.apply(tx, None, tg, None, align_corners, _lib.VMODE_NONE)
    gx, gg = torch.autograd.grad((xa * T(cot)).sum(), [tx, tg])
    cx = torch.tensor(x, requires_grad=True)
    cg = torch.tensor(grid, requires_grad=True)
    ref = Fn.grid_sample(cx.transpose(1, 2).reshape(B * F, C, H, W), cg, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    ref = ref.reshape(B, F, C, H, W).transpose(1, 2)
    rgx, rgg = torch.autograd.grad((ref * torch.tensor(cot)).sum(), [cx, cg])
    assert_close(N(xa), ref.detach().numpy(), what="thin warp x")
    assert_close(N(gx), rgx.numpy(), what="thin warp g_x")
    assert_close(N(gg), rgg.numpy(), what="thin warp g_grid")


@pytest.mark.parametrize("C,H,W,scale,rot,jitter", [
    (3, 64, 128, 1.0, 0.0, 0.0),       # exact identity: every north / south and east / west pair merges
    (3, 96, 200, 1.08, 0.15, 0.001),   # upscaling + rotation: columns skip, rows drift
    (3, 70, 90, 0.8, -0.2, 0.002),     # downscaling: two output pixels share a source column / row
    (5, 40, 70, 1.03, 0.05, 0.0),      # generic channel count: two channel groups, grid gradient accumulated over them
    (1, 33, 65, 0.95, 0.3, 0.01),      # ragged tiles (one row / one column beyond the tile)
    (3, 64, 64, 1.0, 0.0, 1.5),        # wild flow: nothing merges
])
def test_warp_tiled_backward_smooth_flows(C, H, W, scale, rot, jitter):
    """The tiled backward (images of at least 32 x 64) sums the contributions of neighbouring output pixels in registers before
    the float atomics; smooth flows are where those merges fire.  Grid and theta sources, input + grid / theta gradients."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(C * 1000 + H + W)
    B, F = 2, 2
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    cot = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    xs, ys = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    flow = np.empty((B * F, H, W, 2), np.float32)
    theta = np.empty((B * F, 2, 3), np.float32)
    for i in range(B * F):
        a, s_ = rot * (1 + 0.1 * i), scale * (1 + 0.01 * i)
        tx, ty = 0.05 * i, -0.03 * i
        theta[i] = [[s_ * np.cos(a), -s_ * np.sin(a), tx], [s_ * np.sin(a), s_ * np.cos(a), ty]]
        flow[i] = np.stack([theta[i, 0, 0] * xs + theta[i, 0, 1] * ys + tx, theta[i, 1, 0] * xs + theta[i, 1, 1] * ys + ty], -1)
    flow = (flow + jitter * rng.standard_normal(flow.shape)).astype(np.float32)
    # grid source (align_corners = True, as align_set)
    tx_, tf = T(x, True), T(flow, True)
    xa, _ = _ops.WarpFn.apply(tx_, None, tf, None, True, _lib.VMODE_NONE)
    gx, gf = torch.autograd.grad((xa * T(cot)).sum(), [tx_, tf])
    src = x.transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W)
    ogx, ogf = O.grid_sample_bilinear_bwd(src, flow, True, cot.transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W))
    assert_close(N(gx).transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W), ogx, what="g_x (grid)")
    assert_close(N(gf), ogf, what="g_grid")
    # theta source (align_corners = False, as CPNet.align)
    tx_, tth = T(x, True), T(theta, True)
    xa, _ = _ops.WarpFn.apply(tx_, None, None, tth, False, _lib.VMODE_NONE)
    gx, gth = torch.autograd.grad((xa * T(cot)).sum(), [tx_, tth])
    grid = O.affine_grid(theta, H, W, False)
    ogx, ogg = O.grid_sample_bilinear_bwd(src, grid, False, cot.transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W))
    assert_close(N(gx).transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W), ogx, what="g_x (theta)")
    assert_close(N(gth), O.affine_grid_bwd(ogg, H, W, False), what="g_theta")


def test_warp_tiled_backward_wraparound_taps():
    """neighbouring lanes whose taps share a flat offset but not a (row, column): lane l samples (row r, x in [-1, 0))
    (only its east column is inside the image) and lane l + 1 samples (row r - 1, x in [W - 1, W)) (only its west column is).
    The duplicate-tap pre-merge of the tiled backward must not treat them as the same tap."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(77)
    B, C, F, H, W = 1, 3, 2, 32, 64
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    cot = rng.standard_normal((B, C, F, H, W)).astype(np.float32)
    ix = np.empty((F, H, W), np.float32)
    iy = np.empty((F, H, W), np.float32)
    rows = (np.arange(H) % (H - 2) + 1).astype(np.float32)[None, :, None]        # r in [1, H - 2]
    even = (np.arange(W) % 2 == 0)[None, None, :]
    ix[:] = np.where(even, -0.4, W - 1 + 0.3)
    iy[:] = np.where(even, rows + 0.3, rows - 1 + 0.3)
    ix[1] = np.where(even[0], W - 1 + 0.6, -0.7)                                  # frame 1: the pair the other way round
    iy[1] = np.where(even[0], rows[0] - 1 + 0.5, rows[0] + 0.5)
    flow = np.stack([2 * ix / (W - 1) - 1, 2 * iy / (H - 1) - 1], -1).astype(np.float32)
    tx_, tf = T(x, True), T(flow, True)
    xa, _ = _ops.WarpFn.apply(tx_, None, tf, None, True, _lib.VMODE_NONE)
    gx, gf = torch.autograd.grad((xa * T(cot)).sum(), [tx_, tf])
    src = x.transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W)
    ogx, ogf = O.grid_sample_bilinear_bwd(src, flow, True, cot.transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W))
    assert float(np.abs(ogx[:, :, :, 0]).max()) > 0 and float(np.abs(ogx[:, :, :, W - 1]).max()) > 0
    assert_close(N(gx).transpose(0, 2, 1, 3, 4).reshape(B * F, C, H, W), ogx, what="g_x (wrap-around taps)")
    assert_close(N(gf), ogf, what="g_grid (wrap-around taps)")


def test_warp_many_large_frames_one_launch():
    """200 frames of 300 x 300 = 352 chunks x 200 frames > 65535, each frame with its own source (gf == 1): the
    forward used to reject what the backward accepts."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(200)
    B, C, F, H, W = 1, 1, 200, 300, 300
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    v = _masks(rng, (B, 1, F, H, W), 0.7)
    flow = (rng.random((B, F, H, W, 2)) * 2.2 - 1.1).astype(np.float32)
    xa, va = _ops.WarpFn.apply(T(x), T(v), T(flow).reshape(-1, H, W, 2), None, True, _lib.VMODE_NEAREST)
    oxa = O.grid_sample_bilinear(x.transpose(0, 2, 1, 3, 4).reshape(F, C, H, W), flow.reshape(F, H, W, 2), True)
    ova = O.grid_sample_nearest(v.transpose(0, 2, 1, 3, 4).reshape(F, 1, H, W), flow.reshape(F, H, W, 2), True)
    assert_close(N(xa).transpose(0, 2, 1, 3, 4).reshape(F, C, H, W), oxa, what="warp x")
    assert_exact(N(va).transpose(0, 2, 1, 3, 4).reshape(F, 1, H, W), ova, what="warp v")


@pytest.mark.parametrize("B,C,R,h,w,variant", [
    (2, 6, 20, 5, 7, "default"),      # R > 16: two passes of the slab kernels; odd hw (one pixel per lane)
    (1, 7, 3, 6, 8, "complete"),      # odd channel count (a wave's second channel is missing)
    (2, 5, 33, 4, 6, "cm_module"),    # three passes
    (1, 1, 1, 3, 3, "default"),       # the smallest of everything
    (1, 130, 9, 9, 14, "default"),    # more channels than one d_match block covers
    (1, 8, 4, 40, 53, "default"),     # few waves, large map: the slab kernels split the pixel axis into 4 ranges (ragged last one)
    (1, 6, 18, 33, 41, "complete"),   # pixel ranges + two reference passes + one pixel per lane
    (1, 2, 2, 65, 128, "cm_module"),  # 16 pixel ranges of 5 steps over 65 steps: the last two ranges are empty
])
def test_context_matching_shape_corners(mods, B, C, R, h, w, variant):
    rng = np.random.default_rng(R * 10 + C)
    H, W = 4 * h, 4 * w
    cf = (0.3 * rng.standard_normal((B, C, R + 1, h, w))).astype(np.float32)
    v_t = np.repeat(np.repeat(_masks(rng, (B, 1, h, w), 0.6), 4, axis=2), 4, axis=3)
    v_al = np.repeat(np.repeat(_masks(rng, (B, 1, R, h, w), 0.7), 4, axis=3), 4, axis=4)
    tcf = T(cf, True)
    if variant == "default":
        p_in, c_mask, _ = mods["cpn"].CPNContextMatchingDefault()(tcf, T(v_t), T(v_al), normalize=True)
    elif variant == "complete":
        p_in, c_mask, _ = mods["cpn"].CPNContextMatchingComplete()(tcf, T(v_t), T(v_al), normalize=True)
    else:
        p_in, c_mask = mods["cpo"].CM_Module()(tcf, T(v_t), T(v_al))
    op, om, _ = O.context_matching(cf, v_t, v_al, True, variant)
    assert_close(N(p_in), op, what="p_in")
    assert_close(N(c_mask), om, what="c_mask")
    cot1 = rng.standard_normal(op.shape).astype(np.float32)
    cot2 = rng.standard_normal(om.shape).astype(np.float32)
    (g,) = torch.autograd.grad((p_in * T(cot1)).sum() + (c_mask * T(cot2)).sum(), [tcf])
    with O.long_sums(np.float64):
        og = O.context_matching_bwd(cf, v_t, v_al, cot1, cot2, True, variant)
    assert_close(N(g), og, what="g_c_feats")


@pytest.mark.parametrize("shape,dim", [((2, 3, 20, 5, 7), 2), ((3, 16, 33), 1), ((2, 1, 4, 9), 0), ((1, 5, 17, 3, 3), 2)])
def test_masked_softmax_vs_oracle_all_paths(mods, shape, dim):
    """R > 16 takes the generic kernel, R <= 16 the register-resident one; dim / broadcast layouts other than the reference's."""
    rng = np.random.default_rng(sum(shape) + dim)
    vec = (3 * rng.standard_normal(shape)).astype(np.float32)
    mshape = list(shape)
    if len(shape) == 5:
        mshape[1] = 1
    mask = _masks(rng, tuple(mshape), 0.6)
    tv = T(vec, True)
    out = mods["cpn"].masked_softmax(tv, T(mask), dim)
    assert_close(N(out), O.masked_softmax(vec, mask, dim), what="masked_softmax")
    cot = rng.standard_normal(shape).astype(np.float32)
    (g,) = torch.autograd.grad((out * T(cot)).sum(), [tv])
    assert_close(N(g), O.masked_softmax_bwd(vec, mask, dim, cot), what="masked_softmax bwd")


@pytest.mark.parametrize("B,C,H,W", [(2, 130, 9, 13), (1, 2, 5, 5), (3, 37, 8, 8), (1, 64, 1, 100)])
def test_corr1x1_vs_oracle_ragged(mods, B, C, H, W):
    """Channel counts that do not divide over the four waves / the unroll, pixel counts that are not multiples of 64."""
    rng = np.random.default_rng(B * 1000 + C)
    f1, f2 = rng.standard_normal((B, C, H, W)).astype(np.float32), rng.standard_normal((B, C, H, W)).astype(np.float32)
    m1, m2 = _masks(rng, (B, 1, H, W), 0.8), _masks(rng, (B, 1, H, W), 0.8)
    t1, t2 = T(f1, True), T(f2, True)
    out = mods["corr"].compute_masked_correlation(t1, t2, T(m1), T(m2))
    assert_close(N(out), O.compute_masked_correlation(f1, f2, m1, m2), what="corr1x1 fwd")
    cot = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    g1, g2 = torch.autograd.grad((out * T(cot)).sum(), [t1, t2])
    o1, o2 = O.compute_masked_correlation_bwd(f1, f2, m1, m2, cot)
    assert_close(N(g1), o1, what="corr1x1 g1")
    assert_close(N(g2), o2, what="corr1x1 g2")


@pytest.mark.parametrize("H,W,h,w", [(22, 27, 5, 7), (20, 36, 5, 9), (17, 9, 20, 13), (24, 40, 3, 5)])
@pytest.mark.parametrize("variant", ["default", "complete"])
def test_ref_importance_odd_and_vector_widths(mods, H, W, h, w, variant):
    """ref_importance at widths that take the one-pixel-per-lane kernel (W % 4 != 0) and the four-pixel one, at
    non-integer mask scales (incl. upscaling)."""
    rng = np.random.default_rng(H * 100 + W)
    B, C, R = 2, 4, 3
    cf = (0.3 * rng.standard_normal((B, C, R + 1, h, w))).astype(np.float32)
    v_t, v_al = _masks(rng, (B, 1, H, W), 0.6), _masks(rng, (B, 1, R, H, W), 0.7)
    cls = mods["cpn"].CPNContextMatchingDefault if variant == "default" else mods["cpn"].CPNContextMatchingComplete
    p_in, c_mask, ref_imp = cls()(T(cf), T(v_t), T(v_al), normalize=True)
    op, om, oref = O.context_matching(cf, v_t, v_al, True, variant)
    assert_close(N(p_in), op, what="p_in")
    assert_close(N(ref_imp), oref, what="ref_importance")


# ------------------------------------------------------------------------------------------------ callers of the path
def _standin_model(flows):
    """A callable with the alignment network's interface that returns fixed flows (the conv stacks are out of scope)."""
    def model(x_t, m_t, x_r, m_r):
        return (None,) + tuple(flows)
    return model


def test_runner_step_functions_vs_oracle():
    """train_step_propagate / infer_step_propagate of thesis_dfpn/runner.py:104-159 on the HIP ops == the same composition of
    the oracle's ops (resize_set is stock F.interpolate on both sides)."""
    import torch.nn.functional as F
    from master_thesis_amd import runners
    rng = np.random.default_rng(77)
    B, Fr, H, W = 2, 3, 256, 256
    t, r_list = 1, [0, 2]
    x = rng.random((B, 3, Fr, H, W)).astype(np.float32)
    y = rng.random((B, 3, Fr, H, W)).astype(np.float32)
    m = 1 - np.repeat(np.repeat(_masks(rng, (B, 1, Fr, H // 16, W // 16), 0.8), 16, axis=3), 16, axis=4)
    def ident(s):
        return np.stack(np.meshgrid(np.linspace(-1, 1, s), np.linspace(-1, 1, s)), axis=-1).astype(np.float32)
    flow_gt = (ident(256)[None, None] * 1.05 + 0.1 * (rng.random((B, Fr, 1, 1, 2)) - 0.5)).astype(np.float32)
    flows = [(ident(s)[None, None] * 0.97 + 0.08 * (rng.random((B, len(r_list), 1, 1, 2)) - 0.5)).astype(np.float32) for s in (16, 64, 256)]
    flows_use = np.array([True, False])
    tflows = [T(f, True) for f in flows]
    out = runners.train_step_propagate(_standin_model(tflows), T(x), T(m), T(y), T(flow_gt), torch.from_numpy(flows_use).to(DEV), t, r_list)
    corr, xs, vs, ys, xs_al, xs_al_gt, vs_al, vs_al_gt, fl, fl_gt, _, v_maps, v_maps_gt = out
    # oracle composition on the CPU
    cx, cm, cy = torch.from_numpy(x), torch.from_numpy(m), torch.from_numpy(y)
    sets = [runners.resize_set(cx, 1 - cm, cy, 16), runners.resize_set(cx, 1 - cm, cy, 64), (cx, 1 - cm, cy)]
    gt256 = flow_gt[:, r_list]
    o_gt = [O.resize_flow(gt256, (16, 16)), O.resize_flow(gt256, (64, 64)), gt256]
    for k in range(3):
        assert_close(N(xs[k]), sets[k][0].numpy(), what="resize_set x")          # stock F.interpolate on two devices
        assert_exact(N(vs[k]), sets[k][1].numpy(), what="resize_set v (nearest)")
        # the oracle warps the SAME resized frames the HIP path warped (F.interpolate's last bit differs between devices;
        # that is glue, not the path): every mask below is then bit-exact
        ox, ov = N(xs[k]), N(vs[k])
        assert_exact(N(fl_gt[k]), o_gt[k], what="flow_gt scale %d" % k)
        oxa, ova = O.align_set(ox[:, :, r_list], ov[:, :, r_list], flows[k])
        assert_close(N(xs_al[k]), oxa, what="x_aligned scale %d" % k)
        assert_exact(N(vs_al[k]), ova, what="v_aligned scale %d" % k)
        oxg, ovg = O.align_set(ox[:, :, r_list], ov[:, :, r_list], o_gt[k])
        assert_close(N(xs_al_gt[k]), oxg, what="x_aligned_gt scale %d" % k)
        assert_exact(N(vs_al_gt[k]), ovg, what="v_aligned_gt scale %d" % k)
        if k > 0:
            assert_exact(N(v_maps[k - 1]), O.v_map(ova, ov[:, :, t]), what="v_map scale %d" % k)
            assert_exact(N(v_maps_gt[k - 1]), O.v_map(ovg, ov[:, :, t]) * flows_use.reshape(-1, 1, 1, 1, 1), what="v_map_gt scale %d" % k)
    # gradients reach the predicted flows through the warps (what the reference's training exercises)
    loss = sum((xa * xa).sum() for xa in xs_al)
    grads = torch.autograd.grad(loss, tflows)
    for k in range(3):
        ox, ov = N(xs[k]), N(vs[k])
        oxa, _ = O.align_set(ox[:, :, r_list], ov[:, :, r_list], flows[k])
        _, ogf = O.align_set_bwd(ox[:, :, r_list], ov[:, :, r_list], flows[k], 2 * oxa)
        assert_close(N(grads[k]), ogf, what="g_flow scale %d" % k)
    # inference unit
    xa, va, vm = runners.infer_step_propagate(_standin_model(tflows), T(x[:, :, t]), T(m[:, :, t]), T(x[:, :, r_list]), T(m[:, :, r_list]))
    oxa, ova = O.align_set(x[:, :, r_list], 1 - m[:, :, r_list], flows[2])
    assert_close(N(xa), oxa, what="infer x_aligned")
    assert_exact(N(va), ova, what="infer v_aligned")
    assert_exact(N(vm), O.v_map(ova, 1 - m[:, :, t]), what="infer v_map")


def test_graft_entry_smoke_runs():
    """The driver's smoke() (one small invocation of every hot-path op against the oracle) must keep working."""
    import __graft_entry__ as entry
    entry.smoke()


# ------------------------------------------------------------------------------------------------ size-independent properties at full size
def test_align_set_properties_full_size(mods):
    """C3 size (32 sequences x 4 references, 256 x 256): the identity grid returns the frames (bilinear weights 1 / 0 up to the
    rounding of the pixel position) and the visibility maps bit-exactly; a grid shifted by exactly one pixel returns the shifted
    frame with a zero column entering (zeros padding); the warp is linear in x."""
    g = torch.Generator(device="cpu").manual_seed(17)
    B, F, S = 32, 4, 256
    x = torch.rand(B, 3, F, S, S, generator=g).to(DEV)
    v = (torch.rand(B, 1, F, S // 8, S // 8, generator=g) < 0.8).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(DEV)
    lin = torch.linspace(-1, 1, S, device=DEV)
    ident = torch.stack(torch.meshgrid(lin, lin, indexing="ij")[::-1], dim=-1).expand(B, F, S, S, 2).contiguous()
    xa, va = mods["flow"].align_set(x, v, ident)
    assert float((xa - x).abs().max()) <= 1e-4     # the pixel position ((g + 1) / 2) * 255 carries ~3e-5 of rounding
    assert torch.equal(va, v)
    shifted = ident.clone()
    shifted[..., 0] += 2.0 / (S - 1)            # sample one pixel to the right
    xs, vs = mods["flow"].align_set(x, v, shifted)
    assert float((xs[..., :-2] - x[..., 1:-1]).abs().max()) <= 1e-4
    assert float(xs[..., -1].abs().max()) <= 1e-4  # the position S falls outside: zeros (up to the rounding of the position)
    y = torch.rand(B, 3, F, S, S, generator=g).to(DEV)
    wob = ident + 0.01 * torch.randn(B, F, 1, 1, 2, generator=g).to(DEV)
    a, _ = mods["flow"].align_set(x, v, wob)
    b, _ = mods["flow"].align_set(y, v, wob)
    ab, _ = mods["flow"].align_set(2 * x - 3 * y, v, wob)
    assert float((ab - (2 * a - 3 * b)).abs().max()) <= 1e-5


def test_context_matching_properties_full_size(mods):
    """C4 size (32 targets, 128 x 11 x 30 x 53): two runs are bit-identical (fixed summation order, no float atomics); a target
    processed alone gives the batch's result (different pixel-range split -> tolerance, not bits); the aggregation weights of every
    pixel sum to one where a reference is visible, so c_mask is in [0, 1] and p_in's first C channels are the target's features."""
    g = torch.Generator(device="cpu").manual_seed(19)
    B, C, R, h, w, H, W = 32, 128, 10, 30, 53, 240, 424
    cf = (0.05 * torch.randn(B, C, R + 1, h, w, generator=g)).to(DEV)
    v_t = (torch.rand(B, 1, H // 8, W // 8, generator=g) < 0.6).float().repeat_interleave(8, 2).repeat_interleave(8, 3).to(DEV)
    v_al = (torch.rand(B, 1, R, H // 8, W // 8, generator=g) < 0.7).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(DEV)
    mod = mods["cpn"].CPNContextMatchingDefault()
    p1, m1, r1 = mod(cf, v_t, v_al, normalize=False)
    p2, m2, r2 = mod(cf, v_t, v_al, normalize=False)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(r1, r2)
    ps, ms, _ = mod(cf[5:6], v_t[5:6], v_al[5:6], normalize=False)
    assert float((ps - p1[5:6]).abs().max()) <= 1e-5 and float((ms - m1[5:6]).abs().max()) <= 1e-5
    assert torch.equal(p1[:, :C], cf[:, :, 0])
    assert float(m1.min()) >= -1e-6 and float(m1.max()) <= 1.0 + 1e-6
    assert torch.equal(p1[:, 2 * C:], m1)
    # a pixel visible in no reference keeps c_mask = 1 and an all-zero aggregated feature
    none = (torch.nn.functional.interpolate(v_al.reshape(B, R, H, W), size=(h, w), mode="bilinear", align_corners=False) > 0.5).sum(1) == 0
    assert bool(none.any())
    assert torch.equal(m1[:, 0][none], torch.ones_like(m1[:, 0][none]))
    assert float(p1[:, C:2 * C].permute(0, 2, 3, 1)[none].abs().max()) == 0.0


@pytest.mark.parametrize("frames", [[0, 1, 3, 4], [2], [4, 0, 1], [0, 2, 4, 1, 3], [1, 2, 3], [1, 1, 3]])
@pytest.mark.parametrize("use_theta", [True, False])
def test_warp_frames_equals_gathered_frames(mods, frames, use_theta):
    """WarpFn(x, ..., frames=r_list) (reference frames read in place, one launch per run of consecutive indices; more than
    four runs fall back to one gather) == WarpFn(x[:, :, r_list], ...) bit for bit: outputs, visibility, and the gradients
    to x (scatter-add into the full tensor), to theta and to the grid."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(len(frames) * 7 + use_theta)
    B, C, F, H, W = 3, 3, 5, 40, 72
    x = rng.random((B, C, F, H, W)).astype(np.float32)
    v = _masks(rng, (B, 1, F, H, W), 0.8)
    n = len(frames)
    theta = (np.tile(np.eye(2, 3, dtype=np.float32), (B * n, 1, 1)) + 0.1 * rng.standard_normal((B * n, 2, 3))).astype(np.float32)
    ident = np.stack(np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H)), axis=-1).astype(np.float32)
    grid = (ident[None] * 0.95 + 0.05 * rng.standard_normal((B * n, 1, 1, 2))).astype(np.float32)
    cot = rng.standard_normal((B, C, n, H, W)).astype(np.float32)

    def run(gather):
        tx = T(x, True)
        pos = T(theta if use_theta else grid, True)
        args = (None, pos) if use_theta else (pos, None)
        if gather:
            xo, vo = _ops.WarpFn.apply(tx[:, :, frames], T(v)[:, :, frames], args[0], args[1], False, _lib.VMODE_BILINEAR_THRESHOLD)
        else:
            xo, vo = _ops.WarpFn.apply(tx, T(v), args[0], args[1], False, _lib.VMODE_BILINEAR_THRESHOLD, frames)
        gx, gp = torch.autograd.grad((xo * T(cot)).sum(), [tx, pos])
        return N(xo), N(vo), N(gx), N(gp)

    a, b = run(True), run(False)
    assert_exact(b[0], a[0], what="x_out")
    assert_exact(b[1], a[1], what="v_out")
    assert_close(b[2], a[2], rtol=1e-6, atol=1e-6, what="g_x")        # float atomics: summation order
    assert_close(b[3], a[3], rtol=1e-5, atol=1e-6, what="g_theta / g_grid")


@pytest.mark.parametrize("B,F,H,W", [(1, 1, 240, 480), (2, 3, 33, 47), (3, 2, 64, 64)])
def test_align_infer_equals_the_unfused_unit(mods, B, F, H, W):
    """tcl_warp_fwd_fused (hole masks flipped per tap, v_map in the epilogue) == align_set(x, 1 - m, flow) followed by
    clamp(v_aligned - (1 - m_target), 0, 1) (thesis_dfpn/runner.py:157-158), bit for bit, and == the oracle; strided inputs too."""
    from master_thesis_amd import _ops
    rng = np.random.default_rng(B * 100 + W)
    x = rng.random((B, 3, F + 2, H, W)).astype(np.float32)
    m = (rng.random((B, 1, F + 2, H, W)) < 0.3).astype(np.float32)
    mt = (rng.random((B, 1, H, W)) < 0.3).astype(np.float32)
    flow = (rng.random((B, F, H, W, 2)) * 2.3 - 1.15).astype(np.float32)
    tx, tm = T(x)[:, :, 1:F + 1], T(m)[:, :, 1:F + 1]          # views into larger tensors (the schedulers slice frames)
    xa, va, vm = _ops.align_infer(tx, tm, T(flow), T(mt))
    xa0, va0 = mods["flow"].align_set(tx, 1 - tm, T(flow))
    vm0 = _ops.vmap(va0, 1 - T(mt))
    assert_exact(N(xa), N(xa0), what="x_aligned")
    assert_exact(N(va), N(va0), what="v_aligned")
    assert_exact(N(vm), N(vm0), what="v_map")
    oxa, ova = O.align_set(x[:, :, 1:F + 1], 1 - m[:, :, 1:F + 1], flow)
    assert_close(N(xa), oxa, what="x_aligned vs oracle")
    assert_exact(N(va), ova, what="v_aligned vs oracle")
    assert_exact(N(vm), np.clip(ova - (1 - mt)[:, :, None], 0, 1), what="v_map vs oracle")


def test_warp_hole_mask_flip_equals_one_minus_m():
    """WarpFn(..., v_is_hole=True) on m == WarpFn on 1 - m, both visibility modes, theta and grid sources."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(31)
    B, F, H, W = 2, 3, 40, 72
    x = rng.random((B, 3, F, H, W)).astype(np.float32)
    m = (rng.random((B, 1, F, H, W)) < 0.4).astype(np.float32)
    theta = (np.tile(np.eye(2, 3, dtype=np.float32), (B * F, 1, 1)) + 0.1 * rng.standard_normal((B * F, 2, 3))).astype(np.float32)
    flow = (rng.random((B * F, H, W, 2)) * 2.2 - 1.1).astype(np.float32)
    for grid, th, ac, mode in ((None, T(theta), False, _lib.VMODE_BILINEAR_THRESHOLD), (T(flow), None, True, _lib.VMODE_NEAREST),
                               (T(flow), None, True, _lib.VMODE_BILINEAR_THRESHOLD), (None, T(theta), False, _lib.VMODE_NEAREST)):
        xa, va = _ops.WarpFn.apply(T(x), T(m), grid, th, ac, mode, None, True)
        xb, vb = _ops.WarpFn.apply(T(x), 1 - T(m), grid, th, ac, mode)
        assert_exact(N(xa), N(xb), what="x")
        assert_exact(N(va), N(vb), what="v")


def test_argmax_flow_under_graph_capture_and_after():
    """The fused argmax (cached workspace outside capture, a private one inside) captured in a HIP graph, replayed on new inputs, and
    called eagerly again afterwards: all three agree with the eager result."""
    from master_thesis_amd import _ops
    g = torch.Generator().manual_seed(5)
    xt = torch.randn(1, 64, 16, 16, generator=g).to(DEV)
    xr = torch.randn(1, 64, 2, 16, 16, generator=g).to(DEV)
    eager = _ops.corr4d_argmax_flow(xt, None, xr, None).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _ops.corr4d_argmax_flow(xt, None, xr, None)          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = _ops.corr4d_argmax_flow(xt, None, xr, None)
    graph.replay()
    torch.cuda.synchronize()
    assert_exact(N(captured), N(eager), what="graph replay")
    xt2 = torch.randn(1, 64, 16, 16, generator=g).to(DEV)
    eager2 = _ops.corr4d_argmax_flow(xt2, None, xr, None).clone()
    xt.copy_(xt2)
    graph.replay()
    torch.cuda.synchronize()
    assert_exact(N(captured), N(eager2), what="graph replay on new inputs")
    assert_exact(N(_ops.corr4d_argmax_flow(xt2, None, xr, None)), N(eager2), what="eager after capture")
