"""Seeded shape fuzz of the HIP path through the C ABI: odd channel counts, one-pixel-wide / non-square maps, every mask pattern, all
three contraction precisions, both warp sources and corner conventions.  The checker is the numpy oracle (correlation, context
matching) or stock ATen on the CPU (the warp: the oracle's align_set is align_corners = True only).  Bars as in test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import tcl_oracle as O
from tests.conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, grad=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(True) if grad else t


def N(t):
    return None if t is None else t.detach().cpu().numpy()


def _mask(rng, shape, kind):
    if kind == 0:
        return None
    if kind == 1:
        return (rng.random(shape) < 0.7).astype(np.float32)
    m = np.ones(shape, np.float32)      # one fully masked image: zero norms everywhere in it
    m[0] = 0.0
    return m


def _corr_cases():
    rng = np.random.default_rng(2024)
    cases = []
    for i in range(60):
        b, f = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        c = int(rng.choice([1, 3, 8, 31, 32, 33, 64, 100, 160]))
        h, w = [(5, 7), (8, 8), (16, 16), (12, 20), (33, 3), (1, 9), (24, 24), (9, 1)][int(rng.integers(0, 8))]
        cases.append((i, b, c, f, h, w, int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(0, 3))))
    return cases


@pytest.mark.parametrize("seed,b,c,f,h,w,kt,kr,prec", _corr_cases())
def test_fuzz_corr4d(seed, b, c, f, h, w, kt, kr, prec):
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(seed)
    xt = rng.standard_normal((b, c, h, w)).astype(np.float32)
    xr = rng.standard_normal((b, c, f, h, w)).astype(np.float32)
    vt, vr = _mask(rng, (b, 1, h, w), kt), _mask(rng, (b, 1, f, h, w), kr)
    cot = rng.standard_normal((b, f, h, w, h, w)).astype(np.float32)
    precision = (_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16)[prec]
    rtol = 1e-2 if precision == _lib.PREC_BF16 else 1e-4
    # absolute floor on unit-norm outputs: 1e-5 for fp32; the split-bf16 product drops lo * lo and rounds lo, ~3 * 2^-18 of |a^||r^| per
    # term, which only shows when C is 1-3 and single components are O(1)
    atol = 1e-2 if precision == _lib.PREC_BF16 else (2e-5 if precision == _lib.PREC_BF16X3 else 1e-5)
    txt, txr = T(xt, True), T(xr, True)
    out = _ops.corr4d(txt, T(vt), txr, T(vr), precision=precision)
    ref = O.compute_masked_4d_correlation(xt, vt, xr, vr)
    assert_close(N(out), ref, rtol=rtol, atol=atol, what="corr fwd")
    g_t, g_r = torch.autograd.grad((out * T(cot)).sum(), [txt, txr])
    if c == 1:
        return   # d(x / |x|) = 0: the gradient is the cancellation of two terms of size |d| / |x|, noise in any summation order
    rg_t, rg_r = O.compute_masked_4d_correlation_bwd(xt, vt, xr, vr, cot)
    assert_close(N(g_t), rg_t, rtol=rtol, atol=atol, what="corr g_x_t")
    assert_close(N(g_r), rg_r, rtol=rtol, atol=atol, what="corr g_x_r")


def _warp_cases():
    rng = np.random.default_rng(7)
    cases = []
    for i in range(60):
        cases.append((i, int(rng.integers(1, 3)), int(rng.choice([1, 2, 3, 3, 3, 4, 7])), int(rng.integers(1, 4)),
                      int(rng.choice([1, 2, 3, 5, 17, 33, 64, 70])), int(rng.choice([1, 2, 3, 6, 31, 64, 65, 130])),
                      bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), float(rng.choice([0.02, 0.3, 1.5]))))
    return cases


@pytest.mark.parametrize("seed,b,c,f,h,w,ac,use_theta,spread", _warp_cases())
def test_fuzz_warp(seed, b, c, f, h, w, ac, use_theta, spread):
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(100 + seed)
    x = rng.random((b, c, f, h, w)).astype(np.float32)
    cot = rng.standard_normal((b, c, f, h, w)).astype(np.float32)
    tx, cx = T(x, True), torch.tensor(x, requires_grad=True)
    if use_theta:
        th = (np.array([[1, 0, 0], [0, 1, 0]], np.float32) + rng.standard_normal((b * f, 2, 3)) * spread).astype(np.float32)
        ts, cs = T(th, True), torch.tensor(th, requires_grad=True)
        xa, _ = _ops.WarpFn.apply(tx, None, None, ts, ac, _lib.VMODE_NONE)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cgrid = Fn.affine_grid(cs, (b * f, c, h, w), align_corners=ac)
    else:
        g = (rng.random((b * f, h, w, 2)) * 2.6 - 1.3).astype(np.float32)
        ts, cs = T(g, True), torch.tensor(g, requires_grad=True)
        xa, _ = _ops.WarpFn.apply(tx, None, ts, None, ac, _lib.VMODE_NONE)
        cgrid = cs
    ref = Fn.grid_sample(cx.transpose(1, 2).reshape(b * f, c, h, w), cgrid, mode="bilinear", padding_mode="zeros", align_corners=ac)
    ref = ref.reshape(b, f, c, h, w).transpose(1, 2)
    gx, gs = torch.autograd.grad((xa * T(cot)).sum(), [tx, ts])
    rgx, rgs = torch.autograd.grad((ref * torch.tensor(cot)).sum(), [cx, cs])
    assert_close(N(xa), ref.detach().numpy(), rtol=2e-4, atol=2e-5, what="warp x")
    assert_close(N(gx), rgx.numpy(), rtol=2e-4, atol=2e-5, what="warp g_x")
    # theta gradients sum h*w products whose un-normalisation ATen's vectorised CPU kernel rounds differently (test_config_golden.py)
    assert_close(N(gs), rgs.numpy(), rtol=3e-3, atol=3e-3 if use_theta else 2e-4, what="warp g_source")


def _ctx_cases():
    rng = np.random.default_rng(11)
    cases = []
    for i in range(40):
        cases.append((i, int(rng.integers(1, 4)), int(rng.choice([1, 2, 5, 16, 33, 64])), int(rng.choice([1, 2, 3, 5, 9, 17])),
                      int(rng.choice([1, 3, 8, 15])), int(rng.choice([1, 4, 13, 27])), ("default", "complete", "cm_module")[int(rng.integers(0, 3))],
                      bool(rng.integers(0, 2))))
    return cases


@pytest.mark.parametrize("seed,b,c,r,h,w,variant,normalize", _ctx_cases())
def test_fuzz_context_matching(seed, b, c, r, h, w, variant, normalize):
    from master_thesis_amd.models import cpn, cpn_original
    rng = np.random.default_rng(500 + seed)
    up = lambda m, ax: np.repeat(np.repeat(m, 4, axis=ax), 4, axis=ax + 1)   # masks live at 4x the feature resolution
    cf = (0.3 * rng.standard_normal((b, c, r + 1, h, w))).astype(np.float32)
    vt = up((rng.random((b, 1, h, w)) < 0.6).astype(np.float32), 2)
    va = (rng.random((b, 1, r, h, w)) < 0.6).astype(np.float32)
    if seed % 3 == 0:
        va[:, :, 0] = 0.0                                        # a reference that never overlaps: the `zeros` branch
    va = up(va, 3)
    tcf = T(cf, True)
    imp = None
    if variant == "default":
        p_in, c_mask, imp = cpn.CPNContextMatchingDefault()(tcf, T(vt), T(va), normalize=normalize)
    elif variant == "complete":
        p_in, c_mask, imp = cpn.CPNContextMatchingComplete()(tcf, T(vt), T(va), normalize=normalize)
    else:
        p_in, c_mask = cpn_original.CM_Module()(tcf, T(vt), T(va))
    rp, rm, rimp = O.context_matching(cf, vt, va, normalize, variant)
    assert_close(N(p_in), rp, what="ctx p_in")
    assert_close(N(c_mask), rm, what="ctx c_mask")
    if imp is not None:
        assert_close(N(imp), rimp, what="ctx ref_importance")
    cot_p = rng.standard_normal(rp.shape).astype(np.float32)
    cot_m = rng.standard_normal(rm.shape).astype(np.float32)
    (g_cf,) = torch.autograd.grad((p_in * T(cot_p)).sum() + (c_mask * T(cot_m)).sum(), [tcf])
    with O.long_sums(np.float64):
        rg = O.context_matching_bwd(cf, vt, va, cot_p, cot_m, normalize, variant)
    assert_close(N(g_cf), rg, what="ctx g_c_feats")


def _small_cases():
    rng = np.random.default_rng(3)
    return [(i, int(rng.integers(1, 4)), int(rng.choice([2, 3, 17, 64, 129])), int(rng.choice([1, 2, 7, 16, 33])), int(rng.choice([1, 3, 8, 21])))
            for i in range(24)]


@pytest.mark.parametrize("seed,b,c,h,w", _small_cases())
def test_fuzz_corr1x1(seed, b, c, h, w):
    from master_thesis_amd import _ops
    rng = np.random.default_rng(900 + seed)
    f1, f2 = rng.standard_normal((b, c, h, w)).astype(np.float32), rng.standard_normal((b, c, h, w)).astype(np.float32)
    m1, m2 = (rng.random((b, 1, h, w)) < 0.7).astype(np.float32), (rng.random((b, 1, h, w)) < 0.7).astype(np.float32)
    cot = rng.standard_normal((b, 1, h, w)).astype(np.float32)
    t1, t2 = T(f1, True), T(f2, True)
    out = _ops.Corr1x1Fn.apply(t1, t2, T(m1), T(m2))
    assert_close(N(out), O.compute_masked_correlation(f1, f2, m1, m2), what="corr1x1 fwd")
    g1, g2 = torch.autograd.grad((out * T(cot)).sum(), [t1, t2])
    r1, r2 = O.compute_masked_correlation_bwd(f1, f2, m1, m2, cot)
    assert_close(N(g1), r1, what="corr1x1 g_1")
    assert_close(N(g2), r2, what="corr1x1 g_2")


@pytest.mark.parametrize("seed,b,r,h,w", _small_cases())
def test_fuzz_masked_softmax(seed, b, r, h, w):
    """(B,1,R,h,w) over the references, as context matching calls it, R on both sides of the 16-reference register path; rows with
    every reference masked out take the `sums < 1e-4` branch."""
    from master_thesis_amd import _ops
    rng = np.random.default_rng(1300 + seed)
    vec = (3 * rng.standard_normal((b, 1, r, h, w))).astype(np.float32)
    mask = (rng.random((b, 1, r, h, w)) < 0.5).astype(np.float32)
    cot = rng.standard_normal(vec.shape).astype(np.float32)
    tv = T(vec, True)
    out = _ops.MaskedSoftmaxFn.apply(tv, T(mask), 2)
    assert_close(N(out), O.masked_softmax(vec, mask, 2), what="masked_softmax fwd")
    (g,) = torch.autograd.grad((out * T(cot)).sum(), [tv])
    assert_close(N(g), O.masked_softmax_bwd(vec, mask, 2, cot), what="masked_softmax bwd")


def _argmax_cases():
    rng = np.random.default_rng(5)
    return [(i, int(rng.integers(1, 4)), int(rng.choice([1, 5, 32, 48, 100, 256])), int(rng.integers(1, 4)), int(rng.choice([2, 3, 7, 8, 12, 16, 31])),
             int(rng.integers(0, 3)), int(rng.integers(0, 3))) for i in range(40)]


@pytest.mark.parametrize("seed,b,c,f,s_,prec,kmask", _argmax_cases())
def test_fuzz_fused_argmax_equals_unfused(seed, b, c, f, s_, prec, kmask):
    """tcl_corr4d_argmax_flow == corr_to_flow(corr4d) at the same precision, bit for bit (first-maximum ties included: a fully masked
    image makes every similarity of its rows equal)."""
    from master_thesis_amd import _lib, _ops
    rng = np.random.default_rng(1700 + seed)
    xt = rng.standard_normal((b, c, s_, s_)).astype(np.float32)
    xr = rng.standard_normal((b, c, f, s_, s_)).astype(np.float32)
    vt, vr = _mask(rng, (b, 1, s_, s_), kmask), _mask(rng, (b, 1, f, s_, s_), (kmask + seed) % 3)
    precision = (_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16)[prec]
    fused = _ops.corr4d_argmax_flow(T(xt), T(vt), T(xr), T(vr), precision=precision)
    vol = _ops.corr4d(T(xt), T(vt), T(xr), T(vr), precision=precision)
    assert_exact(N(fused), N(_ops.corr_to_flow(vol)), what="fused argmax vs unfused")


def _infer_cases():
    rng = np.random.default_rng(9)
    return [(i, int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.choice([2, 3, 16, 33, 64, 90])), int(rng.choice([2, 5, 31, 64, 65, 130])))
            for i in range(24)]


@pytest.mark.parametrize("seed,b,f,h,w", _infer_cases())
def test_fuzz_inference_unit_equals_unfused(seed, b, f, h, w):
    """align_infer (one launch: hole masks flipped per tap, v_map in the epilogue) == align_set(x, 1 - m, flow) + vmap, bit for bit."""
    from master_thesis_amd import _ops
    from master_thesis_amd.utils import flow as tflow
    rng = np.random.default_rng(2100 + seed)
    x = rng.random((b, 3, f, h, w)).astype(np.float32)
    m = (rng.random((b, 1, f, h, w)) < 0.3).astype(np.float32)
    mt = (rng.random((b, 1, h, w)) < 0.3).astype(np.float32)
    fl = (rng.random((b, f, h, w, 2)) * 2.3 - 1.15).astype(np.float32)
    xa, va, vm = _ops.align_infer(T(x), T(m), T(fl), T(mt))
    xa0, va0 = tflow.align_set(T(x), 1 - T(m), T(fl))
    assert_exact(N(xa), N(xa0), what="x_aligned")
    assert_exact(N(va), N(va0), what="v_aligned")
    assert_exact(N(vm), N(_ops.vmap(va0, 1 - T(mt))), what="v_map")


def _mid_cases():
    rng = np.random.default_rng(13)
    sizes = [(24, 24), (32, 32), (40, 40), (48, 48), (56, 56), (64, 64), (30, 53), (20, 77)]
    cases = [(i, int(rng.integers(1, 3)), int(rng.choice([64, 96, 200, 256, 512])), int(rng.integers(1, 3)),
              *sizes[int(rng.integers(0, len(sizes)))], int(rng.integers(0, 3))) for i in range(18)]
    # shapes the pipelined bf16x3 backward takes (C % 256 == 0, P % 128 == 0), several reference frames = k segments of d_a
    cases += [(100, 1, 256, 2, 32, 32, 1), (101, 2, 512, 3, 16, 16, 1), (102, 1, 256, 1, 16, 24, 1), (103, 1, 256, 2, 16, 24, 2), (104, 1, 256, 3, 32, 32, 0)]
    return cases


@pytest.mark.parametrize("seed,b,c,f,h,w,prec", _mid_cases())
def test_fuzz_corr4d_mid_sizes(seed, b, c, f, h, w, prec):
    """Map sizes around the tile and wide-kernel eligibility edges (P = 576 ... 4096, not all multiples of 128 / 256), checked against
    the same formula in fp64 on the GPU (stock torch ops; the numpy oracle takes too long at these sizes)."""
    from master_thesis_amd import _lib, _ops
    g = torch.Generator(device="cpu").manual_seed(3000 + seed)
    xt = torch.randn(b, c, h, w, generator=g).to(DEV).requires_grad_(True)
    xr = torch.randn(b, c, f, h, w, generator=g).to(DEV).requires_grad_(True)
    vt = (torch.rand(b, 1, h, w, generator=g) < 0.8).float().to(DEV)
    vr = (torch.rand(b, 1, f, h, w, generator=g) < 0.8).float().to(DEV)
    cot = torch.randn(b, f, h, w, h, w, generator=g).to(DEV)
    precision = (_lib.PREC_F32, _lib.PREC_BF16X3, _lib.PREC_BF16)[prec]
    out = _ops.corr4d(xt, vt, xr, vr, precision=precision)
    g_t, g_r = torch.autograd.grad((out * cot).sum(), [xt, xr])
    xt64, xr64 = xt.detach().double().requires_grad_(True), xr.detach().double().requires_grad_(True)
    a = (xt64 * vt.double()).reshape(b, c, h * w)
    r = (xr64 * vr.double()).reshape(b, c, f, h * w)
    a = a / (a.norm(dim=1, keepdim=True) + 1e-9)
    r = r / (r.norm(dim=1, keepdim=True) + 1e-9)
    ref = torch.einsum("bcp,bcfq->bfpq", a, r).reshape(b, f, h, w, h, w)
    rg_t, rg_r = torch.autograd.grad((ref * cot.double()).sum(), [xt64, xr64])
    rtol, atol = (1e-2, 1e-2) if precision == _lib.PREC_BF16 else (1e-4, 1e-5)
    assert_close(N(out), N(ref), rtol=rtol, atol=atol, what="corr fwd")
    assert_close(N(g_t), N(rg_t), rtol=rtol, atol=atol, what="corr g_x_t")
    assert_close(N(g_r), N(rg_r), rtol=rtol, atol=atol, what="corr g_x_r")


def _flow_cases():
    rng = np.random.default_rng(17)
    out = []
    for i in range(30):
        b, f = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        h, w = int(rng.choice([2, 3, 8, 17, 64, 90])), int(rng.choice([2, 5, 16, 33, 64, 130]))
        ch, cw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
        out.append((i, b, f, h, w, ch, cw, int(rng.integers(0, h - ch + 1)), int(rng.integers(0, w - cw + 1)),
                    int(rng.choice([1, 2, 7, 16, 50, 128])), int(rng.choice([1, 3, 8, 31, 64, 200]))))
    return out


@pytest.mark.parametrize("seed,b,f,h,w,ch,cw,py,px,rh,rw", _flow_cases())
def test_fuzz_flow_utils(seed, b, f, h, w, ch, cw, py, px, rh, rw):
    """A11-A13: identity-grid add / subtract, crop (python-float scale factors) and legacy-nearest resize, values and indices exact."""
    from master_thesis_amd.utils import flow as tflow
    rng = np.random.default_rng(2500 + seed)
    fl = (rng.random((b, f, h, w, 2)) * 2.4 - 1.2).astype(np.float32)
    assert_exact(N(tflow.flow_abs_to_relative(T(fl))), O.flow_abs_to_relative(fl), what="abs -> relative")
    assert_exact(N(tflow.flow_relative_to_abs(T(fl))), O.flow_relative_to_abs(fl), what="relative -> abs")
    assert_exact(N(tflow.resize_flow(T(fl), (rh, rw))), O.resize_flow(fl, (rh, rw)), what="resize_flow")
    if h > 1 and w > 1 and ch > 0 and cw > 0:
        assert_close(N(tflow.crop_flow(T(fl), (ch, cw), (py, px))), O.crop_flow(fl, (ch, cw), (py, px)), rtol=1e-6, atol=1e-6, what="crop_flow")


@pytest.mark.parametrize("seed,b,f,s_", [(i, 1 + i % 3, 1 + i % 2, s_) for i, s_ in enumerate([2, 3, 5, 8, 13, 16, 24, 32])])
def test_fuzz_corr_to_flow_with_ties(seed, b, f, s_):
    """First-maximum argmax over volumes quantised to a few levels (ties everywhere), indices exact."""
    from master_thesis_amd.utils import flow as tflow
    rng = np.random.default_rng(2900 + seed)
    vol = (rng.integers(0, 4, (b, f, s_, s_, s_, s_)) / 4.0).astype(np.float32)
    assert_exact(N(tflow.corr_to_flow(T(vol))), O.corr_to_flow(vol), what="corr_to_flow")
