#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ from the REAL reference.

Run ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's hot-path modules by file path (utils/correlation.py, utils/flow.py,
utils/transforms.py, models/cpn.py, models/cpn_original.py) with empty stub modules for the
packages the image lacks (cv2, skimage) and runs the reference functions on small seeded inputs with
torch (CPU, fp32).  Every fixture stores inputs, outputs and, where the op is differentiable,
the autograd gradients for a random cotangent.  The fixtures are DATA ONLY (.npz with
allow_pickle=False); no reference source text is stored.

cv2 is absent from the image: the only hot-path use is `cv2.dilate(img, MORPH_CROSS 3x3)` inside
`ImageTransforms.dilatate` (utils/transforms.py:87-105), reached from `add_match_border` /
`add_border` (models/cpn.py:23-44).  The stub below restates it with scipy.ndimage.grey_dilation
(3x3 cross footprint, constant -inf border = cv2's default morphology border), so the border mask of
`ref_importance` is pinned by that restatement and not by cv2 itself ("A9" in SURVEY.md §8c).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.ndimage
import torch

REF = os.environ.get("TCL_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- reference loader
def _stub_modules():
    cv2 = types.ModuleType("cv2")
    cv2.MORPH_CROSS = 1

    def get_structuring_element(shape, ksize):
        assert shape == cv2.MORPH_CROSS and tuple(ksize) == (3, 3)
        return np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=np.uint8)

    def dilate(img, kernel, iterations=1):
        out = np.asarray(img)
        squeeze = out.ndim == 3 and out.shape[2] == 1
        if squeeze:
            out = out[:, :, 0]
        for _ in range(iterations):
            out = scipy.ndimage.grey_dilation(out, footprint=kernel.astype(bool), mode="constant", cval=-np.inf)
        return out  # cv2 drops a trailing singleton channel

    cv2.getStructuringElement = get_structuring_element
    cv2.dilate = dilate
    sys.modules["cv2"] = cv2
    for name in ["skimage", "skimage.transform", "skimage.draw", "skimage.metrics"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.transform"].AffineTransform = None
    sys.modules["skimage.transform"].warp = None


def load_reference():
    _stub_modules()
    for pkg in ["utils", "models"]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]  # bare package: __init__.py is NOT executed
        sys.modules[pkg] = m
    mods = {}
    for name, rel in [
        ("utils.correlation", "utils/correlation.py"),
        ("utils.flow", "utils/flow.py"),
        ("utils.transforms", "utils/transforms.py"),
        ("models.cpn", "models/cpn.py"),
        ("models.cpn_original", "models/cpn_original.py"),
    ]:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        setattr(sys.modules[name.split(".")[0]], name.split(".")[1], m)
        mods[name] = m
    return mods


# ----------------------------------------------------------------------------- helpers
def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def randn(g, *shape):
    return torch.randn(*shape, generator=g)


def rand(g, *shape):
    return torch.rand(*shape, generator=g)


def bern(g, p, *shape):
    return (torch.rand(*shape, generator=g) < p).float()


def npy(t):
    if t is None:
        return None
    return t.detach().cpu().contiguous().numpy()


def grads(outs, cots, ins):
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    return torch.autograd.grad(loss, ins, allow_unused=True)


def save(name, cases):
    flat = {}
    for ci, case in enumerate(cases):
        for k, v in case.items():
            if v is None:
                continue
            flat["c%d/%s" % (ci, k)] = np.asarray(v)
    flat["n_cases"] = np.asarray(len(cases))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print("%-28s %3d cases %8.1f KiB" % (name, len(cases), os.path.getsize(path) / 1024))


def blocky_mask(g, b, f, H, W, cell, p):
    """(b,1,f,H,W) {0,1} visibility made of cell x cell blocks (like dilated rectangles)."""
    coarse = bern(g, p, b, 1, f, H // cell, W // cell)
    return coarse.repeat_interleave(cell, 3).repeat_interleave(cell, 4)


# ----------------------------------------------------------------------------- A1 corr4d
def make_corr4d(ref):
    fn = ref["utils.correlation"].compute_masked_4d_correlation
    cases = []
    specs = [
        # seed, B, C, F, H, W, mask_t, mask_r, relu
        (0, 2, 8, 3, 4, 4, True, True, False),
        (1, 1, 16, 2, 8, 8, False, False, False),
        (2, 2, 8, 1, 3, 5, True, False, True),
        (3, 1, 12, 2, 6, 4, True, True, True),
    ]
    for seed, B, C, F_, H, W, mt, mr, relu in specs:
        g = gen(100 + seed)
        xt, xr = randn(g, B, C, H, W), randn(g, B, C, F_, H, W)
        if relu:  # post-ReLU-like features with some all-zero pixels (norm == 0, unmasked)
            xt, xr = xt.clamp(min=0), xr.clamp(min=0)
            xt[:, :, 0, 1] = 0
            xr[:, :, 0, 1, 2] = 0
        vt = bern(g, 0.7, B, 1, H, W) if mt else None
        vr = bern(g, 0.7, B, 1, F_, H, W) if mr else None
        if vt is not None:
            vt[:, :, 1, 1] = 0
        if vr is not None:
            vr[:, :, :, 2, 0] = 0
        xt.requires_grad_(True)
        xr.requires_grad_(True)
        out = fn(xt, vt, xr, vr)
        cot = randn(g, *out.shape)
        gxt, gxr = grads([out], [cot], [xt, xr])
        cases.append(dict(x_t=npy(xt), v_t=npy(vt), x_r=npy(xr), v_r=npy(vr), out=npy(out), cot=npy(cot),
                          g_x_t=npy(gxt), g_x_r=npy(gxr)))
    save("corr4d", cases)


# ----------------------------------------------------------------------------- A2 corr1x1
def make_corr1x1(ref):
    fn = ref["utils.correlation"].compute_masked_correlation
    cases = []
    for seed, B, C, H, W in [(0, 2, 8, 5, 7), (1, 1, 3, 4, 4), (2, 3, 32, 2, 9)]:
        g = gen(200 + seed)
        f1, f2 = randn(g, B, C, H, W), randn(g, B, C, H, W)
        m1, m2 = bern(g, 0.8, B, 1, H, W), bern(g, 0.8, B, 1, H, W)
        f1.requires_grad_(True)
        f2.requires_grad_(True)
        out = fn(f1, f2, m1, m2)
        cot = randn(g, *out.shape)
        g1, g2 = grads([out], [cot], [f1, f2])
        cases.append(dict(f1=npy(f1), f2=npy(f2), m1=npy(m1), m2=npy(m2), out=npy(out), cot=npy(cot),
                          g_f1=npy(g1), g_f2=npy(g2)))
    save("corr1x1", cases)


# ----------------------------------------------------------------------------- A3 align_set
def identity_grid(H, W):
    ys = torch.linspace(-1, 1, H).view(H, 1).expand(H, W)
    xs = torch.linspace(-1, 1, W).view(1, W).expand(H, W)
    return torch.stack([xs, ys], dim=2)


def make_align_set(ref):
    fn = ref["utils.flow"].align_set
    cases = []
    for seed, B, F_, H, W, amp in [(0, 2, 2, 8, 8, 0.3), (1, 1, 3, 17, 23, 0.6), (2, 1, 2, 9, 9, 0.0), (3, 2, 1, 5, 12, 1.2)]:
        g = gen(300 + seed)
        x = rand(g, B, 3, F_, H, W)
        v = bern(g, 0.7, B, 1, F_, H, W)
        flow = identity_grid(H, W).view(1, 1, H, W, 2).repeat(B, F_, 1, 1, 1) + amp * (rand(g, B, F_, H, W, 2) * 2 - 1)
        if seed == 2:
            # exact half-pixel positions (nearest ties -> half-to-even) and exact borders; W-1 = 8 so all are exact
            ks = torch.arange(W).float()
            half = (2 * ks + 1) / (W - 1) - 1
            flow[0, 0, 0, :, 0] = half
            flow[0, 0, 1, :, 1] = half
            flow[0, 0, 2, :, 0] = half
            flow[0, 0, 2, :, 1] = half.flip(0)
            flow[0, 1, 3, :, 0] = -1.0
            flow[0, 1, 4, :, 0] = 1.0
            flow[0, 1, 5, :, 1] = -1.0
            flow[0, 1, 6, :, 1] = 1.0
            flow[0, 1, 7, :, 0] = torch.tensor([-1.5, -1.25, -1.0625, 1.0625, 1.25, 1.5, 3.0, -3.0, 0.0])
            flow[0, 1, 8, :, 1] = torch.tensor([-1.5, -1.25, -1.0625, 1.0625, 1.25, 1.5, 3.0, -3.0, 0.0])
        x.requires_grad_(True)
        flow.requires_grad_(True)
        xa, va = fn(x, v, flow)
        cot = randn(g, *xa.shape)
        gx, gflow = grads([xa], [cot], [x, flow])
        cases.append(dict(x=npy(x), v=npy(v), flow=npy(flow), x_aligned=npy(xa), v_aligned=npy(va), cot=npy(cot),
                          g_x=npy(gx), g_flow=npy(gflow)))
    save("align_set", cases)


# ----------------------------------------------------------------------------- A4 CPN affine align
def make_cpn_align(ref):
    """Runs the REAL CPNet.align body (models/cpn.py:266-298) with the conv encoder / regressor replaced by
    stand-ins that return a prescribed theta, so only affine_grid + grid_sample (+ >0.5) are exercised."""
    cpn = ref["models.cpn"]
    cases = []
    for seed, B, F_, H, W, amp in [(0, 2, 4, 12, 20, 0.15), (1, 1, 3, 9, 7, 0.4), (2, 1, 5, 16, 16, 0.05)]:
        g = gen(400 + seed)
        t = F_ // 2
        r_list = [i for i in range(F_) if i != t]
        x = rand(g, B, 3, F_, H, W)
        y = rand(g, B, 3, F_, H, W)
        m = 1 - blocky_mask(g, B, F_, H - H % 2, W - W % 2, 2, 0.7) if (H % 2 == 0 and W % 2 == 0) else bern(g, 0.3, B, 1, F_, H, W)
        theta = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]).view(1, 2, 3) + amp * randn(g, B * len(r_list), 2, 3)
        theta.requires_grad_(True)
        x.requires_grad_(True)
        fake = types.SimpleNamespace(
            alignment_encoder=lambda a, b_: torch.zeros(a.size(0), 1, 1, 1),
            alignment_regressor=lambda a, b_: theta,
            training=False,
        )
        xa, va, ya = cpn.CPNet.align(fake, x, m, y, t, r_list)
        cot = randn(g, *xa.shape)
        gx, gth = grads([xa], [cot], [x, theta])
        cases.append(dict(x=npy(x), m=npy(m), y=npy(y), theta=npy(theta), t=np.asarray(t), r_list=np.asarray(r_list),
                          x_aligned=npy(xa), v_aligned=npy(va), y_aligned=npy(ya), cot=npy(cot),
                          g_x=npy(gx), g_theta=npy(gth)))
    save("cpn_align", cases)


# ----------------------------------------------------------------------------- A5 masked_softmax
def make_masked_softmax(ref):
    fn = ref["models.cpn"].masked_softmax
    fn2 = ref["models.cpn_original"].CM_Module().masked_softmax
    cases = []
    for seed, shape, scale in [(0, (2, 4, 3, 5, 7), 1.0), (1, (1, 2, 5, 3, 3), 30.0), (2, (2, 1, 10, 4, 6), 5.0)]:
        g = gen(500 + seed)
        vec = scale * randn(g, *shape)
        mask = bern(g, 0.6, shape[0], 1, *shape[2:])
        mask[:, :, :, 0, 0] = 0  # visible in no reference
        vec.requires_grad_(True)
        out = fn(vec, mask, 2)
        out2 = fn2(vec, mask, 2)
        assert torch.equal(out, out2)
        cot = randn(g, *out.shape)
        (gv,) = grads([out], [cot], [vec])
        cases.append(dict(vec=npy(vec), mask=npy(mask), out=npy(out), cot=npy(cot), g_vec=npy(gv)))
    save("masked_softmax", cases)


# ----------------------------------------------------------------------------- A6/A7/A8/A9 context matching
def make_context_matching(ref):
    cpn, cpo = ref["models.cpn"], ref["models.cpn_original"]
    cases = []
    specs = [
        # seed, B, C, R, h, w, cell(=mask stride), feat scale
        (0, 2, 8, 3, 5, 7, 8, 0.3),
        (1, 1, 4, 4, 6, 5, 8, 1.0),
        (2, 2, 6, 2, 4, 9, 4, 0.2),
        (3, 1, 8, 10, 3, 4, 8, 0.5),
    ]
    for seed, B, C, R, h, w, cell, fs in specs:
        g = gen(600 + seed)
        H, W = h * cell, w * cell
        c_feats = fs * randn(g, B, C, R + 1, h, w)
        half = cell // 2
        v_t = blocky_mask(g, B, 1, H, W, half, 0.6)[:, :, 0]
        v_al = blocky_mask(g, B, R, H, W, half, 0.65)
        v_al[0, :, 1] = 0.0  # a reference with zero overlap (v_sum_zeros)
        v_al[:, :, :, :cell, :cell] = 0.0  # a pixel visible in no reference
        for variant, normalize in [("default", True), ("default", False), ("complete", True), ("complete", False), ("cm_module", True)]:
            cf = c_feats.clone().requires_grad_(True)
            if variant == "default":
                p_in, c_mask, ref_imp = cpn.CPNContextMatchingDefault()(cf, v_t, v_al, normalize=normalize)
            elif variant == "complete":
                p_in, c_mask, ref_imp = cpn.CPNContextMatchingComplete()(cf, v_t, v_al, normalize=normalize)
            else:
                p_in, c_mask = cpo.CM_Module()(cf, v_t, v_al)
                ref_imp = None
            gg = gen(700 + seed)
            cot1, cot2 = randn(gg, *p_in.shape), randn(gg, *c_mask.shape)
            (gcf,) = grads([p_in, c_mask], [cot1, cot2], [cf])
            cases.append(dict(variant=np.asarray(variant), normalize=np.asarray(normalize), c_feats=npy(c_feats), v_t=npy(v_t),
                              v_aligned=npy(v_al), p_in=npy(p_in), c_mask=npy(c_mask), ref_importance=npy(ref_imp),
                              cot_p_in=npy(cot1), cot_c_mask=npy(cot2), g_c_feats=npy(gcf)))
    save("context_matching", cases)


def make_mask_resize(ref):
    """The `(F.interpolate(v, (h,w), bilinear, align_corners=False) > 0.5)` step of models/cpn.py:98,104 at
    non-integer scales (generated through CPNContextMatchingDefault: the resized target map can be read back
    from p_in only indirectly, so it is captured here by calling the same ATen op the reference line calls)."""
    import torch.nn.functional as F
    cases = []
    for seed, H, W, h, w in [(0, 240, 424, 30, 53), (1, 37, 51, 5, 7), (2, 64, 64, 16, 16), (3, 30, 53, 30, 53), (4, 17, 9, 20, 13)]:
        g = gen(800 + seed)
        v = (rand(g, 2, 1, H, W) < 0.5).float()
        if seed == 0:
            v = blocky_mask(g, 2, 1, H, W, 4, 0.5)[:, :, 0]
        out = (F.interpolate(v, size=(h, w), mode="bilinear", align_corners=False) > 0.5).float()
        cases.append(dict(v=npy(v).astype(np.uint8), size=np.asarray([h, w]), out=npy(out).astype(np.uint8)))
    save("mask_resize", cases)


# ----------------------------------------------------------------------------- A10-A13 flow utilities
def make_flow_utils(ref):
    fl = ref["utils.flow"]
    cases = []
    # corr_to_flow with ties
    g = gen(900)
    corr = randn(g, 1, 2, 4, 4, 4, 4)
    corr[0, 0, 0, 0] = 0.25  # constant row: argmax must be index 0
    corr[0, 0, 1, 1] = corr[0, 0, 1, 1].clamp(max=0.5)
    corr[0, 0, 1, 1, 2, 3] = 0.5
    corr[0, 0, 1, 1, 3, 1] = 0.5  # duplicate maximum: first index wins
    corr[0, 1, 2, 2] = -1.0
    cases.append(dict(op=np.asarray("corr_to_flow"), corr=npy(corr), out=npy(fl.corr_to_flow(corr))))
    corr = randn(g, 2, 1, 5, 5, 5, 5)
    cases.append(dict(op=np.asarray("corr_to_flow"), corr=npy(corr), out=npy(fl.corr_to_flow(corr))))
    # abs <-> rel
    flow = randn(g, 2, 3, 5, 7, 2)
    cases.append(dict(op=np.asarray("flow_abs_to_relative"), flow=npy(flow), out=npy(fl.flow_abs_to_relative(flow))))
    cases.append(dict(op=np.asarray("flow_relative_to_abs"), flow=npy(flow), out=npy(fl.flow_relative_to_abs(flow))))
    # crop_flow
    for (H, W, cs, cp) in [(8, 10, (3, 4), (1, 2)), (16, 16, (16, 16), (0, 0)), (9, 6, (4, 5), (5, 1))]:
        flow = randn(g, 2, 2, H, W, 2)
        out = fl.crop_flow(flow.clone(), cs, cp)
        cases.append(dict(op=np.asarray("crop_flow"), flow=npy(flow), crop_size=np.asarray(cs), crop_position=np.asarray(cp), out=npy(out)))
    # resize_flow (default interpolate mode = legacy nearest)
    for (H, W, size) in [(8, 8, (4, 4)), (8, 8, (3, 5)), (5, 7, (16, 16)), (64, 64, (16, 16)), (30, 53, (7, 11))]:
        flow = randn(g, 1, 2, H, W, 2)
        out = fl.resize_flow(flow, size)
        cases.append(dict(op=np.asarray("resize_flow"), flow=npy(flow), size=np.asarray(size), out=npy(out)))
    save("flow_utils", cases)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref = load_reference()
    make_corr4d(ref)
    make_corr1x1(ref)
    make_align_set(ref)
    make_cpn_align(ref)
    make_masked_softmax(ref)
    make_context_matching(ref)
    make_mask_resize(ref)
    make_flow_utils(ref)
    print("torch", torch.__version__)


if __name__ == "__main__":
    main()
