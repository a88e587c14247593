#!/usr/bin/env python3
"""Checksums of the REAL reference's outputs at the BASELINE.json config shapes -> tests/golden/configs.npz.

Run ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden_configs.py

Inputs come from tests/config_cases.py (numpy PCG64, regenerated from the seed by the tests); the reference functions are
the real files, imported by path as in make_golden.py / make_golden_sequences.py (scipy restatement of cv2.dilate, empty
auto-attribute modules for skeltorch & co.).  Stored per output: the checksums of tests/config_cases.py (data only).

    corr_c2     compute_masked_4d_correlation, one pair 64x64, C=256, fwd + autograd            utils/correlation.py:28-54
    corr_c3     same at 16x16x512, 4 references, post-ReLU-like features
    align_c3    align_set, 4 references at 256x256, fwd + autograd (x, flow)                     utils/flow.py:93-101
    cpn_c4      the body of CPNet.align at 240x424, 10 references (regressor -> prescribed theta) models/cpn.py:266-298
    ctx_c4/*    CPNContextMatchingDefault / Complete, CM_Module at (1,128,11,30,53), 240x424 masks models/cpn.py:94-195,
                                                                                                  models/cpn_original.py:201-243
    loss_c3     ThesisAlignmentRunner.compute_loss with the seeded stand-in extractor            thesis_dfpn/runner.py:162-206
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden  # noqa: E402  (the cv2 restatement)
import make_golden_sequences as mgs  # noqa: E402  (hollow modules for absent third-party packages)
from tests import config_cases as cc  # noqa: E402


def load_reference():
    make_golden._stub_modules()                      # cv2 (scipy restatement) + skimage stubs
    sys.meta_path.insert(0, mgs._AbsentFinder())     # skeltorch, progressbar, torchvision, ...
    for pkg in ["utils", "models", "thesis", "thesis_dfpn", "thesis_cpn", "thesis_inpainting"]:
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(make_golden.REF, pkg)]
        sys.modules[pkg] = m
    names = ["utils.correlation", "utils.flow", "utils.transforms", "models.cpn", "models.cpn_original", "utils.losses", "thesis_dfpn.runner"]
    return {n: importlib.import_module(n) for n in names}


def T(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.requires_grad_(True) if grad else t


def npy(t):
    return t.detach().contiguous().numpy()


def put(flat, prefix, checksum):
    for k, v in checksum.items():
        flat["%s/%s" % (prefix, k)] = np.asarray(v)


def main():
    torch.manual_seed(0)
    ref = load_reference()
    flat = {}

    # ---- A1
    for name, make in [("corr_c2", cc.corr_c2), ("corr_c3", cc.corr_c3)]:
        d = make()
        xt, xr = T(d["x_t"], True), T(d["x_r"], True)
        out = ref["utils.correlation"].compute_masked_4d_correlation(xt, T(d["v_t"]), xr, T(d["v_r"]))
        gt, gr = torch.autograd.grad((out * T(d["cot"])).sum(), [xt, xr])
        put(flat, name + "/out", cc.float_checksum(npy(out)))
        put(flat, name + "/g_x_t", cc.float_checksum(npy(gt)))
        put(flat, name + "/g_x_r", cc.float_checksum(npy(gr)))
        print(name, tuple(out.shape))

    # ---- A3
    d = cc.align_c3()
    x, fl = T(d["x"], True), T(d["flow"], True)
    xa, va = ref["utils.flow"].align_set(x, T(d["v"]), fl)
    gx, gf = torch.autograd.grad((xa * T(d["cot"])).sum(), [x, fl])
    put(flat, "align_c3/x_aligned", cc.float_checksum(npy(xa)))
    put(flat, "align_c3/v_aligned", cc.mask_checksum(npy(va)))
    put(flat, "align_c3/g_x", cc.float_checksum(npy(gx)))
    put(flat, "align_c3/g_flow", cc.float_checksum(npy(gf)))
    print("align_c3", tuple(xa.shape), "outside: %.3f" % float((npy(fl).__abs__() > 1).any(-1).mean()))

    # ---- A4: the real CPNet.align body, encoder / regressor replaced by a prescribed theta
    d = cc.cpn_align_c4()
    x, theta = T(d["x"], True), T(d["theta"], True)
    fake = types.SimpleNamespace(alignment_encoder=lambda a, b_: torch.zeros(a.size(0), 1, 1, 1),
                                 alignment_regressor=lambda a, b_: theta, training=False)
    xa, va, ya = ref["models.cpn"].CPNet.align(fake, x, T(d["m"]), T(d["y"]), d["t"], d["r_list"])
    gx, gth = torch.autograd.grad((xa * T(d["cot"])).sum(), [x, theta])
    put(flat, "cpn_c4/x_aligned", cc.float_checksum(npy(xa)))
    put(flat, "cpn_c4/y_aligned", cc.float_checksum(npy(ya)))
    put(flat, "cpn_c4/v_aligned", cc.mask_checksum(npy(va)))
    put(flat, "cpn_c4/g_x", cc.float_checksum(npy(gx)))
    flat["cpn_c4/g_theta"] = npy(gth)
    print("cpn_c4", tuple(xa.shape))

    # ---- A6 / A7 / A8
    for variant, normalize in cc.CTX_VARIANTS:
        d = cc.ctx_c4(normalize)
        cf = T(d["c_feats"], True)
        if variant == "default":
            p_in, c_mask, imp = ref["models.cpn"].CPNContextMatchingDefault()(cf, T(d["v_t"]), T(d["v_aligned"]), normalize=normalize)
        elif variant == "complete":
            p_in, c_mask, imp = ref["models.cpn"].CPNContextMatchingComplete()(cf, T(d["v_t"]), T(d["v_aligned"]), normalize=normalize)
        else:
            p_in, c_mask = ref["models.cpn_original"].CM_Module()(cf, T(d["v_t"]), T(d["v_aligned"]))
            imp = None
        c1, c2 = cc.ctx_cots(tuple(p_in.shape), tuple(c_mask.shape))
        (g,) = torch.autograd.grad((p_in * T(c1)).sum() + (c_mask * T(c2)).sum(), [cf])
        key = "ctx_c4/%s_%d" % (variant, int(normalize))
        put(flat, key + "/p_in", cc.float_checksum(npy(p_in)))
        put(flat, key + "/c_mask", cc.float_checksum(npy(c_mask)))
        if imp is not None:
            put(flat, key + "/ref_importance", cc.float_checksum(npy(imp)))
        put(flat, key + "/g_c_feats", cc.float_checksum(npy(g)))
        print(key, tuple(p_in.shape), "c_mask mean %.3f" % float(c_mask.mean()))

    # ---- N2: ThesisAlignmentRunner.compute_loss
    from master_thesis_amd import harness_c3
    d = cc.loss_c3()
    torch.manual_seed(cc_seed := 77)
    vgg = harness_c3.FeatureStandIn()
    losses = ref["utils.losses"].LossesUtils(None, "cpu")
    xs, vs, ys = [T(d["x%d" % k]) for k in range(3)], [T(d["v%d" % k]) for k in range(3)], [T(d["y%d" % k]) for k in range(3)]
    xas = [T(d["xa%d" % k]) for k in range(3)]
    flows, flows_gt = [T(d["flow%d" % k]) for k in range(3)], [T(d["flow_gt%d" % k]) for k in range(3)]
    total, items = ref["thesis_dfpn.runner"].ThesisAlignmentRunner.compute_loss(
        vgg, losses, T(d["corr"]), xs, vs, ys, xas, flows, flows_gt, torch.from_numpy(d["flows_use"]), d["t"], d["r_list"])
    flat["loss_c3/total"] = np.asarray(float(total), dtype=np.float64)
    flat["loss_c3/items"] = np.asarray([float(i) for i in items], dtype=np.float64)
    flat["loss_c3/vgg_seed"] = np.asarray(cc_seed)
    print("loss_c3", float(total), [round(float(i), 6) for i in items])

    path = os.path.join(HERE, "configs.npz")
    np.savez_compressed(path, **flat)
    print("configs.npz %.1f KiB, torch %s" % (os.path.getsize(path) / 1024, torch.__version__))


if __name__ == "__main__":
    main()
