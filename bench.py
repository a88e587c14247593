#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json's metric, aligned frame-pairs/sec through corr + warp + match, forward and backward.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU, under torch.distributed.run or, without a
                                                            launcher, as N child processes this script starts itself)

One pass of the hot path takes PAIRS_PER_PASS = 640 synthetic frame pairs per GPU (inputs resident in HBM) through the three
legs, each at the shape its BASELINE config names, through the drop-in functions and the C ABI:

  corr   C2: compute_masked_4d_correlation fwd + bwd, 64x64 maps, C=256, batches of 16 pairs        (40 batches per pass)
  warp   C3: the six align_set calls of train_step_propagate (ground-truth and predicted flow at 16 / 64 / 256 px) and the
             grid-gradient backward of the three predicted ones, 128 pairs per batch                   (5 batches per pass)
  match  C4: CPN temporal copying, 32 targets x 10 references at 240x424: affine warp (theta -> positions in the kernel) +
             context matching incl. ref_importance, backward to theta and to the features              (2 batches per pass)

A step is `inner` such passes (inner >= 1 is chosen so that the timed region lasts at least a second whatever --steps is).
Ranks work on independent pairs: no data-path collective, weak scaling.  Rank 0 prints ONE JSON line: whole-job pairs/s,
per-leg GPU time inside the timed region (HIP events on the launch stream), `legs`: every dominant kernel / entry point
timed cache-cold through the C ABI against its roofline, `roofline`: the dominant kernel of the step, and `cpu_baseline`:
the numpy oracle on the host cores per leg (all BLAS threads and one thread).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks: /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_F32_MATRIX_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA

METRIC = "aligned frame-pairs/sec (corr+warp+match fwd+bwd) at 1/2/4/8 GPUs; % roofline"
PAIRS_PER_PASS = 640
CORR = dict(B=16, C=256, F=1, S=64)               # C2
WARP = dict(B=32, R=4, S=256)                     # C3: 32 sequences x 4 references
C4 = dict(B=32, C=128, R=10, h=30, w=53, H=240, W=424)
N_CORR, N_WARP, N_C4 = PAIRS_PER_PASS // 16, PAIRS_PER_PASS // 128, PAIRS_PER_PASS // 320


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------------ synthetic inputs
def _blocky(g, shape, cell, p):
    m = (torch.rand(*shape, generator=g) < p).float()
    return m.repeat_interleave(cell, -2).repeat_interleave(cell, -1)


def _ident(S):
    lin = torch.linspace(-1, 1, S)
    ys, xs = torch.meshgrid(lin, lin, indexing="ij")
    return torch.stack([xs, ys], -1)


def make_corr(dev, g):
    B, C, F, S = CORR["B"], CORR["C"], CORR["F"], CORR["S"]
    xt = torch.randn(B, C, S, S, generator=g)
    xr = torch.randn(B, C, F, S, S, generator=g)
    vt = (torch.rand(B, 1, S, S, generator=g) >= 0.1).float()
    vr = (torch.rand(B, 1, F, S, S, generator=g) >= 0.1).float()
    cot = torch.randn(B, F, S, S, S, S, generator=g)
    xt, vt, xr, vr, cot = [t.to(dev) for t in (xt, vt, xr, vr, cot)]
    return dict(xt=xt.requires_grad_(True), vt=vt, xr=xr.requires_grad_(True), vr=vr, cot=cot)


def make_warp_set(dev, g):
    """One C3 batch: frames / visibility at 16, 64, 256 px, ground-truth and predicted flows (smooth similarity + offset)."""
    B, R = WARP["B"], WARP["R"]
    d = {}
    for s in (16, 64, 256):
        d["x%d" % s] = torch.rand(B, 3, R, s, s, generator=g).to(dev)
        d["v%d" % s] = _blocky(g, (B, 1, R, s // 8, s // 8), 8, 0.85).to(dev)
        for name in ("gt", "pr"):
            scale = 1 + 0.1 * (torch.rand(B, R, 1, 1, 1, generator=g) * 2 - 1)
            shift = 0.2 * (torch.rand(B, R, 1, 1, 2, generator=g) * 2 - 1)
            d["%s%d" % (name, s)] = (_ident(s)[None, None] * scale + shift).to(dev).contiguous()
        d["pr%d" % s].requires_grad_(True)
        d["cot%d" % s] = torch.randn(B, R, 3, s, s, generator=g).to(dev).transpose(1, 2)   # laid out like the warp's output
    return d


def make_c4_set(dev, g):
    B, C, R, h, w, H, W = (C4[k] for k in ("B", "C", "R", "h", "w", "H", "W"))
    d = {"r_list": [i for i in range(R + 1) if i != R // 2]}
    d["x"] = torch.rand(B, 3, R + 1, H, W, generator=g).to(dev)
    d["m"] = (1 - _blocky(g, (B, 1, R + 1, H // 8, W // 8), 8, 0.85)).to(dev)
    d["theta"] = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + 0.05 * torch.randn(B * R, 2, 3, generator=g)).to(dev).requires_grad_(True)
    d["cf"] = (0.05 * torch.randn(B, C, R + 1, h, w, generator=g)).to(dev).requires_grad_(True)
    d["cot_x"] = torch.randn(B, R, 3, H, W, generator=g).to(dev).transpose(1, 2)
    d["cot_p"] = torch.randn(B, 2 * C + 1, h, w, generator=g).to(dev)
    return d


# ------------------------------------------------------------------------------------------------ the three legs (drop-in functions)
def corr_leg(d):
    from master_thesis_amd.utils import correlation
    out = correlation.compute_masked_4d_correlation(d["xt"], d["vt"], d["xr"], d["vr"])
    return torch.autograd.grad(out, [d["xt"], d["xr"]], grad_outputs=d["cot"])


def warp_leg(d):
    """thesis_dfpn/runner.py:118-120,132-134 and the backward of the reconstruction losses into the predicted flows."""
    from master_thesis_amd.utils import flow
    with torch.no_grad():
        for s in (16, 64, 256):
            flow.align_set(d["x%d" % s], d["v%d" % s], d["gt%d" % s])
    outs = [flow.align_set(d["x%d" % s], d["v%d" % s], d["pr%d" % s])[0] for s in (16, 64, 256)]
    return torch.autograd.grad(outs, [d["pr%d" % s] for s in (16, 64, 256)], grad_outputs=[d["cot%d" % s] for s in (16, 64, 256)])


def c4_leg(d, module):
    """models/cpn.py:280-295 (affine warp of frames + visibility) and :319 (context matching, normalize=False), with the
    backward to theta (through the warp) and to the encoder features (through the matching)."""
    from master_thesis_amd.models import cpn
    x_al, v_al, _ = cpn.affine_align(d["x"], d["m"], None, d["theta"], d["r_list"], training=True)
    t = C4["R"] // 2
    p_in, c_mask, ref_imp = module(d["cf"], 1 - d["m"][:, :, t], v_al, normalize=False)
    return torch.autograd.grad([x_al, p_in], [d["theta"], d["cf"]], grad_outputs=[d["cot_x"], d["cot_p"]])


# ------------------------------------------------------------------------------------------------ cache-cold kernel timing (C ABI)
def time_rotating(launchers, stream, rounds=6, warm=1):
    """Average duration of one launch: the launchers (same call on different buffer sets, together far larger than the
    256 MB Infinity Cache) run back to back, HIP events on the launch stream around the whole run."""
    for _ in range(warm):
        for fn in launchers:
            fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(rounds):
        for fn in launchers:
            fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / (rounds * len(launchers))


def measure_legs(dev, corr, warp_sets, c4_sets, groups=("corr", "warp", "c4", "infer")):
    """Every dominant kernel / entry point of the step alone, cache-cold, against its roofline (SURVEY 8d byte counts)."""
    from master_thesis_amd import _lib
    lib = _lib.lib()
    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    legs = {}

    def add(name, kernel, ms, nbytes=None, flops=None, note=None, peak_tf=PEAK_F32_MATRIX_TFLOPS):
        r = {"kernel": kernel, "ms_per_launch": round(ms, 4)}
        if flops is not None:
            tf = flops / (ms * 1e-3) / 1e12
            r.update(bound="mfma", algorithmic_flops=int(flops), achieved=round(tf, 2), peak=peak_tf, unit="TFLOP/s",
                     frac=round(tf / peak_tf, 4))
        else:
            gbs = nbytes / (ms * 1e-3) / 1e9
            r.update(bound="hbm", algorithmic_bytes=int(nbytes), achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                     frac=round(gbs / PEAK_HBM_GBS, 4))
        if note:
            r["note"] = note
        legs[name] = r

    if "corr" in groups:
        _legs_corr(dev, corr, add, lib, stream, sp)
    if "warp" in groups:
        _legs_warp(dev, warp_sets, add, stream, sp)
    if "c4" in groups:
        _legs_c4(dev, c4_sets, add, lib, stream, sp)
    if "infer" in groups:
        _legs_infer(dev, legs, stream)
    return legs


def _legs_corr(dev, corr, add, lib, stream, sp):
    from master_thesis_amd import _lib
    # ---- corr (2.3 GB working set: cold by itself)
    B, C, F, S = CORR["B"], CORR["C"], CORR["F"], CORR["S"]
    P = S * S
    xt, xr, vt, vr, cot = corr["xt"].detach(), corr["xr"].detach(), corr["vt"], corr["vr"], corr["cot"]
    ws = torch.empty(lib.tcl_corr4d_fwd_workspace_bytes(B, C, F, P, 0), dtype=torch.uint8, device=dev)
    ws2 = torch.empty(max(256, lib.tcl_corr4d_bwd_workspace_bytes(B, C, F, P, 0)), dtype=torch.uint8, device=dev)
    out = torch.empty(B, F, P, P, device=dev)
    gt, gr = torch.empty_like(xt), torch.empty_like(xr)
    _lib.call("tcl_corr4d_fwd", _p(xt), _p(vt), _p(xr), _p(vr), _p(out), B, C, F, P, 0, _p(ws), ws.numel(), sp)
    flops = 2.0 * P * P * C * B * F            # 8.59 GFLOP per pair forward
    # MFMA-bound legs: ~0.1 s of the same launch first, so that the timed launches run at the sustained clock the >= 1 s timed
    # region of the step sees (the first ~20 ms after an idle gap run 10 % slower)
    ms = time_rotating([lambda: _lib.call("tcl_corr4d_contract", _p(ws), _p(out), B, C, F, P, 0, sp)], stream, rounds=40, warm=80)
    add("corr_fwd_contraction", "gemm_f32_kernel<128,128,2,2,...,FULL> (tcl_corr4d_contract)", ms, flops=flops)
    for prec, name in ((2, "bf16x3"), (1, "bf16")):
        # the variant's own forward first: its workspace holds the bf16 (hi, lo) operand images the contraction reads
        wsv = torch.empty(lib.tcl_corr4d_fwd_workspace_bytes(B, C, F, P, prec), dtype=torch.uint8, device=dev)
        _lib.call("tcl_corr4d_fwd", _p(xt), _p(vt), _p(xr), _p(vr), _p(out), B, C, F, P, prec, _p(wsv), wsv.numel(), sp)
        ms = time_rotating([lambda: _lib.call("tcl_corr4d_contract", _p(wsv), _p(out), B, C, F, P, prec, sp)], stream, rounds=20, warm=3)
        nsl = 2 if prec == 2 else 1
        add("corr_fwd_contraction_" + name, "gemm_bf16op_kernel<128,128,2,2,NS=%d> (variant, not the headline)" % nsl, ms,
            nbytes=B * F * P * P * 4 + (B + B * F) * C * P * 2 * nsl, note="bound by the store of the fp32 volume (1 GiB per launch); bf16 operand images")
        # N1: the same contraction fused with corr_to_flow: no volume, bound by the bf16 matrix pipe
        wsa = torch.empty(lib.tcl_corr4d_argmax_flow_workspace_bytes(B, C, F, P, prec), dtype=torch.uint8, device=dev)
        fl = torch.empty(B, F, P, 2, device=dev)
        ms = time_rotating([lambda: _lib.call("tcl_corr4d_argmax_flow", _p(xt), _p(vt), _p(xr), _p(vr), _p(fl), B, C, F, S, prec, _p(wsa), wsa.numel(), sp)],
                           stream, rounds=20, warm=20)
        nm = 3 if prec == 2 else 1
        add("corr_argmax_flow_" + name, "tcl_corr4d_argmax_flow(%s)" % name, ms, flops=flops * nm, peak_tf=PEAK_BF16_TFLOPS,
            note="whole entry point: 2 x corr_normalise + gemm_bf16op_kernel<EPI_ARGMAX> + argmax_tiles_to_flow; flops = the bf16 MFMA flops "
                 "issued (%d x the algorithmic 2 P P C), against the dense bf16 peak" % nm)
        del wsv, wsa, fl
    wsa = torch.empty(lib.tcl_corr4d_argmax_flow_workspace_bytes(B, C, F, P, 0), dtype=torch.uint8, device=dev)
    fl = torch.empty(B, F, P, 2, device=dev)
    ms = time_rotating([lambda: _lib.call("tcl_corr4d_argmax_flow", _p(xt), _p(vt), _p(xr), _p(vr), _p(fl), B, C, F, S, 0, _p(wsa), wsa.numel(), sp)],
                       stream, rounds=20, warm=40)
    add("corr_argmax_flow_f32", "tcl_corr4d_argmax_flow(f32)", ms, flops=flops, note="2 x corr_normalise + gemm_f32_kernel<EPI_ARGMAX> + argmax_tiles_to_flow")
    del wsa, fl
    ms = time_rotating([lambda: _lib.call("tcl_corr4d_bwd", _p(cot), _p(xt), _p(vt), _p(xr), _p(vr), _p(ws), _p(gt), _p(gr), B, C, F, P, 0,
                                          _p(ws2), ws2.numel(), sp)], stream, rounds=20, warm=40)
    add("corr_bwd", "2 x gemm_f32_wide_kernel (256x256 tiles) + corr_normalise_bwd (tcl_corr4d_bwd)", ms, flops=2 * flops)
    # the two products alone (each still followed by its normalise-backward kernel, ~38 us of the figure)
    for name, a_t, a_r in (("corr_bwd_d_a", gt, None), ("corr_bwd_d_r", None, gr)):
        ms = time_rotating([lambda: _lib.call("tcl_corr4d_bwd", _p(cot), _p(xt), _p(vt), _p(xr), _p(vr), _p(ws), _p(a_t), _p(a_r), B, C, F, P, 0,
                                              _p(ws2), ws2.numel(), sp)], stream, rounds=20, warm=20)
        add(name, "gemm_f32_wide_kernel<%s> + corr_normalise_bwd" % ("k-contiguous G" if a_t is not None else "k-major G"), ms, flops=flops)
    del out, ws, ws2, gt, gr


def _legs_warp(dev, warp_sets, add, stream, sp):
    from master_thesis_amd import _lib
    # ---- C3 warp at 256 px over the rotating sets
    Bw, R, S = WARP["B"], WARP["R"], 256
    npix = Bw * R * S * S
    bufs = []
    for d in warp_sets:
        x, v, fl, cotw = d["x256"], d["v256"], d["pr256"].detach(), d["cot256"]
        xo, vo = torch.empty(Bw, R, 3, S, S, device=dev), torch.empty(Bw, R, 1, S, S, device=dev)
        bufs.append((x, v, fl, cotw, xo, vo, torch.empty_like(fl), torch.zeros_like(x)))
    wsw = torch.empty(256, dtype=torch.uint8, device=dev)

    def wfwd(b):
        x, v, fl, cotw, xo, vo, gg, gx = b
        return lambda: _lib.call("tcl_warp_fwd", _p(x), x.stride(0), x.stride(2), x.stride(1), _p(v), v.stride(0), v.stride(2), _p(fl), None,
                                 _p(xo), xo.stride(0), xo.stride(1), xo.stride(2), _p(vo), vo.stride(0), vo.stride(1), Bw, R, 3, S, S, 1, 1, sp)

    def wbwd(b, with_gx):
        x, v, fl, cotw, xo, vo, gg, gx = b

        def run():
            if with_gx:
                gx.zero_()
            _lib.call("tcl_warp_bwd", _p(cotw), cotw.stride(0), cotw.stride(2), cotw.stride(1), _p(x), x.stride(0), x.stride(2), x.stride(1),
                      _p(fl), None, _p(gx) if with_gx else None, gx.stride(0), gx.stride(2), gx.stride(1), _p(gg), None,
                      Bw, R, 3, S, S, 1, _p(wsw), wsw.numel(), sp)
        return run
    add("warp_fwd_c3", "warp_fwd_kernel<grid,nearest,C3> (align_set, 128 pairs at 256x256)", time_rotating([wfwd(b) for b in bufs], stream),
        nbytes=npix * (4 * 4 + 8 + 4 * 4), note="read x(3)+v(1)+grid(2), write x(3)+v(1)")
    add("warp_bwd_grid_c3", "warp_bwd_kernel<grid,no gx> (what the reference's training exercises)",
        time_rotating([wbwd(b, False) for b in bufs], stream), nbytes=npix * (3 * 4 + 3 * 4 + 8 + 8), note="read grad_out(3)+x(3)+grid(2), write grad_grid(2)")
    add("warp_bwd_gx_c3", "warp_bwd_kernel<grid,gx> + memset of grad_x (scatter-add)",
        time_rotating([wbwd(b, True) for b in bufs], stream), nbytes=npix * (3 * 4 + 3 * 4 + 8 + 8 + 3 * 4),
        note="as above + write grad_x(3); north_star's scatter-add backward, not exercised by the reference")
    del bufs


def _legs_c4(dev, c4_sets, add, lib, stream, sp):
    from master_thesis_amd import _lib
    # ---- C4
    B, C, R, h, w, H, W = (C4[k] for k in ("B", "C", "R", "h", "w", "H", "W"))
    hw, npix = h * w, B * R * H * W
    sets = []
    for d in c4_sets:
        xr_, vr_ = d["x"][:, :, d["r_list"]], (1 - d["m"])[:, :, d["r_list"]].contiguous()
        vt_ = (1 - d["m"])[:, 0, R // 2].contiguous()
        th = d["theta"].detach()
        xo, vo = torch.empty(B, R, 3, H, W, device=dev), torch.empty(B, R, 1, H, W, device=dev)
        cf = d["cf"].detach()
        p_in, c_mask, c_match = torch.empty(B, 2 * C + 1, h, w, device=dev), torch.empty(B, h, w, device=dev), torch.empty(B, R, h, w, device=dev)
        wsf = torch.empty(lib.tcl_ctx_workspace_bytes(B, C, R, h, w), dtype=torch.uint8, device=dev)
        wsb = torch.empty(max(256, lib.tcl_ctx_bwd_workspace_bytes(B, C, R, h, w)), dtype=torch.uint8, device=dev)
        wst = torch.empty(max(256, lib.tcl_warp_bwd_workspace_bytes(B, R, H, W, 1)), dtype=torch.uint8, device=dev)
        sets.append(dict(xr=xr_, vr=vr_, vt=vt_, th=th, xo=xo, vo=vo, cf=cf, p_in=p_in, c_mask=c_mask, c_match=c_match, wsf=wsf, wsb=wsb, wst=wst,
                         cot_x=d["cot_x"], cot_p=d["cot_p"], g_cf=torch.empty_like(cf), g_th=torch.empty_like(th),
                         imp=torch.empty(B, 3, R + 1, H, W, device=dev)))

    def afwd(q):
        x, v, xo, vo = q["xr"], q["vr"], q["xo"], q["vo"]
        return lambda: _lib.call("tcl_warp_fwd", _p(x), x.stride(0), x.stride(2), x.stride(1), _p(v), v.stride(0), v.stride(2), None, _p(q["th"]),
                                 _p(xo), xo.stride(0), xo.stride(1), xo.stride(2), _p(vo), vo.stride(0), vo.stride(1), B, R, 3, H, W, 0, 2, sp)

    def abwd(q):
        x, go = q["xr"], q["cot_x"]
        return lambda: _lib.call("tcl_warp_bwd", _p(go), go.stride(0), go.stride(2), go.stride(1), _p(x), x.stride(0), x.stride(2), x.stride(1),
                                 None, _p(q["th"]), None, 0, 0, 0, None, _p(q["g_th"]), B, R, 3, H, W, 0, _p(q["wst"]), q["wst"].numel(), sp)

    def cfwd(q):
        return lambda: _lib.call("tcl_ctx_fwd", _p(q["cf"]), _p(q["vt"]), _p(q["vo"]), _p(q["p_in"]), _p(q["c_mask"]), _p(q["c_match"]),
                                 B, C, R, h, w, H, W, 0, 0, _p(q["wsf"]), q["wsf"].numel(), sp)

    def cbwd(q):
        return lambda: _lib.call("tcl_ctx_bwd", _p(q["cot_p"]), None, _p(q["cf"]), _p(q["c_match"]), _p(q["wsf"]), _p(q["g_cf"]),
                                 B, C, R, h, w, 0, 0, _p(q["wsb"]), q["wsb"].numel(), sp)

    def imp(q):
        return lambda: _lib.call("tcl_ref_importance", _p(q["c_match"]), _p(q["vt"]), _p(q["imp"]), B, R, h, w, H, W, 0, sp)
    for q in sets:
        afwd(q)()
        cfwd(q)()
    add("affine_warp_fwd_c4", "warp_fwd_kernel<theta,bilinear>0.5,C3> (320 pairs at 240x424)", time_rotating([afwd(q) for q in sets], stream),
        nbytes=npix * (4 * 4 + 4 * 4), note="read x(3)+v(1), write x(3)+v(1); no grid in HBM")
    add("affine_warp_bwd_theta_c4", "warp_bwd_kernel<theta,no gx> + warp_theta_reduce", time_rotating([abwd(q) for q in sets], stream),
        nbytes=npix * (3 * 4 + 3 * 4), note="read grad_out(3)+x(3); theta gradient only (x is data in CPN training)")
    ctx_fwd_bytes = B * (C * (R + 1) * hw * 4 + (R + 1) * H * W * 4 + (2 * C + 1) * hw * 4 + hw * 4)
    add("ctx_fwd_c4", "ctx_resize_masks + ctx_sim + ctx_weights + ctx_aggregate (tcl_ctx_fwd, 32 targets)", time_rotating([cfwd(q) for q in sets], stream),
        nbytes=ctx_fwd_bytes, note="SURVEY 8d: one read of c_feats + masks, one write of p_in (15.1 MB per target)")
    add("ctx_bwd_c4", "ctx_bwd_dmatch + jacobian + finalize + feats (tcl_ctx_bwd)", time_rotating([cbwd(q) for q in sets], stream),
        nbytes=B * (2 * C * (R + 1) * hw * 4 + (2 * C + 1) * hw * 4), note="read c_feats + grad p_in, write grad c_feats")
    add("ref_importance_c4", "ref_importance_kernel<4>", time_rotating([imp(q) for q in sets], stream), nbytes=B * 3 * (R + 1) * H * W * 4 + B * H * W * 4,
        note="write-only visualisation output (A9)")


def _legs_infer(dev, legs, stream):
    """The unit the reference's sequence schedulers issue thousands of times per sequence (thesis_dfpn/runner.py:153-159 after
    the network): B = 1, one reference, 240 x 480.  Wall time per call (launch-bound: host work) against the GPU time of the same
    kernels replayed from a captured graph, for the unfused formulation (1 - m, align_set, 1 - m, v_map: the reference's op list
    on this package's kernels) and the fused single launch."""
    from master_thesis_amd import _ops
    from master_thesis_amd.utils import flow as flow_ops
    g = torch.Generator(device="cpu").manual_seed(99)
    H, W = 240, 480
    x_ref = torch.rand(1, 3, 1, H, W, generator=g).to(dev)
    m_ref = (torch.rand(1, 1, 1, H // 8, W // 8, generator=g) < 0.15).float().repeat_interleave(8, -2).repeat_interleave(8, -1).to(dev)
    m_tgt = (torch.rand(1, 1, H // 8, W // 8, generator=g) < 0.15).float().repeat_interleave(8, -2).repeat_interleave(8, -1).to(dev)
    lin_y, lin_x = torch.linspace(-1, 1, H), torch.linspace(-1, 1, W)
    ys, xs = torch.meshgrid(lin_y, lin_x, indexing="ij")
    fl = (torch.stack([xs, ys], -1)[None, None] * 1.03 + 0.02).to(dev).contiguous()

    def unfused():
        xa, va = flow_ops.align_set(x_ref, 1 - m_ref, fl)
        return xa, va, _ops.vmap(va, 1 - m_tgt)

    def fused():
        return _ops.align_infer(x_ref, m_ref, fl, m_tgt)

    def wall(fn, n=400):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    def replay(fn, n=200):
        try:
            side = torch.cuda.Stream(dev)
            with torch.cuda.stream(side):
                fn()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    for _ in range(10):
                        fn()
                gr.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for _ in range(n // 10):
                    gr.replay()
                e1.record(side)
                e1.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        except Exception as e:   # graph capture unavailable: report the wall time only
            return None
    with torch.no_grad():
        a, b = unfused(), fused()
        same = all(torch.equal(p, q) for p, q in zip(a, b))
        w_u, w_f = wall(unfused), wall(fused)
        k_u, k_f = replay(unfused), replay(fused)
    rnd = lambda v: None if v is None else round(v, 2)
    legs["infer_step_propagate"] = {
        "shape": "B=1, 1 reference, 240x480 (thesis_dfpn/runner.py:153-159 after the network)", "bit_identical": bool(same),
        "unfused": {"launches": 4, "wall_us_per_call": rnd(w_u), "gpu_us_per_call_graph_replay": rnd(k_u),
                    "host_overhead_us": rnd(w_u - k_u) if k_u is not None else None},
        "fused": {"launches": 1, "wall_us_per_call": rnd(w_f), "gpu_us_per_call_graph_replay": rnd(k_f),
                  "host_overhead_us": rnd(w_f - k_f) if k_f is not None else None},
        "wall_reduction": round(1 - w_f / w_u, 3)}


# ------------------------------------------------------------------------------------------------ CPU baseline (torch CPU, oracle-checked)
def _cpu_corr(xt, vt, xr, vr):
    """utils/correlation.py:28-54 as stock torch ops (multi-threaded ATen): mask, L2-normalise over C, batched matmul."""
    b, c, f, h, w = xr.shape
    t = (xt * vt).reshape(b, c, h * w).transpose(1, 2)
    t = t / (t.norm(dim=2, keepdim=True) + 1e-9)
    r = (xr * vr).reshape(b, c, f, h * w).permute(0, 2, 1, 3)
    r = r / (r.norm(dim=2, keepdim=True) + 1e-9)
    return torch.matmul(t.unsqueeze(1), r).reshape(b, f, h, w, h, w)


def _cpu_align_set(x, v, flow):
    """utils/flow.py:93-101 as stock torch ops."""
    import torch.nn.functional as F
    b, c, f, h, w = x.shape
    grid = flow.reshape(b * f, h, w, 2)
    xa = F.grid_sample(x.transpose(1, 2).reshape(b * f, c, h, w), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    va = F.grid_sample(v.transpose(1, 2).reshape(b * f, 1, h, w), grid, mode="nearest", padding_mode="zeros", align_corners=True)
    return xa.reshape(b, f, c, h, w).transpose(1, 2), va.reshape(b, f, 1, h, w).transpose(1, 2)


def _cpu_affine_align(x, m, theta, r_list):
    """models/cpn.py:280-290 as stock torch ops."""
    import torch.nn.functional as F
    b, c, _, h, w = x.shape
    r = len(r_list)
    grid = F.affine_grid(theta, (b * r, c, h, w), align_corners=False)
    xa = F.grid_sample(x[:, :, r_list].transpose(1, 2).reshape(b * r, c, h, w), grid, align_corners=False)
    va = (F.grid_sample((1 - m[:, :, r_list]).transpose(1, 2).reshape(b * r, 1, h, w), grid, align_corners=False) > 0.5).float()
    return xa.reshape(b, r, c, h, w).transpose(1, 2), va.reshape(b, r, 1, h, w).transpose(1, 2)


def _cpu_context_matching(cf, v_t, v_al):
    """models/cpn.py:94-135 (normalize=False) vectorised over the references, stock torch ops; ref_importance = the bilinear
    upsample of the weights (the border drawing of add_match_border is left out)."""
    import torch.nn.functional as F
    b, c, f, h, w = cf.shape
    r = f - 1
    H, W = v_t.shape[-2:]
    vt = (F.interpolate(v_t, size=(h, w), mode="bilinear", align_corners=False) > 0.5).float()                      # (b,1,h,w)
    vr = (F.interpolate(v_al.reshape(b, r, H, W), size=(h, w), mode="bilinear", align_corners=False) > 0.5).float()  # (b,r,h,w)
    vmap = vt * vr
    v_sum = vmap.sum(dim=(2, 3))
    zeros = v_sum < 1e-4
    gs = (vmap.unsqueeze(1) * cf[:, :, :1] * cf[:, :, 1:]).sum(dim=(1, 3, 4))                                        # (b,r)
    gs = torch.where(zeros, torch.zeros_like(gs), gs)
    mv = gs[:, :, None, None] * vr                                                                                   # masked_softmax, dim = refs
    e = torch.exp(mv - mv.max(dim=1, keepdim=True)[0]) * vr
    ssum = e.sum(dim=1, keepdim=True)
    ssum = ssum + (ssum < 1e-4).float()
    c_match = e / ssum                                                                                                # (b,r,h,w)
    c_out = (cf[:, :, 1:] * c_match.unsqueeze(1)).sum(dim=2)
    c_mask = 1 - (c_match * vr).sum(dim=1, keepdim=True)
    imp = F.interpolate(c_match, size=(H, W), mode="bilinear", align_corners=False).unsqueeze(1).expand(b, 3, r, H, W).contiguous()
    return torch.cat([cf[:, :, 0], c_out, c_mask], dim=1), c_mask, imp


def cpu_baseline(budget_s=26.0):
    """The three legs as stock torch-CPU ops (ATen's multi-threaded matmul / grid_sample / interpolate kernels: what the
    reference's own CPU path executes), written here, forward + backward through torch.autograd, on bounded samples of the
    bench's shapes; every leg is first checked against the numpy oracle (the checker, not the thing timed).  Timed with all
    threads torch uses on this host and with one; composite pairs/s = 1 / sum of the legs' seconds per pair."""
    from oracle import tcl_oracle as O
    g = torch.Generator(device="cpu").manual_seed(1234)
    C, F_, S = CORR["C"], CORR["F"], CORR["S"]
    nb = 2
    xt = torch.randn(nb, C, S, S, generator=g, requires_grad=True)
    xr = torch.randn(nb, C, F_, S, S, generator=g, requires_grad=True)
    vt = (torch.rand(nb, 1, S, S, generator=g) >= 0.1).float()
    vr = (torch.rand(nb, 1, F_, S, S, generator=g) >= 0.1).float()
    cot = torch.randn(nb, F_, S, S, S, S, generator=g)

    def corr():
        out = _cpu_corr(xt, vt, xr, vr)
        torch.autograd.grad(out, [xt, xr], grad_outputs=cot)
        return nb * F_

    s = 256
    x = torch.rand(2, 3, 4, s, s, generator=g)
    v = (torch.rand(2, 1, 4, s, s, generator=g) < 0.85).float()
    fl_gt = (_ident(s)[None, None] * 1.05 + 0.1).repeat(2, 4, 1, 1, 1).contiguous()
    fl_pr = (_ident(s)[None, None] * 0.97 - 0.05).repeat(2, 4, 1, 1, 1).contiguous().requires_grad_(True)
    cw = torch.randn(2, 3, 4, s, s, generator=g)

    def warp():   # per pair at 256 px: the ground-truth-flow warp, the predicted-flow warp and its grid-gradient backward; 16 / 64 px add 7 %
        with torch.no_grad():
            _cpu_align_set(x, v, fl_gt)
        xa, _ = _cpu_align_set(x, v, fl_pr)
        torch.autograd.grad(xa, [fl_pr], grad_outputs=cw)
        return 8

    B, Cc, R, h, w, H, W = 1, C4["C"], C4["R"], C4["h"], C4["w"], C4["H"], C4["W"]
    r_list = [i for i in range(R + 1) if i != R // 2]
    x4 = torch.rand(B, 3, R + 1, H, W, generator=g)
    m4 = (torch.rand(B, 1, R + 1, H, W, generator=g) < 0.15).float()
    th = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + 0.05 * torch.randn(B * R, 2, 3, generator=g)).requires_grad_(True)
    cf = (0.05 * torch.randn(B, Cc, R + 1, h, w, generator=g)).requires_grad_(True)
    cx = torch.randn(B, 3, R, H, W, generator=g)
    cp = torch.randn(B, 2 * Cc + 1, h, w, generator=g)

    def match():
        xa, va = _cpu_affine_align(x4, m4, th, r_list)
        p_in, c_mask, imp = _cpu_context_matching(cf, 1 - m4[:, :, R // 2], va)
        torch.autograd.grad([xa, p_in], [th, cf], grad_outputs=[cx, cp])
        return R

    # ---- the oracle as the checker of what is about to be timed
    n = lambda t: t.detach().numpy()
    checks = {}
    out = _cpu_corr(xt[:1], vt[:1], xr[:1], vr[:1])
    checks["corr"] = bool(np.allclose(n(out), O.compute_masked_4d_correlation(n(xt[:1]), n(vt[:1]), n(xr[:1]), n(vr[:1])), rtol=1e-4, atol=1e-5))
    xa, va = _cpu_align_set(x[:1, :, :1], v[:1, :, :1], fl_pr[:1, :1])
    oxa, ova = O.align_set(n(x[:1, :, :1]), n(v[:1, :, :1]), n(fl_pr[:1, :1]))
    checks["warp"] = bool(np.allclose(n(xa), oxa, rtol=1e-4, atol=1e-5) and np.array_equal(n(va), ova))
    xa, va = _cpu_affine_align(x4, m4, th, r_list)
    oxa, ova, _ = O.cpn_align(n(x4), n(m4), None, n(th), r_list)
    p_in, c_mask, _ = _cpu_context_matching(cf, 1 - m4[:, :, R // 2], torch.from_numpy(ova))
    op, om, _ = O.context_matching(n(cf), n(1 - m4[:, :, R // 2]), ova, False, "default")
    # ATen's vectorised CPU grid_sample un-normalises as (g + 1) * (W / 2) - 0.5 (one fp32 ulp off the scalar / GPU formula near
    # x = 400: DESIGN.md section 2), hence 2e-4 absolute on the frames and a 1e-4 fraction of threshold flips on the visibility
    checks["match"] = bool(np.allclose(n(xa), oxa, rtol=1e-4, atol=2e-4) and np.mean(n(va) != ova) < 1e-4 and
                           np.allclose(n(p_in), op, rtol=1e-4, atol=1e-5) and np.allclose(n(c_mask), om, rtol=1e-4, atol=1e-5))

    def rate(fn, budget):
        fn()
        t0, cnt = time.perf_counter(), 0
        while True:
            cnt += fn()
            el = time.perf_counter() - t0
            if el > budget:
                return cnt / el, cnt, el

    all_threads = torch.get_num_threads()
    share = budget_s / 8.0
    res = {}
    for label, nthreads, scale in (("all", all_threads, 1.5), ("one", 1, 1.0)):
        torch.set_num_threads(nthreads)
        legs, sample = {}, []
        for name, fn, bsec in (("corr", corr, share * scale * 1.3), ("warp", warp, share * scale * 0.85), ("match", match, share * scale * 0.85)):
            r, cnt, el = rate(fn, bsec)
            legs[name] = round(r, 3)
            sample.append("%s: %d pairs in %.1f s" % (name, cnt, el))
        res[label] = dict(value=round(1.0 / sum(1.0 / legs[k] for k in legs), 3), legs_pairs_per_s=legs, cores=nthreads, sample="; ".join(sample))
    torch.set_num_threads(all_threads)
    a = res["all"]
    return dict(value=a["value"], unit="pairs/s", cores=a["cores"], kind="port", legs_pairs_per_s=a["legs_pairs_per_s"],
                cores_per_leg={k: a["cores"] for k in a["legs_pairs_per_s"]},
                sample="stock torch-CPU ops (ATen matmul / grid_sample / interpolate + autograd, written in bench.py), fwd+bwd per leg at the "
                       "bench's shapes (" + a["sample"] + "); composite = 1 / sum of seconds per pair; every leg checked against the numpy oracle first",
                oracle_check=checks, one_thread=dict(value=res["one"]["value"], legs_pairs_per_s=res["one"]["legs_pairs_per_s"], cores=1,
                                                     sample=res["one"]["sample"]))


# ------------------------------------------------------------------------------------------------ main
def launch_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run: start N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, one per GPU) BEFORE this process makes any GPU call, pass rank 0's JSON line through and
    return the worst exit code.  The parent never initialises the GPU and replaces no running program."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the cache-cold per-kernel measurements")
    ap.add_argument("--legs-only", default="", help="development: only the cache-cold measurements of these groups (corr,warp,c4), one line per leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))   # no launcher around us: start the ranks ourselves (nothing has touched the GPU yet)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # the line must never carry an n_gpus the caller did not ask for
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("TCL_BENCH_DRY") == "1":
        # launch-plumbing check for boxes without a GPU (tests/test_parallel.py): rendezvous over gloo, count the ranks, print a
        # line that carries no measurement
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n = torch.ones(1, dtype=torch.int64)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.all_reduce(n)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "dry_run": True, "n_gpus": world, "ranks_seen": int(n.item())}), flush=True)
        return
    # rehearsal switch (not used by the driver): TCL_BENCH_REHEARSE=1 runs every rank on GPU 0 with gloo for the barrier /
    # max-over-ranks, so the N > 1 code path can be exercised on a one-GPU box
    rehearse = os.environ.get("TCL_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from master_thesis_amd.models import cpn
    from master_thesis_amd.utils import correlation

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    if args.legs_only:
        groups = tuple(args.legs_only.split(","))
        legs = measure_legs(dev, make_corr(dev, g) if "corr" in groups else None, [make_warp_set(dev, g) for _ in range(N_WARP)] if "warp" in groups else None,
                            [make_c4_set(dev, g) for _ in range(N_C4)] if "c4" in groups else None, groups)
        for k, v in legs.items():
            if "ms_per_launch" not in v:
                print(k, json.dumps(v), flush=True)
                continue
            print("%-28s %8.4f ms  %8.1f %s  frac %.3f" % (k, v["ms_per_launch"], v["achieved"], v["unit"], v["frac"]), flush=True)
        return
    corr = make_corr(dev, g)
    warp_sets = [make_warp_set(dev, g) for _ in range(N_WARP)]
    c4_sets = [make_c4_set(dev, g) for _ in range(N_C4)]
    module = cpn.CPNContextMatchingDefault()
    stream = torch.cuda.current_stream(dev)
    marks = []   # (e0, e1, e2, e3) per pass: leg boundaries on the launch stream

    def one_pass(record):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if record else None
        if record:
            ev[0].record(stream)
        for _ in range(N_CORR):
            corr_leg(corr)
        if record:
            ev[1].record(stream)
        for d in warp_sets:
            warp_leg(d)
        if record:
            ev[2].record(stream)
        for d in c4_sets:
            c4_leg(d, module)
        if record:
            ev[3].record(stream)
            marks.append(ev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; the last warm-up pass also sizes `inner` so that the timed region lasts >= 1 s whatever --steps is
    one_pass(False)
    barrier()
    t0 = time.perf_counter()
    one_pass(False)
    barrier()
    t_pass = time.perf_counter() - t0
    inner = max(1, int(math.ceil(1.0 / (max(args.steps, 1) * t_pass))))
    if world > 1:
        ti = torch.tensor([inner], device="cpu" if rehearse else dev, dtype=torch.int64)
        dist.all_reduce(ti, op=dist.ReduceOp.MAX)
        inner = int(ti.item())
    for _ in range(max(0, args.warmup - 2)):
        for _ in range(inner):
            one_pass(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(inner):
            one_pass(True)
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    n_pass = args.steps * inner
    leg_ms = {k: sum(ev[i].elapsed_time(ev[i + 1]) for ev in marks) / len(marks) for i, k in enumerate(("corr", "warp", "match"))}

    # the other arithmetic variants of the correlation leg, this rank only (reported beside the headline, never as `value`):
    # bf16x3 = hi/lo split on bf16 MFMA, meets the fp32 bar; bf16 = plain bf16 operands, 1e-2 bar
    variants = {}
    if not args.no_variants and rank == 0:
        for prec in ("bf16x3", "bf16"):
            def vstep():
                out = correlation.compute_masked_4d_correlation(corr["xt"], corr["vt"], corr["xr"], corr["vr"], precision=prec)
                return torch.autograd.grad(out, [corr["xt"], corr["xr"]], grad_outputs=corr["cot"])
            for _ in range(2):
                vstep()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(10):
                vstep()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - tv) / 10 * 1e3
            variants[prec] = {"corr_leg_ms_per_16_pairs": round(ms, 4), "tolerance": "fp32 bar (1e-4)" if prec == "bf16x3" else "1e-2"}
    result = None
    if rank == 0:
        legs = None if args.no_legs else measure_legs(dev, corr, warp_sets, c4_sets)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("gemm_f32_fwd_bytes_per_launch")
        result = {
            "metric": METRIC,
            "value": round(PAIRS_PER_PASS * n_pass * world / el, 2),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2 + C3 + C4 legs of one pass: corr 64x64 C=256 (40 x 16 pairs) + align_set x6 at 16/64/256 px with grid-gradient "
                                   "backward (5 x 128 pairs) + CPN affine warp and context matching at 240x424, 10 references (2 x 320 pairs), fwd+bwd",
                       "pairs_per_pass_per_gpu": PAIRS_PER_PASS, "passes_per_step": inner, "timed_region_s": round(el, 3),
                       "precision": "fp32 (exact fp32 MFMA for the contraction)", "sharding": "independent pairs per rank, no collective"},
            "leg_ms_per_pass": {k: round(v, 4) for k, v in leg_ms.items()},
        }
        if legs:
            k = legs["corr_fwd_contraction"]
            result["roofline"] = {"bound": "mfma", "kernel": k["kernel"], "achieved": k["achieved"], "peak": k["peak"], "unit": k["unit"],
                                  "frac": k["frac"], "ms_per_launch": k["ms_per_launch"], "traffic": traffic,
                                  "share_of_step": round(N_CORR * k["ms_per_launch"] / sum(leg_ms.values()), 3)}
            result["legs"] = legs
        if variants:
            result["variants"] = variants
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
