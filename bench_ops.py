#!/usr/bin/env python3
"""Per-op roofline report for the bandwidth-bound rows of SURVEY.md §8 (C3 warp, C4 affine warp + context matching)
and the other correlation variants.  Not the headline bench (that is bench.py); prints one JSON line per op with the
achieved algorithmic GB/s (or TFLOP/s) against the MI355X peak, timed with HIP events on the launch stream.

    python bench_ops.py [--iters 20] [--ops warp,affine,ctx,corr_model,argmax,cpu]      (cpu = the numpy port on the host cores)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0
PEAK_F32_TF = 157.3
DEV = "cuda:0"


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, bytes_=None, flops=None, units=None, unit_name="pairs", note=""):
    r = {"op": name, "ms": round(ms, 4)}
    if bytes_ is not None:
        gbs = bytes_ / (ms * 1e-3) / 1e9
        r.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), algorithmic_bytes=int(bytes_))
    if flops is not None:
        tf = flops / (ms * 1e-3) / 1e12
        r.update(bound="mfma", achieved=round(tf, 2), peak=PEAK_F32_TF, unit="TFLOP/s", frac=round(tf / PEAK_F32_TF, 4))
    if units:
        r["%s_per_s" % unit_name] = round(units / (ms * 1e-3), 1)
    if note:
        r["note"] = note
    print(json.dumps(r), flush=True)


def bench_warp(iters):
    """C3: align_set at 256^2, 32 sequences x 4 reference frames (x 3ch bilinear + v nearest, one grid read)."""
    from master_thesis_amd.utils import flow as F
    B, R, H, W = 32, 4, 256, 256
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, R, H, W, generator=g).to(DEV)
    v = (torch.rand(B, 1, R, H, W, generator=g) < 0.85).float().to(DEV)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    ident = torch.stack([xs, ys], -1)
    theta = torch.randn(B, R, 1, 1, 2, generator=g) * 0.15
    flow = (ident[None, None] * (1 + 0.1 * torch.randn(B, R, 1, 1, 1, generator=g)) + theta).to(DEV)
    npix = B * R * H * W
    ms = timed(lambda: F.align_set(x, v, flow), iters)
    report("align_set fwd 256^2 (C3)", ms, bytes_=npix * (4 * 4 + 8 + 4 * 4), units=B * R, note="read x(3)+v(1)+grid(2), write x(3)+v(1)")
    flow_g = flow.clone().requires_grad_(True)
    xa, _ = F.align_set(x, v, flow_g)
    cot = torch.randn_like(xa)
    ms = timed(lambda: torch.autograd.grad(xa, flow_g, cot, retain_graph=True), iters)
    report("align_set bwd grad-grid only 256^2 (what the reference exercises)", ms, bytes_=npix * (3 * 4 + 3 * 4 + 8 + 8), units=B * R,
           note="read grad_out(3)+x(3)+grid(2), write grad_grid(2)")
    x_g = x.clone().requires_grad_(True)
    xa2, _ = F.align_set(x_g, v, flow_g)
    ms = timed(lambda: torch.autograd.grad(xa2, [x_g, flow_g], cot, retain_graph=True), iters)
    report("align_set bwd grad-grid + grad-input (atomics) 256^2", ms, bytes_=npix * (3 * 4 + 3 * 4 + 8 + 8 + 3 * 4), units=B * R,
           note="incl. memset of grad_x; atomics bound ~1.3 TB/s of added bytes")


def bench_affine(iters):
    """C4 warp: CPN affine alignment at 240x424, 32 targets x 10 references, theta -> coordinates on the fly."""
    from master_thesis_amd.models import cpn
    B, R, H, W = 32, 10, 240, 424
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, R + 1, H, W, generator=g).to(DEV)
    m = (torch.rand(B, 1, R + 1, H, W, generator=g) < 0.15).float().to(DEV)
    r_list = [i for i in range(R + 1) if i != R // 2]
    theta = (torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]) + 0.05 * torch.randn(B * R, 2, 3, generator=g)).to(DEV)
    xr, vr = x[:, :, r_list].contiguous(), (1 - m[:, :, r_list]).contiguous()
    from master_thesis_amd import _lib, _ops
    npix = B * R * H * W
    ms = timed(lambda: _ops.WarpFn.apply(xr, vr, None, theta, False, _lib.VMODE_BILINEAR_THRESHOLD), iters)
    report("affine warp fwd 240x424 (C4)", ms, bytes_=npix * (4 * 4 + 4 * 4), units=B * R, note="read x(3)+v(1), write x(3)+v(1); no grid in HBM")


def bench_ctx(iters):
    """C4 context matching: c_feats (32,128,11,30,53), masks 240x424."""
    from master_thesis_amd.models import cpn
    B, C, R, h, w, H, W = 32, 128, 10, 30, 53, 240, 424
    g = torch.Generator().manual_seed(2)
    cf = (0.05 * torch.randn(B, C, R + 1, h, w, generator=g)).to(DEV)
    v_t = (torch.rand(B, 1, H // 8, W // 8, generator=g) < 0.6).float().repeat_interleave(8, 2).repeat_interleave(8, 3).to(DEV)
    v_al = (torch.rand(B, 1, R, H // 8, W // 8, generator=g) < 0.7).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(DEV)
    mod = cpn.CPNContextMatchingDefault()
    from master_thesis_amd import _lib, _ops
    hw = h * w
    fwd_bytes = B * (C * (R + 1) * hw * 4 + (R + 1) * H * W * 4 + (2 * C + 1) * hw * 4 + hw * 4)
    ms = timed(lambda: _ops.ContextMatchingFn.apply(cf, v_t, v_al, False, _lib.CTX_DEFAULT), iters)
    report("context matching fwd (C4, no ref_importance)", ms, bytes_=fwd_bytes, units=B * R, note="one read of c_feats + masks, one write of p_in")
    ms = timed(lambda: mod(cf, v_t, v_al, normalize=False), iters)
    report("context matching fwd + ref_importance (C4)", ms, bytes_=fwd_bytes + B * 3 * (R + 1) * H * W * 4, units=B * R)
    cfg = cf.clone().requires_grad_(True)
    p_in, c_mask, _ = _ops.ContextMatchingFn.apply(cfg, v_t, v_al, False, _lib.CTX_DEFAULT)
    cot = torch.randn_like(p_in)
    ms = timed(lambda: torch.autograd.grad(p_in, cfg, cot, retain_graph=True), iters)
    report("context matching bwd (C4)", ms, bytes_=B * (2 * C * (R + 1) * hw * 4 + (2 * C + 1) * hw * 4), units=B * R,
           note="read c_feats + grad p_in, write grad c_feats")


def bench_corr_model(iters):
    """C1/C3 correlation shape: 16x16 maps, C=512, 32 sequences x 4 refs = 128 pairs."""
    from master_thesis_amd.utils import correlation
    B, C, F, S = 32, 512, 4, 16
    g = torch.Generator().manual_seed(3)
    xt = torch.randn(B, C, S, S, generator=g).to(DEV)
    xr = torch.randn(B, C, F, S, S, generator=g).to(DEV)
    vt = (torch.rand(B, 1, S, S, generator=g) < 0.9).float().to(DEV)
    vr = (torch.rand(B, 1, F, S, S, generator=g) < 0.9).float().to(DEV)
    ms = timed(lambda: correlation.compute_masked_4d_correlation(xt, vt, xr, vr), iters)
    report("corr4d fwd 16x16 C=512 x128 pairs (model shape)", ms, flops=2.0 * 256 * 256 * C * B * F, units=B * F, note="launch-latency regime")


def bench_argmax(iters):
    """N1: fused correlation + argmax flow at the C2 shape (volume never stored)."""
    from master_thesis_amd import _ops
    B, C, F, S = 16, 256, 1, 64
    g = torch.Generator().manual_seed(4)
    xt = torch.randn(B, C, S, S, generator=g).to(DEV)
    xr = torch.randn(B, C, F, S, S, generator=g).to(DEV)
    ms = timed(lambda: _ops.corr4d_argmax_flow(xt, None, xr, None), iters)
    report("corr4d + corr_to_flow fused 64x64 C=256 x16 pairs", ms, flops=2.0 * 4096 * 4096 * C * B * F, units=B * F)


def bench_cpu(iters):
    """The CPU restatement of the reference (numpy oracle, `kind: port`) on the host cores, on bounded samples of the same
    workloads: pairs/s beside the GPU rows above (a reported baseline, not a target)."""
    import time
    import numpy as np
    from oracle import tcl_oracle as O
    rng = np.random.default_rng(0)
    def run(name, fn, units, budget=6.0):
        fn()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            fn(); n += 1
        el = time.perf_counter() - t0
        print(json.dumps({"op": name, "cpu_baseline": {"value": round(units * n / el, 2), "unit": "pairs/s", "kind": "port",
                                                       "cores": os.cpu_count() if "BLAS" in name else 1,
                                                       "sample": "%d calls of %d pairs in %.1f s" % (n, units, el)}}), flush=True)
    B, R, H, W = 1, 4, 256, 256
    x = rng.random((B, 3, R, H, W)).astype(np.float32)
    v = (rng.random((B, 1, R, H, W)) < 0.85).astype(np.float32)
    ident = np.stack(np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H)), axis=-1).astype(np.float32)
    flow = (ident[None, None] * 1.05 + 0.1).astype(np.float32).repeat(R, axis=1)
    run("CPU port: align_set fwd 256^2 (C3)", lambda: O.align_set(x, v, flow), B * R)
    cot = rng.standard_normal(x.shape).astype(np.float32)
    run("CPU port: align_set bwd 256^2 (C3)", lambda: O.align_set_bwd(x, v, flow, cot), B * R)
    C, R, h, w, H, W = 128, 10, 30, 53, 240, 424
    cf = (0.05 * rng.standard_normal((1, C, R + 1, h, w))).astype(np.float32)
    v_t = (rng.random((1, 1, H, W)) < 0.6).astype(np.float32)
    v_al = (rng.random((1, 1, R, H, W)) < 0.7).astype(np.float32)
    run("CPU port: context matching fwd (C4)", lambda: O.context_matching(cf, v_t, v_al, False, "default"), R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ops", default="warp,affine,ctx,corr_model,argmax")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    table = dict(warp=bench_warp, affine=bench_affine, ctx=bench_ctx, corr_model=bench_corr_model, argmax=bench_argmax, cpu=bench_cpu)
    for op in a.ops.split(","):
        table[op](a.iters)


if __name__ == "__main__":
    main()
