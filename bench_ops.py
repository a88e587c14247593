#!/usr/bin/env python3
"""Per-op roofline report for the bandwidth-bound rows of SURVEY.md §8 (C3 warp, C4 affine warp + context matching)
and the other correlation variants.  Not the headline bench (that is bench.py); prints one JSON line per op with the
achieved algorithmic GB/s (or TFLOP/s) against the MI355X peak, timed with HIP events on the launch stream.

    python bench_ops.py [--iters 20] [--ops warp,affine,ctx,corr_model,argmax,movement,sched,misc,cpu]      (cpu = the numpy port on the host cores)
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


COLD = False      # --cold: evict L2 and the 256 MB Infinity Cache before every timed call
_flush = None


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s = torch.cuda.current_stream()
    if COLD:
        # One event pair per call, a read of 1 GiB in front of each: the working sets of C3 / C4 (0.3 - 1 GB) would
        # otherwise be partly served from the Infinity Cache between iterations.  The flush takes ~0.4 ms on the GPU, so the
        # host enqueues the next call's kernels while it runs and no launch gap lands inside the event pair.
        global _flush
        if _flush is None:
            _flush = torch.zeros(1 << 28, dtype=torch.float32, device=DEV)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for e0, e1 in pairs:
            _flush.sum()   # read-only on purpose: a writing flush leaves 256 MB of dirty lines whose write-back the timed kernel then pays
            e0.record(s)
            fn()
            e1.record(s)
        pairs[-1][1].synchronize()
        return sum(e0.elapsed_time(e1) for e0, e1 in pairs) / iters
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, bytes_=None, flops=None, units=None, unit_name="pairs", note="", peak_tf=None):
    r = {"op": name, "ms": round(ms, 4)}
    if bytes_ is not None:
        gbs = bytes_ / (ms * 1e-3) / 1e9
        r.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), algorithmic_bytes=int(bytes_))
    if flops is not None:
        tf = flops / (ms * 1e-3) / 1e12
        pk = PEAK_F32_TF if peak_tf is None else peak_tf
        r.update(bound="mfma", achieved=round(tf, 2), peak=pk, unit="TFLOP/s", frac=round(tf / pk, 4))
    if units:
        r["%s_per_s" % unit_name] = round(units / (ms * 1e-3), 1)
    if note:
        r["note"] = note
    if COLD:
        r["cache"] = "cold (1 GiB read before every call)"
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
    # backward through the C ABI directly (an autograd.grad call costs more host time than these kernels run)
    import ctypes
    from master_thesis_amd import _lib
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    xa, _ = F.align_set(x, v, flow)
    cot = torch.randn_like(xa)          # (B,3,R,H,W) view of a (B,R,3,H,W) buffer, like the forward's output
    g_grid, g_x = torch.empty_like(flow), torch.zeros_like(x)
    ws = torch.empty(256, dtype=torch.uint8, device=DEV)
    def bwd(with_gx):
        if with_gx:
            g_x.zero_()
        _lib.call("tcl_warp_bwd", P(cot), cot.stride(0), cot.stride(2), cot.stride(1), P(x), x.stride(0), x.stride(2), x.stride(1),
                  P(flow), None, P(g_x) if with_gx else None, g_x.stride(0), g_x.stride(2), g_x.stride(1), P(g_grid), None,
                  B, R, 3, H, W, 1, P(ws), ws.numel(), None)
    ms = timed(lambda: bwd(False), iters)
    report("align_set bwd grad-grid only 256^2 (what the reference exercises)", ms, bytes_=npix * (3 * 4 + 3 * 4 + 8 + 8), units=B * R,
           note="read grad_out(3)+x(3)+grid(2), write grad_grid(2)")
    ms = timed(lambda: bwd(True), iters)
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
    from master_thesis_amd import _lib
    for name, prec in (("f32", _lib.PREC_F32), ("bf16x3", _lib.PREC_BF16X3), ("bf16", _lib.PREC_BF16)):
        ms = timed(lambda: _ops.corr4d_argmax_flow(xt, None, xr, None, precision=prec), iters)
        nm = 3 if name == "bf16x3" else 1
        report("corr4d + corr_to_flow fused 64x64 C=256 x16 pairs (%s)" % name, ms, flops=2.0 * 4096 * 4096 * C * B * F * nm, units=B * F,
               peak_tf=None if name == "f32" else 2500.0,
               note="whole entry point incl. both normalisations; MFMA flops issued (%dx the algorithmic) against the pipe's dense peak" % nm)


def bench_movement(iters):
    """N4: MovementSimulator.simulate_movement_batch, 64 frames -> 64 x 5-frame sequences at 256^2, beside the stock
    torch formulation the reference uses (repeat + affine_grid + grid_sample)."""
    import numpy as np
    import torch.nn.functional as F
    from master_thesis_amd.utils.movement import MovementSimulator
    B, n, H, W = 64, 5, 256, 256
    sim = MovementSimulator(20, 0.1, 0.2)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(6)).to(DEV)
    np.random.seed(0)
    _, _, mats = sim.simulate_movement_batch(x, n)
    ms = timed(lambda: sim.simulate_movement_batch(x, n, mats), iters)
    npix = B * n * H * W
    report("simulate_movement_batch 64x5 256^2 (N4), incl. host 3x3 chain", ms, bytes_=B * 3 * H * W * 4 + npix * (3 * 4 + 8), units=B * n, unit_name="frames",
           note="read x once per sequence, write y(3) + flow_inv(2); the rest of the wall time is the host 3x3 chain + 2 H2D copies")
    from master_thesis_amd import _lib, _ops
    theta = torch.eye(2, 3).repeat(B * n, 1, 1).to(DEV) + 0.05 * torch.randn(B * n, 2, 3, generator=torch.Generator().manual_seed(7)).to(DEV)
    xe = x.unsqueeze(2).expand(B, 3, n, H, W)
    ms = timed(lambda: (_ops.WarpFn.apply(xe, None, None, theta, True, _lib.VMODE_NONE), _ops.AffineGridFn.apply(theta, H, W, True)), iters)
    report("  device part only: warp(theta, frame stride 0) + affine_grid", ms, bytes_=B * 3 * H * W * 4 + npix * (3 * 4 + 8), units=B * n, unit_name="frames")
    def stock():
        flow = F.affine_grid(theta, [B * n, 3, H, W], align_corners=True)
        F.affine_grid(theta, [B * n, 3, H, W], align_corners=True)
        return F.grid_sample(x.unsqueeze(1).repeat(1, n, 1, 1, 1).reshape(B * n, 3, H, W), flow, align_corners=True)
    ms = timed(stock, iters)
    report("  stock ATen formulation of the same (repeat + 2x affine_grid + grid_sample)", ms, bytes_=B * 3 * H * W * 4 + npix * (3 * 4 + 8), units=B * n, unit_name="frames")


def bench_sched(iters):
    """N3: the copy-and-propagate scheduler on 32 sequences of 8 frames at 256^2, batched in lock-step vs the reference's
    one-sequence-at-a-time order (same HIP ops, same weight-free stand-in networks from tests/standins.py)."""
    import time
    import numpy as np
    from master_thesis_amd import schedulers as S
    from tests import standins
    n_seq, f, H, W = 32, 8, 256, 256
    seqs = [standins.moving_hole_sequence(i, 3, f, H, W, 40 + 4 * (i % 8), 6 + i % 5) for i in range(n_seq)]
    x = torch.from_numpy(np.stack([s[0] for s in seqs])).to(DEV)
    m = torch.from_numpy(np.stack([s[1] for s in seqs])).to(DEV)
    pairs = [0]
    step = S.runners.ThesisInpaintingRunner.infer_step_propagate
    def counted(*a):
        pairs[0] += a[2].size(0)
        return step(*a)
    def run(batched):
        S.runners.ThesisInpaintingRunner.infer_step_propagate = staticmethod(counted)
        try:
            pairs[0] = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if batched:
                y = S.inpainting_algorithm_cp_batched(x, m, standins.torch_flow, standins.torch_inpaint, N=4, s=1, e=1)
            else:
                y = torch.stack([S.inpainting_algorithm_cp(x[i], m[i], standins.torch_flow, standins.torch_inpaint, N=4, s=1, e=1) for i in range(n_seq)])
            torch.cuda.synchronize()
            return y, time.perf_counter() - t0, pairs[0]
        finally:
            S.runners.ThesisInpaintingRunner.infer_step_propagate = staticmethod(step)
    run(True)
    yb, tb, pb = run(True)
    ys, ts, ps = run(False)
    assert pb == ps and torch.allclose(yb, ys, atol=1e-6)
    print(json.dumps({"op": "inpainting_algorithm_cp, 32 sequences x 8 frames 256^2, N=4 (N3)", "pairs": pb,
                      "batched_ms": round(tb * 1e3, 2), "batched_pairs_per_s": round(pb / tb, 1),
                      "sequential_ms": round(ts * 1e3, 2), "sequential_pairs_per_s": round(ps / ts, 1),
                      "speedup": round(ts / tb, 2), "note": "identical outputs; stand-in networks, so the time is the schedulers' own ops + launches"}), flush=True)


def bench_misc(iters):
    """A5 standalone masked softmax at the C4 shape and A2 per-pixel correlation (both are utilities off the hot loop:
    the context-matching kernels fuse A5, nothing in the reference calls A2)."""
    from master_thesis_amd.models import cpn
    from master_thesis_amd.utils import correlation
    g = torch.Generator().manual_seed(8)
    B, C, R, h, w = 32, 128, 10, 30, 53
    vec = torch.randn(B, C, R, h, w, generator=g).to(DEV)
    mask = (torch.rand(B, 1, R, h, w, generator=g) < 0.7).float().to(DEV)
    ms = timed(lambda: cpn.masked_softmax(vec, mask, 2), iters)
    report("masked_softmax fwd (B,128,10,30,53) (A5 standalone)", ms, bytes_=vec.numel() * 8 + mask.numel() * 4, units=B, unit_name="targets")
    B, C, H, W = 32, 256, 64, 64
    f1, f2 = torch.randn(B, C, H, W, generator=g).to(DEV), torch.randn(B, C, H, W, generator=g).to(DEV)
    m1, m2 = (torch.rand(B, 1, H, W, generator=g) < 0.9).float().to(DEV), (torch.rand(B, 1, H, W, generator=g) < 0.9).float().to(DEV)
    ms = timed(lambda: correlation.compute_masked_correlation(f1, f2, m1, m2), iters)
    report("compute_masked_correlation fwd (32,256,64,64) (A2)", ms, bytes_=2 * f1.numel() * 4 + 3 * B * H * W * 4, units=B)


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
    ap.add_argument("--cold", action="store_true", help="flush L2 / Infinity Cache before every timed call")
    a = ap.parse_args()
    global COLD
    COLD = a.cold
    torch.cuda.set_device(0)
    table = dict(misc=bench_misc, movement=bench_movement, sched=bench_sched, warp=bench_warp, affine=bench_affine, ctx=bench_ctx, corr_model=bench_corr_model, argmax=bench_argmax, cpu=bench_cpu)
    for op in a.ops.split(","):
        table[op](a.iters)


if __name__ == "__main__":
    main()
