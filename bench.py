#!/usr/bin/env python3
"""Headline benchmark: aligned frame-pairs/sec on BASELINE.json configs[1]
("utils.correlation all-pairs cost volume alone, 64x64 feature maps C=256, batch 16, fwd+bwd on 1 MI355X").

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one forward + backward of compute_masked_4d_correlation over this rank's 16 synthetic frame pairs
(inputs resident in HBM), through the drop-in function and the C ABI.  Ranks work on independent pairs: no
data-path collective, weak scaling.  Rank 0 prints ONE JSON line with the whole-job pairs/s, the roofline of the
dominant kernel (timed live with HIP events on the launch stream) and the CPU baseline (the numpy oracle on the
host cores, bounded sample).
"""
import argparse
import json
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

WORKLOADS = {
    # name: (pairs per GPU, C, S)
    "c2": dict(B=16, C=256, F=1, S=64, label="C2: utils.correlation all-pairs cost volume, 64x64 maps, C=256, batch 16, fwd+bwd"),
}


def make_inputs(w, dev, rank):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    B, C, F, S = w["B"], w["C"], w["F"], w["S"]
    xt = torch.randn(B, C, S, S, generator=g)
    xr = torch.randn(B, C, F, S, S, generator=g)
    vt = (torch.rand(B, 1, S, S, generator=g) >= 0.1).float()
    vr = (torch.rand(B, 1, F, S, S, generator=g) >= 0.1).float()
    cot = torch.randn(B, F, S, S, S, S, generator=g)
    return [t.to(dev) for t in (xt, vt, xr, vr, cot)]


def time_dominant_kernel(w, dev, iters=20):
    """Average duration of the forward contraction kernel (gemm_f32_kernel<128,128,...>) alone, HIP events on the launch stream."""
    import ctypes
    from master_thesis_amd import _lib
    B, C, F, S = w["B"], w["C"], w["F"], w["S"]
    P = S * S
    xt, vt, xr, vr, _ = make_inputs(w, dev, 0)
    lib = _lib.lib()
    ws = torch.empty(lib.tcl_corr4d_fwd_workspace_bytes(B, C, F, P, 0), dtype=torch.uint8, device=dev)
    out = torch.empty(B, F, P, P, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.call("tcl_corr4d_fwd", p(xt), p(vt.reshape(B, P)), p(xr), p(vr.reshape(B, F, P)), p(out), B, C, F, P, 0, p(ws), ws.numel(), sp)
    for _ in range(3):
        _lib.call("tcl_corr4d_contract", p(ws), p(out), B, C, F, P, 0, sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        _lib.call("tcl_corr4d_contract", p(ws), p(out), B, C, F, P, 0, sp)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * P * P * C * B * F            # 8.59 GFLOP per pair forward
    del out, ws
    return ms, flops


def cpu_baseline(w, budget_s=12.0):
    """The numpy oracle (fwd + bwd of A1) on the host cores, on a bounded sample of the same workload."""
    from oracle import tcl_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    C, F, S = w["C"], w["F"], w["S"]
    nb = 2
    rng = np.random.default_rng(1234)
    xt = rng.standard_normal((nb, C, S, S)).astype(np.float32)
    xr = rng.standard_normal((nb, C, F, S, S)).astype(np.float32)
    vt = (rng.random((nb, 1, S, S)) >= 0.1).astype(np.float32)
    vr = (rng.random((nb, 1, F, S, S)) >= 0.1).astype(np.float32)
    cot = rng.standard_normal((nb, F, S, S, S, S)).astype(np.float32)
    O.compute_masked_4d_correlation(xt[:1], vt[:1], xr[:1], vr[:1])  # warm-up (BLAS threads)
    t0 = time.perf_counter()
    pairs = 0
    while True:
        O.compute_masked_4d_correlation(xt, vt, xr, vr)
        O.compute_masked_4d_correlation_bwd(xt, vt, xr, vr, cot)
        pairs += nb * F
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return dict(value=round(pairs / el, 3), unit="pairs/s", cores=int(cores), kind="port",
                sample="%d pairs of the C2 workload (fwd+bwd, numpy oracle, %.1f s)" % (pairs, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    from master_thesis_amd.utils import correlation

    w = WORKLOADS[args.workload]
    xt, vt, xr, vr, cot = make_inputs(w, dev, rank)
    xt.requires_grad_(True)
    xr.requires_grad_(True)

    def step():
        out = correlation.compute_masked_4d_correlation(xt, vt, xr, vr)
        g = torch.autograd.grad(out, [xt, xr], grad_outputs=cot)
        return g

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    pairs_per_step = w["B"] * w["F"] * world

    # the other arithmetic variants of the same path, same workload, this rank only (reported beside the headline,
    # never as `value`): bf16x3 = hi/lo split on bf16 MFMA, meets the fp32 bar; bf16 = plain bf16 operands, 1e-2 bar
    variants = {}
    if not args.no_variants:
        for prec in ("bf16x3", "bf16"):
            def vstep():
                out = correlation.compute_masked_4d_correlation(xt, vt, xr, vr, precision=prec)
                return torch.autograd.grad(out, [xt, xr], grad_outputs=cot)
            for _ in range(2):
                vstep()
            torch.cuda.synchronize()
            tv = time.perf_counter()
            for _ in range(10):
                vstep()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - tv) / 10 * 1e3
            variants[prec] = {"ms_per_step": round(ms, 4), "pairs_per_s_per_gpu": round(w["B"] * w["F"] / ms * 1e3, 1),
                              "tolerance": "fp32 bar (1e-4)" if prec == "bf16x3" else "1e-2"}
    result = None
    if rank == 0:
        kms, kflops = time_dominant_kernel(w, dev)
        achieved = kflops / (kms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("gemm_f32_fwd_bytes_per_launch")
        result = {
            "metric": "aligned frame-pairs/sec (corr fwd+bwd)",
            "value": round(pairs_per_step * args.steps / el, 2),
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
            "config": {"workload": w["label"], "pairs_per_gpu": w["B"] * w["F"], "C": w["C"], "map": "%dx%d" % (w["S"], w["S"]),
                       "precision": "fp32 MFMA (exact)", "sharding": "independent pairs per rank, no collective"},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel<128,128,2,2> (tcl_corr4d_contract, forward contraction)",
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4), "ms_per_launch": round(kms, 4), "traffic": traffic},
        }
        if variants:
            result["variants"] = variants
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(w)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
