#!/usr/bin/env python3
"""profiles/traffic.json from the raw PMC passes (round 3: r3_legs_hbm_bytes_raw.json; round 2: r2_legs_hbm_bytes_raw.json; round 1:
r1_c2_hbm_bytes_raw.json, r1_ops_hbm_bytes_raw.json).

The raw files are what `profiles/microbench/pmc_bytes.sh` writes: per kernel, the average FETCH_SIZE and WRITE_SIZE (KiB) of two separate
`rocprofv3 --pmc` passes over `bench.py --legs-only corr,warp,c4` (rounds 2 and 3: the cache-cold, rotating-buffer launches of every leg;
round 1 used `bench.py --steps 1 ...` and `bench_ops.py --iters 2`).  hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (a 128-byte request is counted as 64 bytes; calibrated there for 16-byte-per-lane
streaming loads, so for the 4- and 8-byte-per-lane loads of the streaming kernels it is an upper estimate).
Algorithmic bytes are the figures of DESIGN.md section 4 for the benchmarked shapes.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MB = 1 << 20
C2_PAIR = 4096 * 4096 * 4                      # one pair's volume, bytes
HAT = 256 * 4096 * 4                           # one normalised operand of one pair
NPIX_C3, NPIX_C4 = 128 * 256 * 256, 320 * 240 * 424
CF = 32 * 128 * 11 * 30 * 53 * 4               # c_feats at C4
ROWS = [  # (label, raw file, kernel-name substring, algorithmic bytes per launch)
    ("gemm_f32_fwd (C2, 16 pairs)", "legs", "gemm_f32_kernel<128, 128, 2, 2, false, false, true, 0", 16 * (C2_PAIR + 2 * HAT)),
    ("gemm_bf16x3_fwd (C2; bf16 hi + lo operand images)", "legs", "gemm_bf16op_kernel<128, 128, 2, 2, 2, 0", 16 * (C2_PAIR + 2 * HAT)),
    ("gemm_bf16_fwd (C2; bf16 operand images)", "legs", "gemm_bf16op_kernel<128, 128, 2, 2, 1, 0", 16 * (C2_PAIR + HAT)),
    ("gemm_bf16_fused_argmax (C2; no volume)", "legs", "gemm_bf16op_kernel<128, 128, 2, 2, 1, 1", 16 * HAT + 16 * 4096 * 32 * 8),
    ("gemm_f32_wide bwd_d_a (C2)", "legs", "gemm_f32_wide_kernel<true, true>", 16 * (C2_PAIR + 2 * HAT)),
    ("gemm_f32_wide bwd_d_r (C2)", "legs", "gemm_f32_wide_kernel<true, false>", 16 * (C2_PAIR + 2 * HAT)),
    ("corr_normalise (one of two launches; fp32 + bf16 images in the mix)", "legs", "corr_normalise_keep_kernel", 16 * 2 * HAT),
    ("corr_normalise_bwd", "legs", "corr_normalise_bwd_keep_kernel", 16 * 3 * HAT),
    ("warp_fwd align_set 256^2 x128 (C3)", "legs", "warp_fwd_kernel<false, true, 1, true", NPIX_C3 * 40),
    ("warp_bwd grad-grid 256^2 x128 (C3)", "legs", "warp_bwd_kernel<false, false, true, true>", NPIX_C3 * 40),
    ("warp_bwd grad-grid + grad-input (tiled)", "legs", "warp_bwd_tile_kernel<false, true, true>", NPIX_C3 * 52),
    ("warp_fwd affine 240x424 x320 (C4)", "legs", "warp_fwd_kernel<true, true, 2, true", NPIX_C4 * 32),
    ("warp_bwd theta gradient (C4, tiled)", "legs", "warp_bwd_tile_kernel<true, false, true>", NPIX_C4 * 24),
    ("ctx_sim (C4, 32 targets)", "legs", "ctx_sim_kernel", CF),
    ("ctx_aggregate (C4)", "legs", "ctx_aggregate_kernel", CF + 32 * 257 * 30 * 53 * 4),
    ("ctx_bwd_dmatch (C4)", "legs", "ctx_bwd_dmatch_kernel", CF * 10 // 11 + 2 * 32 * 128 * 30 * 53 * 4),
    ("ctx_bwd_feats (C4)", "legs", "ctx_bwd_feats_", 2 * CF + 32 * 257 * 30 * 53 * 4),
    ("ref_importance (C4)", "legs", "ref_importance_kernel", 32 * 3 * 11 * 240 * 424 * 4),
]


def main():
    raw = {"legs": json.load(open(os.path.join(HERE, "r3_legs_hbm_bytes_raw.json")))}
    out = {"_method": __doc__.strip()}
    for label, src, needle, algo in ROWS:
        hit = [v for k, v in raw[src].items() if needle in k]
        if not hit:
            continue
        f, w = hit[0]["FETCH_SIZE_kib_avg"], hit[0]["WRITE_SIZE_kib_avg"]
        hbm = int((2 * f + w) * 1024)
        out[label] = {"fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1), "hbm_bytes_per_launch": hbm,
                      "algorithmic_bytes_per_launch": int(algo), "ratio": round(hbm / algo, 3)}
        if "grad-input" in label:
            out[label]["note"] = "WRITE_SIZE counts every float atomic request as a write (does not include the caller's memset of grad_x)"
    out["gemm_f32_fwd_bytes_per_launch"] = out["gemm_f32_fwd (C2, 16 pairs)"]["hbm_bytes_per_launch"]   # read by bench.py
    json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print("%-45s hbm %7.1f MB  algorithmic %7.1f MB  x%.2f" % (k, v["hbm_bytes_per_launch"] / 1e6, v["algorithmic_bytes_per_launch"] / 1e6, v["ratio"]))


if __name__ == "__main__":
    main()
