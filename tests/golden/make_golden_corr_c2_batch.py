#!/usr/bin/env python3
"""tests/golden/corr_c2_batch.npz: checksums of the REAL reference's compute_masked_4d_correlation (utils/correlation.py:28-54) and
its autograd gradients at BASELINE configs[1]'s full batch (16 pairs, 64 x 64 maps, C = 256).  Run in the build container
(needs /root/reference; ~1 minute, ~6 GiB); the inputs are regenerated from tests/config_cases.corr_c2_batch by the tests."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import config_cases as cc  # noqa: E402
from tests.golden import make_golden_configs as mgc  # noqa: E402


def main():
    ref = mgc.load_reference()
    d = cc.corr_c2_batch()
    xt, xr = mgc.T(d["x_t"], True), mgc.T(d["x_r"], True)
    out = ref["utils.correlation"].compute_masked_4d_correlation(xt, mgc.T(d["v_t"]), xr, mgc.T(d["v_r"]))
    gt, gr = torch.autograd.grad(out, [xt, xr], grad_outputs=mgc.T(d["cot"]))
    flat = {}
    mgc.put(flat, "corr_c2_batch/out", cc.float_checksum(mgc.npy(out)))
    mgc.put(flat, "corr_c2_batch/g_x_t", cc.float_checksum(mgc.npy(gt)))
    mgc.put(flat, "corr_c2_batch/g_x_r", cc.float_checksum(mgc.npy(gr)))
    np.savez_compressed(os.path.join(HERE, "corr_c2_batch.npz"), **flat)
    print("corr_c2_batch", tuple(out.shape), {k: v.shape for k, v in flat.items()})


if __name__ == "__main__":
    main()
