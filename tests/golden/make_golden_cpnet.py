#!/usr/bin/env python3
"""Golden vector for the whole CPNet.forward (models/cpn.py:244-363, SURVEY 8b boundary row) from the REAL reference.

Run ONLY in the build container:   python tests/golden/make_golden_cpnet.py

The reference's CPNet('full') is instantiated, this package's CPNet('full') is built from a fixed seed, and the reference
net LOADS THE PACKAGE'S state_dict with strict=True: that pins the sub-module names / parameter shapes (a checkpoint of one
loads in the other) and gives both the same weights without storing 86 MB of them.  The reference then runs forward on the CPU
(eval mode) on a small seeded sequence; inputs, every output and a per-parameter checksum are stored in cpnet.npz (data only).
The GPU test rebuilds the weights from the same seed, checks the checksums, and compares its forward with these outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402

from tests import standins  # noqa: E402

SEED = standins.CPNET_SEED
build_package_net, inputs = standins.cpnet_package_net, standins.cpnet_inputs


def main():
    ref = MG.load_reference()
    mine = build_package_net()
    ref_net = ref["models.cpn"].CPNet('full')
    missing = ref_net.load_state_dict(mine.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ref_net.eval()
    x, m, y, t, r_list = inputs()
    with torch.no_grad():
        y_hat, y_hat_comp, c_mask, ref_importance, (x_aligned, v_aligned) = ref_net(x, m, y, t, r_list)
    sd = mine.state_dict()
    names = sorted(sd.keys())
    out = dict(x=x.numpy(), m=m.numpy(), y=y.numpy(), t=np.asarray(t), r_list=np.asarray(r_list),
               y_hat=y_hat.numpy(), y_hat_comp=y_hat_comp.numpy(), c_mask=c_mask.numpy(), ref_importance=ref_importance.numpy(),
               x_aligned=x_aligned.numpy(), v_aligned=v_aligned.numpy(),
               param_names=np.asarray(names), param_numel=np.asarray([sd[k].numel() for k in names]),
               param_abs_sum=np.asarray([float(sd[k].double().abs().sum()) for k in names]), seed=np.asarray(SEED))
    np.savez_compressed(os.path.join(HERE, "cpnet.npz"), **out)
    print("cpnet.npz", {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})
    print("v_aligned mean", float(v_aligned.mean()), "y_hat mean", float(y_hat.mean()), "theta-driven shift: x_aligned vs x", float((x_aligned - x[:, :, r_list]).abs().mean()))


if __name__ == "__main__":
    main()
