#!/usr/bin/env python3
"""Golden vector for CPNOriginal (models/cpn_original.py:244-315: align, copy_and_paste, forward) from the REAL reference.

Run ONLY in the build container:   python tests/golden/make_golden_cpn_original.py

Same method as make_golden_cpnet.py: the reference's CPNOriginal LOADS THIS PACKAGE'S state_dict with strict=True (which pins
the parameter names the pretrained weights/cpn/cpn.pth checkpoint uses), then runs its own align / forward on the CPU."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402
from tests import standins  # noqa: E402


def main():
    ref = MG.load_reference()
    mine = standins.cpn_original_package_net()
    ref_net = ref["models.cpn_original"].CPNOriginal()
    res = ref_net.load_state_dict(mine.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ref_net.eval()
    x_target, m_target, x_refs, m_refs = standins.cpn_original_inputs()
    with torch.no_grad():
        x_aligned, v_aligned, v_maps = ref_net.align(x_target, m_target, x_refs, m_refs)
        y_hat, y_hat_comp = ref_net(x_target, m_target, x_refs, m_refs)
    sd = mine.state_dict()
    names = sorted(sd.keys())
    out = dict(x_target=x_target.numpy(), m_target=m_target.numpy(), x_refs=x_refs.numpy(), m_refs=m_refs.numpy(),
               x_aligned=x_aligned.numpy(), v_aligned=v_aligned.numpy(), v_maps=v_maps.numpy(), y_hat=y_hat.numpy(), y_hat_comp=y_hat_comp.numpy(),
               param_names=np.asarray(names), param_numel=np.asarray([sd[k].numel() for k in names]),
               param_abs_sum=np.asarray([float(sd[k].double().abs().sum()) for k in names]))
    np.savez_compressed(os.path.join(HERE, "cpn_original.npz"), **out)
    print("shift |x_aligned - x_refs| %.3f  v_aligned mean %.3f  v_maps mean %.3f  y_hat mean %.3f std %.3f clamped %.3f" % (
        float((x_aligned - x_refs).abs().mean()), float(v_aligned.mean()), float(v_maps.mean()), float(y_hat.mean()), float(y_hat.std()),
        float(((y_hat == 0) | (y_hat == 1)).float().mean())))


if __name__ == "__main__":
    main()
