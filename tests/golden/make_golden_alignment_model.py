#!/usr/bin/env python3
"""Golden vector for ThesisAlignmentModel.forward (models/thesis_alignment.py:166-225) from the REAL reference.

Run ONLY in the build container:   python tests/golden/make_golden_alignment_model.py      (about a minute of CPU convolutions)

Same method as make_golden_cpnet.py: the reference's ThesisAlignmentModel (around the seeded stand-in feature extractor of
SURVEY 8c; the pretrained VGG-16 is not available offline) LOADS THIS PACKAGE'S state_dict with strict=True, then runs its
own forward on the CPU, calling the reference's own utils.correlation.  Inputs, the four outputs and per-parameter checksums
go to alignment_model.npz (data only; the volume is stored whole: 2 x 256 x 256 floats)."""
import importlib.util
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
    MG.load_reference()
    for name, rel in [("utils.movement", "utils/movement.py"), ("models.thesis_alignment", "models/thesis_alignment.py")]:
        spec = importlib.util.spec_from_file_location(name, os.path.join(MG.REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        setattr(sys.modules[name.split(".")[0]], name.split(".")[1], mod)
    ref_ta = sys.modules["models.thesis_alignment"]
    from master_thesis_amd import harness_c3
    mine = standins.alignment_package_net()
    ref_net = ref_ta.ThesisAlignmentModel(harness_c3.FeatureStandIn())
    res = ref_net.load_state_dict(mine.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ref_net.eval()
    x_target, m_target, x_refs, m_refs = standins.alignment_inputs()
    with torch.no_grad():
        corr, flow_16, flow_64, flow_hw = ref_net(x_target, m_target, x_refs, m_refs)
    sd = mine.state_dict()
    names = sorted(sd.keys())
    out = dict(x_target=x_target.numpy(), m_target=m_target.numpy(), x_refs=x_refs.numpy(), m_refs=m_refs.numpy(),
               corr=corr.numpy(), flow_16=flow_16.numpy(), flow_64=flow_64.numpy(), flow_hw=flow_hw.numpy(),
               param_names=np.asarray(names), param_numel=np.asarray([sd[k].numel() for k in names]),
               param_abs_sum=np.asarray([float(sd[k].double().abs().sum()) for k in names]))
    np.savez_compressed(os.path.join(HERE, "alignment_model.npz"), **out)
    for k in ("corr", "flow_16", "flow_64", "flow_hw"):
        print(k, out[k].shape, "mean %.4g std %.4g max|.| %.4g" % (out[k].mean(), out[k].std(), np.abs(out[k]).max()))


if __name__ == "__main__":
    main()
