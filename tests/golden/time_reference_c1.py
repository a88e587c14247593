#!/usr/bin/env python3
"""BASELINE.json configs[0] (C1) on the CPU with the REAL reference: ThesisAlignmentModel forward (reference correlation on
ATen-CPU) + the runner's align_set (grid_sample) on one synthetic 5-frame 256 x 256 sequence, fp32.  Build container only
(needs /root/reference); prints one JSON line that DESIGN.md quotes beside the GPU harness's C1 time."""
import importlib.util
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402


def main():
    ref = MG.load_reference()
    for name, rel in [("utils.movement", "utils/movement.py"), ("models.thesis_alignment", "models/thesis_alignment.py")]:
        spec = importlib.util.spec_from_file_location(name, os.path.join(MG.REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    from master_thesis_amd import harness_c3
    torch.manual_seed(0)
    net = sys.modules["models.thesis_alignment"].ThesisAlignmentModel(harness_c3.FeatureStandIn()).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 3, 5, 256, 256, generator=g)
    m = (torch.rand(1, 1, 5, 16, 16, generator=g) < 0.1).float().repeat_interleave(16, 3).repeat_interleave(16, 4)
    t, r_list = 2, [0, 1, 3, 4]

    def step():
        with torch.no_grad():
            corr, f16, f64, f256 = net(x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])
            return ref["utils.flow"].align_set(x[:, :, r_list], 1 - m[:, :, r_list], f256)

    step()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    print(json.dumps({"workload": "C1: reference ThesisAlignmentModel forward + align_set, 1 x 5 frames 256^2, fp32, CPU", "threads": torch.get_num_threads(),
                      "s_per_sequence": round(sorted(times)[1], 3), "pairs_per_s": round(4 / sorted(times)[1], 3), "torch": torch.__version__}))


if __name__ == "__main__":
    main()
