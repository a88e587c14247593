"""Harness for BASELINE.json configs[3] (C4): copy-and-paste inference around the HIP hot path.

  alignment   -> `CPNOriginal.align`: alignment encoder + regressor (stock convolutions) -> theta; then the affine warp of
                 frames and visibility maps on the HIP kernel (`models.cpn.affine_align`, grid never materialised) and `v_map`
  copy        -> copy encoder (stock convolutions, stride 4: 240 x 424 -> 60 x 106 x 128) -> `models.cpn_original.CM_Module`
                 on the HIP context-matching kernels (similarity, masked softmax over the references, aggregation)
  paste       -> decoder with the dilated block (stock convolutions)

The network is `models.cpn_original.CPNOriginal` of this package (the reference's layer structure, models/cpn_original.py:52-183,
246-315, on stock torch convolutions, state_dict-compatible); the pretrained weights (weights/cpn/cpn.pth) are not available
offline (random init, the regressor's last layer is perturbed so that the thetas are not all the identity).  The reference only runs this network in inference
(thesis_cpn/runner.py:59-60 raises in train_step), so this harness is forward-only.

Run:  python -m master_thesis_amd.harness_c4 --targets 4 --refs 10
"""
import argparse
import json
import time

import torch

from .models import cpn, cpn_original


def CopyPasteNet():
    """The drop-in network (models.cpn_original.CPNOriginal, pinned against the reference's forward by
    tests/golden/cpn_original.npz) with the regressor's last layer perturbed so that the thetas are not all the identity."""
    net = cpn_original.CPNOriginal()
    with torch.no_grad():
        net.A_Regressor.fc.weight.normal_(0, 2e-3)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=4)
    ap.add_argument("--refs", type=int, default=10)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=424)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(7)
    B, R, H, W = args.targets, args.refs, args.height, args.width
    x = torch.rand(B, 3, R + 1, H, W, generator=g).to(dev)
    m = (torch.rand(B, 1, R + 1, H // 8, W // 8, generator=g) < 0.15).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(dev)
    torch.manual_seed(0)
    net = CopyPasteNet().to(dev).eval()
    t, r_list = R // 2, [i for i in range(R + 1) if i != R // 2]

    def step():
        with torch.no_grad():
            return net(x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y_hat, y_comp = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the hot path alone, on the same tensors
    with torch.no_grad():
        xa, va, _ = net.align(x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])
        feats = torch.randn(B, 128, R + 1, H // 4, W // 4, device=dev) * 0.05
        theta = torch.eye(2, 3, device=dev).repeat(B * R, 1, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            cpn.affine_align(x[:, :, r_list], m[:, :, r_list], None, theta, list(range(R)))
            net.CM_Module(feats, 1 - m[:, :, t], va)
        e1.record()
        e1.synchronize()
    print(json.dumps({"workload": "C4: copy-and-paste inference, %dx%d, %d references" % (H, W, R), "targets_per_step": B,
                      "pairs_per_s": round(B * R * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 2),
                      "hot_path_ms_per_step": round(e0.elapsed_time(e1) / 5, 3), "parameters": sum(p.numel() for p in net.parameters()),
                      "y_hat_mean": float(y_hat.mean())}), flush=True)


if __name__ == "__main__":
    main()
