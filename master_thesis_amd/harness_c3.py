"""Harness for BASELINE.json configs[0] / configs[2] (C1 / C3): the dense-flow alignment network's step around the HIP hot path.

Only the hot path is this package's product; the network exists here so that the configuration can run end to end and
the share of the hot path in a real step can be measured:

  feature extractor -> `FeatureStandIn`: a seeded conv stack with VGG-16's stride (256 px -> 16 x 16 x 512).  The reference
                       uses pretrained VGG-16 weights (models/vgg_16.py) that are not available offline; SURVEY.md §8d
                       prescribes the seeded stand-in.
  cost volume       -> `utils.correlation.compute_masked_4d_correlation` on the HIP kernels (A1), 16 x 16 x 512
  the network       -> `models.thesis_alignment.ThesisAlignmentModel` of this package (the reference's layer structure,
                       models/thesis_alignment.py:8-225: separable 4-D convolution over the volume, correlation mixer, two flow
                       estimators, on stock torch convolutions, state_dict-compatible); bf16 autocast for the convolutions in
                       C3, the volume and the warps stay fp32
  after the network -> `runners.train_step_propagate`: six `align_set` warps at 16 / 64 / 256 px, two `resize_flow`s, `v_map`s
  loss              -> `runners.ThesisAlignmentRunner.compute_loss` (thesis_dfpn/runner.py:162-206, reference signature)

Run:  python -m master_thesis_amd.harness_c3 --sequences 32 --bf16          (C3, 1 GPU: 128 pairs per step)
      python -m master_thesis_amd.harness_c3 --sequences 1 --forward-only    (C1's shape on the GPU)
      python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m master_thesis_amd.harness_c3 --sequences 256 --bf16
"""
import argparse
import json
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import runners
from .models import thesis_alignment
from .utils.losses import LossesUtils


def conv_stack(spec, in_c, final_relu=True):
    """spec: list of (out_c, kernel, stride) | ('T', out_c, kernel) for a stride-2 transposed convolution."""
    layers, c = [], in_c
    for item in spec:
        if item[0] == 'T':
            _, oc, k = item
            layers.append(nn.ConvTranspose2d(c, oc, k, stride=2, padding=k // 2, output_padding=1))
        else:
            oc, k, st = item
            layers.append(nn.Conv2d(c, oc, k, stride=st, padding=k // 2))
        layers.append(nn.ReLU())
        c = oc
    if not final_relu:
        layers.pop()
    return nn.Sequential(*layers)


class FeatureStandIn(nn.Module):
    """(N,3,256,256) -> [.., .., .., (N,512,16,16)] like the VGG wrapper's fourth feature level (stride 16)."""

    def __init__(self):
        super().__init__()
        self.body = conv_stack([(64, 3, 2), (128, 3, 2), (256, 3, 2), (512, 3, 2)], 3)
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x, normalize_input=False):
        return [None, None, None, self.body(x)]


def AlignmentNet(features):
    """The drop-in network (models.thesis_alignment.ThesisAlignmentModel of this package, pinned against the reference's
    forward by tests/golden/alignment_model.npz) around the given feature extractor."""
    return thesis_alignment.ThesisAlignmentModel(features)


def main():
    import torch.distributed as dist

    from . import parallel
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=8, help="sequences of the whole job per step (C3: 32 per GPU)")
    ap.add_argument("--micro-batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--bf16", action="store_true", help="bf16 autocast for the stock convolutions (C3)")
    ap.add_argument("--forward-only", action="store_true", help="C1: the forward of the network + runner step, no loss / backward")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lo, hi = parallel.partition(args.sequences, world, rank)
    n, S, frames = hi - lo, 256, 5
    t, r_list = 2, [0, 1, 3, 4]
    g = torch.Generator().manual_seed(4321 + rank)
    y = torch.rand(n, 3, frames, S, S, generator=g).to(dev)
    m = (torch.rand(n, 1, frames, S // 16, S // 16, generator=g) < 0.1).float().repeat_interleave(16, 3).repeat_interleave(16, 4).to(dev)
    x = (1 - m) * y + m * torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1, 1)
    ys_, xs_ = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    ident = torch.stack([xs_, ys_], -1).to(dev)
    flow_gt = (ident[None, None] * 1.02 + 0.03 * torch.randn(n, frames, 1, 1, 2, generator=g).to(dev)).contiguous()
    flows_use = torch.ones(n, dtype=torch.bool, device=dev)

    torch.manual_seed(0)
    features = FeatureStandIn().to(dev)
    net = AlignmentNet(features).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    buckets = parallel.GradientBuckets(params, overlap=True)
    opt = torch.optim.Adam(params, lr=1e-4)
    losses = LossesUtils(None, dev)
    hot = {"ms": 0.0}

    def one(sl):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            out = runners.train_step_propagate(net, x[sl], m[sl], y[sl], flow_gt[sl], flows_use[sl], t, r_list)
        if args.forward_only:
            return out[0].float().mean()
        corr, xs, vs, ys, xs_al, _, _, _, flows, flows_gt, use, _, _ = out
        loss, _ = runners.ThesisAlignmentRunner.compute_loss(features, losses, corr, xs, vs, ys, xs_al, flows, flows_gt, use, t, r_list)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        buckets.reduce().wait()
        opt.step()
        return loss.detach()

    def step():
        for i in range(0, n, args.micro_batch):
            loss = one(slice(i, min(i + args.micro_batch, n)))
        return loss

    ctx = torch.no_grad() if args.forward_only else torch.enable_grad()
    with ctx:
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"workload": "C1: alignment network forward" if args.forward_only else "C3: alignment network training step",
                          "n_gpus": world, "sequences_per_step": args.sequences, "convs": "bf16 autocast" if args.bf16 else "fp32",
                          "pairs_per_s": round(args.sequences * len(r_list) * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 2),
                          "parameters": sum(p.numel() for p in params), "loss": float(loss)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
