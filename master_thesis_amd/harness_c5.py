"""Harness for BASELINE.json configs[4] (C5): the data-parallel inpainting training step around the HIP hot path.

Only the hot path is this package's product; the rest of the step exists here so that C5 can run end to end:

  frozen aligner   -> `runners.infer_step_propagate` (align_set + v_map on the HIP kernels; the flow comes from a
                      stand-in callable because the alignment network's conv stacks and weights are out of scope)
  inpainting net   -> `InpaintingNet`: the architecture of ThesisInpaintingVisible / RRDBNet(in=9, out=3, nb=20, nf=64, gc=32)
                      (models/thesis_inpainting.py:6-36, models/rrdb_net.py:14-76) restated on stock torch convolutions
                      (MIOpen); 14,690,883 parameters = 58.76 MB of fp32 gradients, as SURVEY.md §8d records
  loss             -> all five terms of ThesisInpaintingRunner.compute_loss (thesis_inpainting/runner.py:114-128) through
                      `runners.ThesisInpaintingRunner.compute_loss`; the perceptual term runs on a seeded stand-in extractor of
                      VGG-16's layout (pretrained weights are not available offline)
  gradient exchange-> `parallel.GradientBuckets`: the one collective of the whole system (RCCL all-reduce over xGMI)

Run:  python -m master_thesis_amd.harness_c5 --sequences 8            (1 GPU)
      python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 -m master_thesis_amd.harness_c5 --sequences 512
"""
import argparse
import json
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResidualDenseBlock(nn.Module):
    def __init__(self, nf=64, gc=32):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(nf + i * gc, gc if i < 4 else nf, 3, 1, 1) for i in range(5)])

    def forward(self, x):
        feats = [x]
        for i, conv in enumerate(self.convs):
            out = conv(torch.cat(feats, 1))
            if i < 4:
                feats.append(F.leaky_relu(out, 0.2))
        return out * 0.2 + x


class RRDB(nn.Module):
    def __init__(self, nf=64, gc=32):
        super().__init__()
        self.blocks = nn.Sequential(*[ResidualDenseBlock(nf, gc) for _ in range(3)])

    def forward(self, x):
        return self.blocks(x) * 0.2 + x


class InpaintingNet(nn.Module):
    """(x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps) -> (y_hat, y_hat_comp), per reference frame."""

    def __init__(self, nb=20, nf=64, gc=32):
        super().__init__()
        self.head = nn.Sequential(nn.Conv2d(9, nf, 3, padding=1), nn.Conv2d(nf, nf, 3, padding=1), nn.Conv2d(nf, nf, 3, stride=2, padding=1),
                                  nn.Conv2d(nf, nf, 3, padding=1), nn.Conv2d(nf, nf, 3, stride=2, padding=1))
        self.trunk = nn.Sequential(*[RRDB(nf, gc) for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, 3, 1, 1)
        self.up1, self.up2 = nn.Conv2d(nf, nf, 3, 1, 1), nn.Conv2d(nf, nf, 3, 1, 1)
        self.hr, self.last = nn.Conv2d(nf, nf, 3, 1, 1), nn.Conv2d(nf, 3, 3, 1, 1)
        self.register_buffer('mean', torch.as_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1))
        self.register_buffer('std', torch.as_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1))

    def backbone(self, x):
        fea = self.head(x)
        fea = fea + self.trunk_conv(self.trunk(fea))
        fea = F.leaky_relu(self.up1(F.interpolate(fea, scale_factor=2, mode='nearest')), 0.2)
        fea = F.leaky_relu(self.up2(F.interpolate(fea, scale_factor=2, mode='nearest')), 0.2)
        return self.last(F.leaky_relu(self.hr(fea), 0.2))

    def forward(self, x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps):
        b, c, f, h, w = x_refs_aligned.size()
        x_target = x_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        v_target = v_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        nn_input = torch.cat([(x_target - self.mean) / self.std, (x_refs_aligned - self.mean) / self.std, v_target, v_refs_aligned, v_maps], dim=1)
        out = self.backbone(nn_input.transpose(1, 2).reshape(b * f, 9, h, w)).reshape(b, f, c, h, w).transpose(1, 2)
        y_hat = torch.clamp(out * self.std + self.mean, 0, 1)
        return y_hat, v_target * x_target + (1 - v_target) * y_hat


class PerceptualStandIn(nn.Module):
    """Five feature levels at strides 2..32 with VGG-16's widths (the interface of models/vgg_16.py:18-27: a list of the five
    max-pool outputs).  The reference's perceptual term uses pretrained VGG-16 weights, which cannot be fetched here; a seeded,
    frozen stack of the same layout keeps the term (and its share of the step) in place."""

    def __init__(self):
        super().__init__()
        widths, c = (64, 128, 256, 512, 512), 3
        self.levels = nn.ModuleList()
        for wd in widths:
            self.levels.append(nn.Sequential(nn.Conv2d(c, wd, 3, stride=2, padding=1), nn.ReLU()))
            c = wd
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x, normalize_input=False):
        out = []
        for level in self.levels:
            x = level(x)
            out.append(x)
        return out


def compute_loss(utils_losses, y_target, v_target, y_hat, y_hat_comp, v_map):
    """All five terms of thesis_inpainting/runner.py:114-128 (non-hole 0.5, visible hole 2, non-visible hole 1, perceptual 0.5,
    Sobel gradient 1) through the drop-in static function."""
    from . import runners
    return runners.ThesisInpaintingRunner.compute_loss(utils_losses, y_target, v_target, y_hat, y_hat_comp, v_map)[0]


def train_step(aligner, net, buckets, optimizer, x, m, y, t, r_list, utils_losses):
    """One data-parallel step on this rank's shard: frozen aligner (hot path) -> net -> loss -> bucketed all-reduce -> Adam."""
    from . import runners
    x_al, v_al, v_map = runners.infer_step_propagate(aligner, x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])
    y_hat, y_hat_comp = net(x[:, :, t], 1 - m[:, :, t], x_al, v_al, v_map)
    loss = compute_loss(utils_losses, y[:, :, t], 1 - m[:, :, t], y_hat, y_hat_comp, v_map)
    optimizer.zero_grad(set_to_none=False)
    loss.backward()
    buckets.reduce().wait()
    optimizer.step()
    return loss.detach()


def main():
    import torch.distributed as dist

    from . import parallel
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=8, help="sequences of the whole job per step (C5: 512 on 8 GPUs)")
    ap.add_argument("--micro-batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--nb", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    a, b = parallel.partition(args.sequences, world, rank)
    g = torch.Generator().manual_seed(1234 + rank)
    S, n = args.size, b - a
    x = torch.rand(n, 3, 2, S, S, generator=g).to(dev)                 # frames_n = 2 (config.inpainting.json:14)
    m = (torch.rand(n, 1, 2, S // 16, S // 16, generator=g) < 0.1).float().repeat_interleave(16, 3).repeat_interleave(16, 4).to(dev)
    y = torch.rand(n, 3, 2, S, S, generator=g).to(dev)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    ident = torch.stack([xs, ys], -1).to(dev)

    def aligner(x_t, m_t, x_r, m_r):   # stand-in for the frozen alignment network: a fixed smooth flow per pair
        return None, None, None, (ident[None, None] * 1.03 + 0.02).expand(x_r.size(0), x_r.size(2), S, S, 2).contiguous()

    torch.manual_seed(0)
    net = InpaintingNet(nb=args.nb).to(dev)
    from .utils.losses import LossesUtils
    utils_losses = LossesUtils(PerceptualStandIn().to(dev), dev)
    buckets = parallel.GradientBuckets(net.parameters(), overlap=True)   # all-reduce of a bucket starts under the rest of backward
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    t, r_list = 1, [0]

    def step():
        for i in range(0, n, args.micro_batch):
            sl = slice(i, min(i + args.micro_batch, n))
            loss = train_step(aligner, net, buckets, opt, x[sl], m[sl], y[sl], t, r_list, utils_losses)
        return loss

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
        print(json.dumps({"workload": "C5: inpainting training step, data parallel", "n_gpus": world, "sequences_per_step": args.sequences,
                          "pairs_per_s": round(args.sequences * len(r_list) * args.steps / el, 2), "ms_per_step": round(el / args.steps * 1e3, 2),
                          "gradient_bytes": buckets.nbytes(), "buckets": len(buckets.buckets), "loss": float(loss)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
