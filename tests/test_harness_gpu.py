"""GPU exercise of the config harnesses around the hot path (BASELINE.json configs[4] = C5 here; C3 / C4 are covered by
tests/test_alignment_model.py, tests/test_cpn_original.py and tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from oracle import tcl_oracle as O
from tests.conftest import assert_close, assert_exact

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_c5_train_step_one_gpu():
    """harness_c5.train_step on one GPU (world size 1: the bucketed all-reduce degenerates to a no-op), 2 sequences, nb = 1:
    frozen aligner on the HIP warp -> RRDB net -> the five loss terms -> GradientBuckets -> Adam.  The hot-path outputs inside
    the step are checked against the oracle; the step must change the parameters and keep the loss finite."""
    from master_thesis_amd import harness_c5, parallel, runners
    from master_thesis_amd.utils.losses import LossesUtils
    g = torch.Generator().manual_seed(5)
    n, S = 2, 64
    x = torch.rand(n, 3, 2, S, S, generator=g).to(DEV)
    y = torch.rand(n, 3, 2, S, S, generator=g).to(DEV)
    m = (torch.rand(n, 1, 2, S // 8, S // 8, generator=g) < 0.2).float().repeat_interleave(8, 3).repeat_interleave(8, 4).to(DEV)
    lin = torch.linspace(-1, 1, S)
    ident = torch.stack(torch.meshgrid(lin, lin, indexing="ij")[::-1], -1).to(DEV)
    flow = (ident[None, None] * 1.03 + 0.02).expand(n, 1, S, S, 2).contiguous()

    def aligner(x_t, m_t, x_r, m_r):
        return None, None, None, flow

    torch.manual_seed(0)
    net = harness_c5.InpaintingNet(nb=1).to(DEV)
    buckets = parallel.GradientBuckets(net.parameters(), overlap=True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    losses = LossesUtils(harness_c5.PerceptualStandIn().to(DEV), DEV)
    t, r_list = 1, [0]

    # the hot path inside the step
    xa, va, vm = runners.infer_step_propagate(aligner, x[:, :, t], m[:, :, t], x[:, :, r_list], m[:, :, r_list])
    xn, mn = x.cpu().numpy(), m.cpu().numpy()
    oxa, ova = O.align_set(xn[:, :, r_list], 1 - mn[:, :, r_list], flow.cpu().numpy())
    assert_close(xa.cpu().numpy(), oxa, what="x_aligned")
    assert_exact(va.cpu().numpy(), ova, what="v_aligned")
    assert_exact(vm.cpu().numpy(), O.v_map(ova, 1 - mn[:, :, t]), what="v_map")

    before = [p.detach().clone() for p in net.parameters()]
    l0 = harness_c5.train_step(aligner, net, buckets, opt, x, m, y, t, r_list, losses)
    l1 = harness_c5.train_step(aligner, net, buckets, opt, x, m, y, t, r_list, losses)
    torch.cuda.synchronize()
    assert np.isfinite(float(l0)) and np.isfinite(float(l1))
    assert float(l1) != float(l0)
    assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


def test_cpnet_forward_under_autocast():
    """the ops take fp32 only; under bf16 autocast the encoder hands CPNet.copy_and_paste bf16 features, which must
    be cast at the op boundary instead of raising (the reference runs under autocast)."""
    from tests import standins
    net = standins.cpnet_package_net().to(DEV)
    x, m, y, t, r_list = [a.to(DEV) if torch.is_tensor(a) else a for a in standins.cpnet_inputs()]
    with torch.no_grad():
        ref = net(x, m, y, t, r_list)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x, m, y, t, r_list)
    y_hat, y_ref = out[0].float(), ref[0].float()
    assert y_hat.shape == y_ref.shape and torch.isfinite(y_hat).all()
    assert float((y_hat - y_ref).abs().mean()) < 0.05     # bf16 convolutions around fp32 ops


def test_alignment_runner_trains_on_the_skeltorch_shim(tmp_path):
    """thesis_dfpn's training loop end to end without skeltorch: a runner whose train_step is the reference's
    (thesis_dfpn/runner.py:50-76: train_step_propagate -> compute_loss, both on the HIP ops) on the hook protocol of
    master_thesis_amd.skeltorch_shim: one epoch, two training sequences and one validation sequence of three 256 x 256 frames,
    checkpoint written, loss finite, parameters updated."""
    from master_thesis_amd import harness_c3, runners, skeltorch_shim as sk
    from master_thesis_amd.utils.losses import LossesUtils

    class SyntheticData(sk.Data):
        def create_datasets(self, data_path):
            g = torch.Generator().manual_seed(3)
            def sample():
                y = torch.rand(3, 3, 256, 256, generator=g)
                m = (torch.rand(1, 3, 16, 16, generator=g) < 0.1).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
                lin = torch.linspace(-1, 1, 256)
                ident = torch.stack(torch.meshgrid(lin, lin, indexing="ij")[::-1], -1)
                flow_gt = (ident[None] * 1.02 + 0.03 * torch.randn(3, 1, 1, 2, generator=g)).contiguous()
                info = ("seq", 0, True, 0, 0, flow_gt)        # info[2] = flows_use, info[5] = flow_gt, as the dataset's tuple
                return ((1 - m) * y, m), y, info
            self.datasets = {"train": [sample(), sample()], "validation": [sample()]}

        def create_loaders(self):
            self.loaders = {k: torch.utils.data.DataLoader(v, batch_size=1) for k, v in self.datasets.items()}

    class AlignmentRunner(sk.Runner):
        def init_model(self, device):
            torch.manual_seed(0)
            self.model_vgg = harness_c3.FeatureStandIn().to(device)
            self.model = harness_c3.AlignmentNet(self.model_vgg).to(device)

        def init_optimizer(self, device):
            self.optimizer = torch.optim.Adam([p for p in self.model.parameters() if p.requires_grad], lr=self.experiment.configuration.get("training", "lr"))

        def init_others(self, device):
            self.utils_losses = LossesUtils(self.model_vgg, device)
            self.items = []

        def train_step(self, it_data, device):
            (x, m), y, info = it_data
            x, m, y, flows_use, flow_gt = x.to(device), m.to(device), y.to(device), info[2], info[5].to(device)
            t, r_list = x.size(2) // 2, [i for i in range(x.size(2)) if i != x.size(2) // 2]
            out = runners.ThesisAlignmentRunner.train_step_propagate(self.model, x, m, y, flow_gt, flows_use.to(device), t, r_list)
            corr, xs, vs, ys, xs_aligned, _, _, _, flows, flows_gt, use, _, _ = out
            loss, items = runners.ThesisAlignmentRunner.compute_loss(self.model_vgg, self.utils_losses, corr, xs, vs, ys, xs_aligned, flows, flows_gt, use, t, r_list)
            self.items.append([float(i) for i in items])
            return loss

        def test(self, epoch, device):
            pass

    cfg = sk.Configuration({"training": {"lr": 1e-4, "batch_size": 1}})
    app = sk.Skeltorch(SyntheticData(), AlignmentRunner(), cfg, root=str(tmp_path), device=DEV)
    r = app.run("train", epochs_max=1, log_period=1)
    torch.cuda.synchronize()
    assert r.counters == {"epoch": 1, "train_it": 2, "validation_it": 1}
    assert len(r.items) == 3 and all(np.isfinite(v) for it in r.items for v in it)
    assert np.isfinite(r.losses_epoch["train"][1]) and np.isfinite(r.losses_epoch["validation"][1])
    assert app.experiment.checkpoints_get() == [1]
