"""The skeltorch stand-in (master_thesis_amd/skeltorch_shim.py): hook order, logging, checkpoint / resume.

  * a runner written against the hook protocol trains, validates, logs and resumes;
  * (build container only: needs /root/reference) the reference's REAL `thesis.runner.ThesisRunner` base class, imported with
    the shim registered as `skeltorch`, runs its own before / after-epoch tasks, iteration logging and
    save_states_others / load_states_others on top of the shim.
"""
import importlib
import os
import sys
import types

import pytest
import torch

from master_thesis_amd import skeltorch_shim as sk

REF = os.environ.get("TCL_REFERENCE", "/root/reference")


class ToyData(sk.Data):
    def __init__(self):
        super().__init__()
        self.regenerated = []

    def create_datasets(self, data_path):
        g = torch.Generator().manual_seed(0)
        xs = torch.randn(12, 3, 8, 8, generator=g)
        self.datasets = {"train": torch.utils.data.TensorDataset(xs[:8], xs[:8] * 0.5), "validation": torch.utils.data.TensorDataset(xs[8:], xs[8:] * 0.5)}

    def create_loaders(self):
        self.loaders = {k: torch.utils.data.DataLoader(v, batch_size=4) for k, v in self.datasets.items()}

    def regenerate_loader_indexes(self, epoch):
        self.regenerated.append(epoch)


def _toy_hooks(base):
    class ToyRunner(base):
        calls = None

        def init_model(self, device):
            torch.manual_seed(1)
            self.model = torch.nn.Conv2d(3, 3, 3, padding=1).to(device)

        def init_optimizer(self, device):
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.experiment.configuration.get("training", "lr"))

        def init_others(self, device):
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=0.5)
            self.losses_items_ids = ["l1"]
            self.calls = []
            super().init_others(device)

        def train_step(self, it_data, device):
            x, y = it_data
            loss = (self.model(x.to(device)) - y.to(device)).abs().mean()
            if getattr(self, "e_train_losses_items", None) is not None:
                (self.e_train_losses_items if self.model.training else self.e_validation_losses_items)["l1"].append(loss.item())
            self.calls.append("train" if self.model.training else "validation")
            return loss

        def test(self, epoch, device):
            self.calls.append("test")
    return ToyRunner


def test_hook_protocol_train_validate_checkpoint_resume(tmp_path):
    class PlainBase(sk.Runner):
        scheduler = None

        def save_states_others(self):
            return {"scheduler": self.scheduler.state_dict()}

        def load_states_others(self, data):
            self.scheduler.load_state_dict(data["scheduler"])

        def train_after_epoch_tasks(self, device):
            self.test(None, device)
            self.scheduler.step()

    cfg = sk.Configuration({"training": {"lr": 0.01, "batch_size": 4}})
    app = sk.Skeltorch(ToyData(), _toy_hooks(PlainBase)(), cfg, root=str(tmp_path))
    r = app.run("train", epochs_max=2, log_period=1)
    assert r.counters == {"epoch": 2, "train_it": 4, "validation_it": 2}
    assert r.calls == ["train", "train", "validation", "test"] * 2           # per epoch: 2 train batches, 1 validation batch, test
    assert [s for s, _ in app.experiment.tbx.scalars["loss/epoch/train"]] == [1, 2]
    assert r.losses_epoch["train"][2] < r.losses_epoch["train"][1]          # it learns something
    assert app.experiment.checkpoints_get() == [1, 2]
    lr_after_2 = r.optimizer.param_groups[0]["lr"]
    assert lr_after_2 == pytest.approx(0.01 * 0.25)
    # resume: a fresh runner picks up epoch 2's model, optimizer, counters and scheduler, and runs epoch 3 only
    app2 = sk.Skeltorch(ToyData(), _toy_hooks(PlainBase)(), cfg, root=str(tmp_path))
    r2 = app2.run("train", epochs_max=3, log_period=1)
    assert r2.counters == {"epoch": 3, "train_it": 6, "validation_it": 3}
    assert r2.calls == ["train", "train", "validation", "test"]
    assert r2.optimizer.param_groups[0]["lr"] == pytest.approx(0.01 * 0.125)
    assert r2.losses_epoch["train"][1] == r.losses_epoch["train"][1]
    # the 'test' command
    app3 = sk.Skeltorch(ToyData(), _toy_hooks(PlainBase)(), cfg, root=str(tmp_path))
    assert app3.run("test", epoch=2).calls == ["test"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "thesis", "runner.py")), reason="needs the reference checkout (build container only)")
def test_reference_base_runner_runs_on_the_shim(tmp_path):
    """The reference's own ThesisRunner (thesis/runner.py:13-110) with `skeltorch` = the shim: its before / after-epoch
    tasks, iteration logging, TensorBoard scalars and save / load_states_others run unchanged."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_sequences as mgs
    saved = dict(sys.modules)
    try:
        sys.meta_path.insert(0, mgs._AbsentFinder())          # cv2, skimage, ... as empty modules (never touched here)
        mgs.ABSENT.discard("skeltorch")
        sys.modules["skeltorch"] = sk
        for pkg in ["utils", "models", "thesis"]:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, pkg)]
            sys.modules[pkg] = m
        ref_runner = importlib.import_module("thesis.runner")
        cfg = sk.Configuration({"training": {"lr": 0.01, "batch_size": 4}})
        data = ToyData()
        app = sk.Skeltorch(data, _toy_hooks(ref_runner.ThesisRunner)(), cfg, root=str(tmp_path))
        r = app.run("train", epochs_max=2, log_period=1)
        tbx = app.experiment.tbx.scalars
        assert data.regenerated == [1, 2]                                    # thesis/runner.py:64
        assert [s for s, _ in tbx["lr"]] == [1, 2]                           # :63
        assert len(tbx["loss_items_train_iteration/l1"]) == 4 and len(tbx["loss_items_validation_iteration/l1"]) == 2   # :66-74
        assert [s for s, _ in tbx["loss_items_train_epoch/l1"]] == [1, 2]    # :76-83
        assert r.calls.count("test") == 2                                    # :82
        assert set(r.losses_epoch_items["train"]["l1"]) == {1, 2}
        app2 = sk.Skeltorch(ToyData(), _toy_hooks(ref_runner.ThesisRunner)(), cfg, root=str(tmp_path))
        r2 = app2.run("train", epochs_max=3, log_period=1)
        assert r2.counters["epoch"] == 3 and set(r2.losses_epoch_items["train"]["l1"]) == {1, 2, 3}    # :47-54 round trip
    finally:
        mgs.ABSENT.add("skeltorch")
        sys.meta_path[:] = [f for f in sys.meta_path if not isinstance(f, mgs._AbsentFinder)]
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.modules.update(saved)
