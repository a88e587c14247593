"""A minimal stand-in for the `skeltorch` package, as far as the reference's runners use it.

The reference's entry points are `skeltorch.Skeltorch(ThesisData(), Runner()).run()` (thesis_dfpn/__main__.py:1-6): skeltorch
(third party, not vendored, not installable here) owns the epoch loop, checkpointing, logging and TensorBoard, and calls the
hooks the reference's runners override.  This module provides exactly those hooks, attributes and call orders, so that a
`ThesisRunner`-style class (thesis/runner.py:13-110) trains end to end around the HIP ops without skeltorch:

  Runner      .experiment .logger .counters{'epoch','train_it','validation_it'} .model .optimizer
              init_model / init_optimizer / init_others(device)      (overridden by the reference)
              train_step(it_data, device) -> loss                    (overridden; called for training AND validation batches)
              train_before_epoch_tasks / train_iteration_log / validation_iteration_log / train_after_epoch_tasks(…)
              save_states_others() / load_states_others(data)        (thesis/runner.py:47-54)
              load_states(epoch, device) / save_states()             ({epoch}.checkpoint.pkl with the key 'model', cf.
                                                                      thesis_dfpn/runner.py:26-31)
              test(epoch, device)
  Experiment  .configuration.get(group, key)   .tbx.add_scalar / add_images   .data   .paths{'checkpoints','results'}
  Data        .datasets .loaders   create_datasets / create_loaders / regenerate_loader_indexes(epoch)
  Skeltorch(data, runner).run(command=…)       'train' | 'test'

It is host-side plumbing (no kernels, no arithmetic of the hot path) and makes no claim about skeltorch beyond the calls the
reference makes; TensorBoard output is recorded in memory (`tbx.scalars`, `tbx.images`).
"""
import json
import logging
import os

import torch


class Configuration:
    """`configuration.get(group, key)` over a {group: {key: value}} dict (the reference's config.*.json layout)."""

    def __init__(self, groups=None):
        self.groups = dict(groups or {})

    @classmethod
    def from_json(cls, path):
        with open(path) as f:
            return cls(json.load(f))

    def get(self, group, key):
        return self.groups[group][key]

    def set(self, group, key, value):
        self.groups.setdefault(group, {})[key] = value


class RecordingTensorBoard:
    """The two SummaryWriter calls the reference makes (thesis/runner.py:63,88-90,106-110,215-217), kept in memory."""

    def __init__(self):
        self.scalars, self.images = {}, {}

    def add_scalar(self, tag, scalar_value, global_step=None):
        self.scalars.setdefault(tag, []).append((global_step, float(scalar_value)))

    def add_images(self, tag, img_tensor, global_step=None):
        self.images.setdefault(tag, []).append((global_step, tuple(img_tensor.shape)))


class Data:
    """Base of the reference's ThesisData (thesis/data.py): subclasses fill `datasets` / `loaders`."""

    def __init__(self):
        self.experiment, self.logger = None, None
        self.datasets, self.loaders = {}, {}

    def init(self, experiment, logger):
        self.experiment, self.logger = experiment, logger

    def create_datasets(self, data_path):
        raise NotImplementedError

    def create_loaders(self):
        raise NotImplementedError

    def regenerate_loader_indexes(self, epoch):
        pass


class Experiment:
    def __init__(self, configuration, data, root=None, name="experiment"):
        self.configuration, self.data, self.name = configuration, data, name
        self.tbx = RecordingTensorBoard()
        self.paths = {}
        if root is not None:
            base = os.path.join(root, name)
            self.paths = {"experiment": base, "checkpoints": os.path.join(base, "checkpoints"), "results": os.path.join(base, "results")}
            for p in self.paths.values():
                os.makedirs(p, exist_ok=True)

    def checkpoint_path(self, epoch):
        return os.path.join(self.paths["checkpoints"], "{}.checkpoint.pkl".format(epoch))

    def checkpoints_get(self):
        if "checkpoints" not in self.paths:
            return []
        return sorted(int(f.split(".")[0]) for f in os.listdir(self.paths["checkpoints"]) if f.endswith(".checkpoint.pkl"))


def _plain(obj):
    """numpy scalars / arrays (the reference's loss histories hold np.mean results) -> Python numbers / tensors, recursively, so that
    a checkpoint stays readable by torch.load(weights_only=True)."""
    import numpy as np
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(obj)
    return obj


class Runner:
    """The hook protocol of skeltorch.Runner as the reference's ThesisRunner uses it."""

    def __init__(self):
        self.experiment, self.logger = None, None
        self.model, self.optimizer = None, None
        self.counters = {"epoch": 0, "train_it": 0, "validation_it": 0}
        self.losses_epoch = {"train": {}, "validation": {}}

    # ---- hooks the reference overrides
    def init_model(self, device):
        raise NotImplementedError

    def init_optimizer(self, device):
        raise NotImplementedError

    def init_others(self, device):
        pass

    def train_step(self, it_data, device):
        raise NotImplementedError

    def test(self, epoch, device):
        raise NotImplementedError

    def load_states_others(self, checkpoint_data):
        pass

    def save_states_others(self):
        return {}

    def train_before_epoch_tasks(self, device):
        self.model.train()

    def train_iteration_log(self, e_train_losses, log_period, device):
        self.logger.info("Train iteration %d | loss %.6f", self.counters["train_it"], sum(e_train_losses[-log_period:]) / max(1, len(e_train_losses[-log_period:])))

    def validation_iteration_log(self, e_validation_losses, log_period, device):
        self.logger.info("Validation iteration %d | loss %.6f", self.counters["validation_it"],
                         sum(e_validation_losses[-log_period:]) / max(1, len(e_validation_losses[-log_period:])))

    def train_after_epoch_tasks(self, device):
        pass

    # ---- the engine
    def init(self, experiment, logger, device):
        self.experiment, self.logger = experiment, logger
        self.init_model(device)
        self.init_optimizer(device)
        self.init_others(device)

    def save_states(self):
        data = {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "counters": dict(self.counters),
                "losses_epoch": self.losses_epoch}
        data.update(self.save_states_others())
        torch.save(_plain(data), self.experiment.checkpoint_path(self.counters["epoch"]))

    def load_states(self, epoch, device):
        # everything save_states writes is tensors + plain containers (state_dicts, counters, float dicts): the weights-only loader
        # reads it and executes nothing from the file (a checkpoint dropped into the experiment directory cannot run code); a
        # subclass whose save_states_others adds another type allow-lists it with torch.serialization.add_safe_globals
        data = torch.load(self.experiment.checkpoint_path(epoch), map_location=device, weights_only=True)
        self.model.load_state_dict(data["model"])
        self.optimizer.load_state_dict(data["optimizer"])
        self.counters, self.losses_epoch = dict(data["counters"]), data["losses_epoch"]
        self.load_states_others(data)

    def train(self, epochs_max, log_period, device):
        """One epoch = before-epoch tasks, the training loader (train_step -> backward -> optimizer step, logging every
        `log_period` iterations), the validation loader under no_grad with the model in eval mode, after-epoch tasks, checkpoint."""
        saved = self.experiment.checkpoints_get()
        if saved:
            self.load_states(saved[-1], device)
        for epoch in range(self.counters["epoch"] + 1, epochs_max + 1):
            self.counters["epoch"] = epoch
            self.train_before_epoch_tasks(device)
            e_train, e_val = [], []
            for it_data in self.experiment.data.loaders["train"]:
                self.counters["train_it"] += 1
                self.optimizer.zero_grad()
                loss = self.train_step(it_data, device)
                loss.backward()
                self.optimizer.step()
                e_train.append(float(loss.detach()))
                if self.counters["train_it"] % log_period == 0:
                    self.train_iteration_log(e_train, log_period, device)
            self.model.eval()
            for it_data in self.experiment.data.loaders.get("validation", []):
                self.counters["validation_it"] += 1
                with torch.no_grad():
                    loss = self.train_step(it_data, device)
                e_val.append(float(loss.detach()))
                if self.counters["validation_it"] % log_period == 0:
                    self.validation_iteration_log(e_val, log_period, device)
            mean = lambda v: sum(v) / len(v) if v else float("nan")
            self.losses_epoch["train"][epoch], self.losses_epoch["validation"][epoch] = mean(e_train), mean(e_val)
            self.experiment.tbx.add_scalar("loss/epoch/train", self.losses_epoch["train"][epoch], epoch)
            if e_val:
                self.experiment.tbx.add_scalar("loss/epoch/validation", self.losses_epoch["validation"][epoch], epoch)
            self.train_after_epoch_tasks(device)
            if "checkpoints" in self.experiment.paths:
                self.save_states()


class Skeltorch:
    """`Skeltorch(data, runner).run(...)`: wires data, experiment and runner together and runs a command."""

    def __init__(self, data, runner, configuration=None, root=None, name="experiment", device="cpu"):
        self.data, self.runner, self.device = data, runner, device
        self.logger = logging.getLogger("skeltorch_shim")
        self.experiment = Experiment(configuration or Configuration(), data, root, name)
        data.init(self.experiment, self.logger)

    def run(self, command="train", epochs_max=1, log_period=1, epoch=None, data_path=None):
        if not self.data.datasets:
            self.data.create_datasets(data_path)
        if not self.data.loaders:
            self.data.create_loaders()
        self.runner.init(self.experiment, self.logger, self.device)
        if command == "train":
            self.runner.train(epochs_max, log_period, self.device)
        elif command == "test":
            self.runner.test(epoch, self.device)
        else:
            raise ValueError("unknown command %r" % (command,))
        return self.runner
