"""A minimal stand-in for the `lightning` package — ONLY what `internal/gaussian_splatting.py` (the reference's LightningModule)
touches while it trains — plus empty stand-ins for the other third-party imports of that module that this container lacks
(torchvision, wandb, viser, torchmetrics, jsonargparse, ...).  Test infrastructure: it lets tests/reference_loop_worker.py run the
reference's UNCHANGED `GaussianSplatting.setup / configure_optimizers / on_train_start / training_step / on_train_batch_end`
with a `gspl_amd` renderer plugged in; nothing of it is used by the product.

What the stand-in reproduces of Lightning's behaviour (manual optimisation, `automatic_optimization = False`):
  * `save_hyperparameters()` collects the constructor arguments into `self.hparams`;
  * `self.optimizers()` hands out wrappers whose `step()` runs `_on_before_step`, the optimizer, `_on_after_step`, and the
    trainer's `global_step` counts those calls — the behaviour `VanillaOptStrategyModule._multiple_optimizer_step_fix`
    (internal/opt_strategies/vanilla.py:15-32) exists to tame;
  * `self.lr_schedulers()`, `self.manual_backward(loss)`, `self.log(...)`, `self.trainer`, `self.logger`, `self.device`.
"""
import importlib.abc
import importlib.machinery
import inspect
import sys
import types

import torch

GENERIC = ("torchvision", "wandb", "viser", "plyfile", "splines", "cv2", "torchmetrics", "jsonargparse", "lightning")


class _Anything:
    """Class handed out for every attribute of a generic stand-in module: constructible, subclassable, callable."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return self

    def to(self, *args, **kwargs):
        return self


class _GenericModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Any submodule of the packages in GENERIC that is not installed resolves to an empty stand-in."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in GENERIC:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _GenericModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


class _OptimizerWrapper:
    """`LightningOptimizer` as far as the reference uses it."""

    def __init__(self, optimizer, trainer):
        self.optimizer, self._trainer = optimizer, trainer
        self._on_before_step = lambda: None
        self._on_after_step = self._count

    def _count(self):
        self._trainer.global_step += 1

    def step(self, *args, **kwargs):
        self._on_before_step()
        out = self.optimizer.step(*args, **kwargs)
        self._on_after_step()
        return out

    def zero_grad(self, *args, **kwargs):
        return self.optimizer.zero_grad(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self.optimizer, name)


class LightningModule(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self._trainer = None
        self._hparams = {}
        self.automatic_optimization = True
        self.logged = {}

    def save_hyperparameters(self, *args, **kwargs):
        frame = inspect.currentframe().f_back
        info = inspect.getargvalues(frame)
        self._hparams = {k: info.locals[k] for k in info.args if k != "self"}

    @property
    def hparams(self):
        return self._hparams

    @property
    def trainer(self):
        if self._trainer is None:
            raise RuntimeError(f"{type(self).__name__} is not attached to a `Trainer`.")
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    @property
    def logger(self):
        return self._trainer.logger if self._trainer is not None else None

    @property
    def device(self):
        return next((p.device for p in self.parameters()), torch.device("cpu"))

    @property
    def global_rank(self):
        return self.trainer.global_rank

    def log(self, name, value, **kwargs):
        self.logged[name] = float(value.detach()) if isinstance(value, torch.Tensor) else float(value)

    def log_dict(self, d, **kwargs):
        for k, v in d.items():
            self.log(k, v)

    def print(self, *args, **kwargs):
        pass

    def manual_backward(self, loss, *args, **kwargs):
        loss.backward(*args, **kwargs)

    def optimizers(self, use_pl_optimizer: bool = True):
        return self.trainer.wrapped_optimizers if use_pl_optimizer else self.trainer.raw_optimizers

    def lr_schedulers(self):
        s = self.trainer.schedulers
        return None if not s else (s[0] if len(s) == 1 else s)

    # hooks the reference chains up to with super()
    def on_train_start(self): pass
    def on_train_batch_start(self, batch, batch_idx): pass
    def on_train_batch_end(self, outputs, batch, batch_idx): pass
    def on_validation_batch_start(self, batch, batch_idx, dataloader_idx=0): pass
    def on_load_checkpoint(self, checkpoint): pass
    def on_save_checkpoint(self, checkpoint): pass
    def transfer_batch_to_device(self, batch, device, dataloader_idx): return batch


class Logger:
    def __init__(self):
        self.metrics = []

    def log_metrics(self, metrics, step=None):
        self.metrics.append((step, dict(metrics)))


class Trainer:
    """What `GaussianSplatting` reads of the trainer; `fit_setup` performs Lightning's call order up to the first batch."""

    def __init__(self, datamodule, max_steps):
        self.datamodule, self.max_steps = datamodule, max_steps
        self.global_step, self.current_epoch = 0, 0
        self.global_rank, self.world_size = 0, 1
        self.strategy, self.logger, self.profiler = object(), Logger(), None
        self.raw_optimizers, self.wrapped_optimizers, self.schedulers = [], [], []

    def fit_setup(self, module):
        module.trainer = self
        module.setup("fit")
        optimizers, schedulers = module.configure_optimizers()
        self.raw_optimizers = list(optimizers)
        self.wrapped_optimizers = [_OptimizerWrapper(o, self) for o in optimizers]
        self.schedulers = list(schedulers)
        module.train()
        module.on_train_start()

    def train_batch(self, module, batch, batch_idx):
        module.on_train_batch_start(batch, batch_idx)
        out = module.training_step(batch, batch_idx)
        module.on_train_batch_end(out, batch, batch_idx)
        return out


def install():
    """Register the stand-ins (idempotent).  `lightning.pytorch.LightningModule` is the functional class above; `jsonargparse.
    lazy_instance(cls, **kw)` constructs the default configuration objects right away."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    sys.meta_path.append(_Finder())
    import lightning
    import lightning.pytorch
    lightning.LightningModule = LightningModule
    lightning.pytorch.LightningModule = LightningModule
    import jsonargparse
    jsonargparse.lazy_instance = lambda cls, **kwargs: cls(**kwargs)
