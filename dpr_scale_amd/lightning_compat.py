"""Minimal stand-ins for the parts of pytorch-lightning 1.6.x that DenseRetrieverTask touches, used ONLY when
pytorch_lightning is not importable (it is not installed in the build image; there is no network).  With the
real package present, dpr_scale_amd.task.dpr_task derives from the real LightningModule and none of this runs.

Covered surface (what dpr_scale/task/dpr_task.py and dpr_scale/main.py use):
  LightningModule: save_hyperparameters, log, log_dict, all_gather, trainer, global_rank, device
  Trainer: fit / test over a datamodule with the hooks setup, configure_optimizers, training_step,
           validation_step + validation_epoch_end, test_step + test_epoch_end, gradient_clip_val, max_steps /
           max_epochs, and a ModelCheckpoint-layout save (state_dict + hyper_parameters + bookkeeping).
This is plumbing for the CPU-runnable BASELINE configs[0]; it is not a Lightning re-implementation.
"""
import inspect
import os
from types import SimpleNamespace

import torch
import torch.distributed as dist


class DDPStrategy:  # markers for the isinstance check the reference does (dpr_task.py:165)
    pass


class DDPShardedStrategy:
    pass


class SingleDeviceStrategy:
    pass


class LightningModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.trainer = None
        self._hparams = {}
        self.logged = {}

    # -- hyper-parameters (checkpoint key "hyper_parameters") -------------------------------------------
    def save_hyperparameters(self, *args, **kwargs):
        frame = inspect.currentframe().f_back
        params = inspect.signature(type(self).__init__).parameters
        self._hparams = {k: frame.f_locals[k] for k in params if k != "self" and k in frame.f_locals}

    @property
    def hparams(self):
        return self._hparams

    @property
    def global_rank(self):
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name, value, **kw):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def log_dict(self, d, **kw):
        for k, v in d.items():
            self.log(k, v, **kw)

    def all_gather(self, data, group=None, sync_grads=False):
        """PL 1.6.4 semantics: per tensor, torch.distributed.all_gather stacked to [W, ...], no grad."""
        if not (dist.is_available() and dist.is_initialized()):
            one = lambda t: t.unsqueeze(0)
        else:
            W = dist.get_world_size(group)

            def one(t):
                with torch.no_grad():
                    t = t.contiguous()
                    out = [torch.zeros_like(t) for _ in range(W)]
                    dist.all_gather(out, t, group=group)
                    return torch.stack(out, 0)

        if isinstance(data, (tuple, list)):
            return type(data)(one(t) for t in data)
        return one(data)


class ModelCheckpoint:
    """Writes <dirpath>/<filename>.ckpt (best by `monitor`) and last.ckpt in the PL checkpoint layout."""

    def __init__(self, dirpath=".", monitor="valid_mrr", mode="max", save_last=True, filename="checkpoint_best",
                 save_top_k=3, verbose=False, **_):
        self.dirpath, self.monitor, self.mode, self.save_last, self.filename = dirpath, monitor, mode, save_last, filename
        self.best_model_score, self.best_model_path = None, ""

    def on_validation_end(self, trainer, task):
        os.makedirs(self.dirpath, exist_ok=True)
        score = task.logged.get(self.monitor)
        if score is not None:
            score = float(score)
            better = self.best_model_score is None or (score > self.best_model_score if self.mode == "max"
                                                       else score < self.best_model_score)
            if better:
                self.best_model_score = score
                self.best_model_path = os.path.join(self.dirpath, self.filename + ".ckpt")
                trainer.save_checkpoint(self.best_model_path)
        if self.save_last:
            trainer.save_checkpoint(os.path.join(self.dirpath, "last.ckpt"))


class Trainer:
    def __init__(self, max_epochs=1, max_steps=None, gradient_clip_val=0.0, callbacks=None, log_every_n_steps=10,
                 limit_train_batches=None, limit_val_batches=None, device=None, precision=32, **_):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.gradient_clip_val = gradient_clip_val
        self.callbacks = callbacks or []
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.precision = precision
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.strategy = DDPStrategy() if self.world_size > 1 else SingleDeviceStrategy()
        self.global_step, self.current_epoch = 0, 0
        self.datamodule, self.task, self.optimizers, self.schedulers = None, None, [], []
        self.train_losses = []

    def _to_device(self, batch):
        if torch.is_tensor(batch):
            return batch.to(self.device)
        if isinstance(batch, dict):
            return {k: self._to_device(v) for k, v in batch.items()}
        if hasattr(batch, "to") and not isinstance(batch, (str, bytes)):
            return batch.to(self.device)
        return batch

    def _attach(self, task, datamodule):
        self.task, self.datamodule = task, datamodule
        task.trainer = self

    def fit(self, task, datamodule=None):
        self._attach(task, datamodule)
        task.setup("fit")
        task.to(self.device)
        opts, scheds = task.configure_optimizers()
        self.optimizers, self.schedulers = opts, scheds
        if hasattr(task, "on_pretrain_routine_start") and not getattr(task, "fp16_grads", False):
            task.on_pretrain_routine_start()
        done = False
        for epoch in range(self.max_epochs or 1):
            self.current_epoch = epoch
            task.train()
            for bi, batch in enumerate(datamodule.train_dataloader()):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                loss = task.training_step(self._to_device(batch), bi)
                for o in opts:
                    o.zero_grad(set_to_none=True)
                loss.backward()
                if self.gradient_clip_val:
                    torch.nn.utils.clip_grad_norm_(task.parameters(), self.gradient_clip_val)
                for o in opts:
                    o.step()
                for s in scheds:
                    s["scheduler"].step()
                self.train_losses.append(float(loss.detach()))
                self.global_step += 1
                if self.max_steps and self.max_steps > 0 and self.global_step >= self.max_steps:
                    done = True
                    break
            self._run_eval("validation")
            for cb in self.callbacks:
                if hasattr(cb, "on_validation_end"):
                    cb.on_validation_end(self, task)
            if done:
                break

    def _run_eval(self, kind):
        task, dm = self.task, self.datamodule
        loader = dm.val_dataloader() if kind == "validation" else dm.test_dataloader()
        if loader is None:
            return
        task.eval()
        outs = []
        with torch.no_grad():
            for bi, batch in enumerate(loader):
                if self.limit_val_batches is not None and bi >= self.limit_val_batches:
                    break
                step = task.validation_step if kind == "validation" else task.test_step
                outs.append(step(self._to_device(batch), bi))
        (task.validation_epoch_end if kind == "validation" else task.test_epoch_end)(outs)

    def test(self, model=None, datamodule=None, ckpt_path=None, verbose=False):
        task = model or self.task
        self._attach(task, datamodule or self.datamodule)
        task.setup("test")
        if ckpt_path and ckpt_path != "best" and os.path.isfile(ckpt_path):
            ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            task.on_load_checkpoint(ck)
            task.load_state_dict(ck["state_dict"])
        task.to(self.device)
        self._run_eval("test")
        return [{k: (float(v) if torch.is_tensor(v) else v) for k, v in task.logged.items()}]

    def save_checkpoint(self, path):
        """The dict layout of a pytorch-lightning 1.6 checkpoint."""
        task = self.task
        ck = {
            "epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "1.6.4",
            "state_dict": task.state_dict(), "hyper_parameters": _plain(task.hparams),
            "optimizer_states": [o.state_dict() for o in self.optimizers],
            "lr_schedulers": [s["scheduler"].state_dict() for s in self.schedulers], "callbacks": {},
        }
        if self.world_size == 1 or task.global_rank == 0:
            torch.save(ck, path)


def _plain(x):
    """Config containers -> plain dict / list (checkpoints stay loadable without this package)."""
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def load_from_checkpoint(cls, path, **overrides):
    """DenseRetrieverTask.load_from_checkpoint for the stand-in (drboost_task.py:29, spar_task.py:31-32 use it)."""
    # PL's own keyword arguments (drboost_task.py:29 passes map_location) are not hyper-parameters of the task
    map_location = overrides.pop("map_location", "cpu")
    strict = overrides.pop("strict", True)
    overrides.pop("hparams_file", None)
    ck = torch.load(path, map_location=map_location if map_location is not None else "cpu", weights_only=False)
    hp = dict(ck.get("hyper_parameters", {}))
    hp.update(overrides)
    task = cls(**hp)
    task.on_load_checkpoint(ck)
    task.load_state_dict(ck["state_dict"], strict=strict)
    return task


strategies = SimpleNamespace(DDPStrategy=DDPStrategy, DDPShardedStrategy=DDPShardedStrategy)
