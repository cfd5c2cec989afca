"""Training-state files (reference: glow_tts_train/checkpoint.py).

The on-disk layout is the reference's, so either side reads the other's files: a `torch.save`d dict
`{"model": state_dict, "global_step": int, "learning_rate": float, "version": int[, "optimizer": Adam state_dict]}`
(checkpoint.py:38-46) whose model keys follow the reference's module tree (weight-norm `weight_g`/`weight_v` pairs
included) and whose optimizer entry is `torch.optim.Adam`'s `{"state": {i: {step, exp_avg, exp_avg_sq}},
"param_groups": [...]}` — `optimize.FlatAdam` slices its flat moment buffers into that shape on save and gathers them
back on load.
"""
from __future__ import annotations

import logging
import typing
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

from .models import ModelType, setup_model
from .optimize import OptimizerType

_LOGGER = logging.getLogger("glow_tts_train.checkpoint")


@dataclass
class Checkpoint:
    model: ModelType
    learning_rate: float
    global_step: int
    version: int
    optimizer: typing.Optional[OptimizerType] = None


def _bare(model):
    """The module whose keys go to disk: a DistributedDataParallel-style wrapper is looked through."""
    return model.module if hasattr(model, "module") else model


def _plain(v):
    """numpy scalars (the Noam rate comes out of numpy arithmetic) -> Python numbers, so the file needs no allow-list."""
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(x) for x in v)
    return v


def _numpy_multiarray():
    """numpy's multiarray module under whichever name this numpy has (numpy >= 2: `numpy._core`, numpy 1.x: `numpy.core`)."""
    try:
        import numpy._core.multiarray as ncm
    except ImportError:                                    # numpy < 2
        import numpy.core.multiarray as ncm
    return ncm


def _read(path):
    """`torch.load` restricted to tensors and plain containers, plus the numpy scalar types the reference's own files
    carry (`learning_rate` and the optimizer's `lr` are numpy float64 there: optimize.py:32-48, checkpoint.py:41).
    A file pickled under numpy 1.x names the reconstructor `numpy.core.multiarray.scalar`, one pickled under numpy >= 2
    `numpy._core.multiarray.scalar`; both spellings are allow-listed whichever numpy is installed here."""
    ncm = _numpy_multiarray()
    allowed = [ncm.scalar, (ncm.scalar, "numpy.core.multiarray.scalar"), (ncm.scalar, "numpy._core.multiarray.scalar"),
               np.dtype]
    allowed += [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64, np.int32)]
    with torch.serialization.safe_globals(allowed):
        return torch.load(path, map_location="cpu", weights_only=True)


def save_checkpoint(checkpoint: Checkpoint, checkpoint_path: Path):
    """Write model / optimizer / counters to `checkpoint_path` (reference checkpoint.py:27-48).  Tensors are copied to
    host memory first, so the file loads on a machine without a GPU."""
    checkpoint_path = Path(checkpoint_path)
    checkpoint_path.parent.mkdir(parents=True, exist_ok=True)
    out = {
        "model": {k: v.detach().cpu() for k, v in _bare(checkpoint.model).state_dict().items()},
        "global_step": _plain(checkpoint.global_step),
        "learning_rate": _plain(checkpoint.learning_rate),
        "version": _plain(checkpoint.version),
    }
    if checkpoint.optimizer is not None:
        opt = checkpoint.optimizer.state_dict()
        opt["state"] = {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in s.items()}
                        for i, s in opt["state"].items()}
        opt["param_groups"] = [{k: _plain(v) for k, v in g.items()} for g in opt["param_groups"]]
        out["optimizer"] = opt
    torch.save(out, checkpoint_path)


def load_checkpoint(checkpoint_path: Path, config, model: typing.Optional[ModelType] = None,
                    optimizer: typing.Optional[OptimizerType] = None, load_optimizer: bool = True,
                    use_cuda: bool = True) -> Checkpoint:
    """Read a checkpoint written by this package or by the reference (reference checkpoint.py:51-106): model and
    optimizer are created from `config` unless passed in; model entries absent from the file keep their initial values
    (with a warning); missing counters default to version 1 / step 1 / learning rate 1.0."""
    saved = _read(checkpoint_path)
    model, optimizer = setup_model(config, model=model, optimizer=optimizer, create_optimizer=load_optimizer,
                                   use_cuda=use_cuda)
    if load_optimizer and optimizer is not None:
        optimizer.load_state_dict(saved["optimizer"])
    target = _bare(model)
    saved_model = saved["model"]
    merged = {}
    for key, value in target.state_dict().items():
        if key in saved_model:
            merged[key] = saved_model[key]
        else:
            _LOGGER.warning("%s is not in the checkpoint", key)
            merged[key] = value
    target.load_state_dict(merged)
    return Checkpoint(model=model, optimizer=optimizer, learning_rate=float(saved.get("learning_rate", 1.0)),
                      global_step=int(saved.get("global_step", 1)), version=int(saved.get("version", 1)))
