"""The inner training step (reference: glow_tts_train/train.py:91-162) without its host synchronisations.

The reference pays >= 521 device->host syncs per step (two `loss.item()` plus one `.item()` per parameter tensor in
`clip_grad_value_`); here the loss stays on device, clipping and Adam/Noam are one kernel each over flat buffers,
the alignment search never leaves the GPU, and (with `reducer`) gradient all-reduce overlaps the backward.
`train()` is the reference's epoch loop around it (checkpoint cadence included); batches reach HBM one step ahead
through `dataset.DeviceBatches`.
"""
from __future__ import annotations

import logging
import time
import typing
from pathlib import Path

import torch

from ._hip import join_side_streams, zero_scope
from .convops import flush_groups
from .utils import clip_grad_value_, duration_loss, mle_loss

_LOGGER = logging.getLogger("glow_tts_train")


def train(train_loader, config, model_dir: Path, model=None, optimizer=None, global_step: int = 1,
          checkpoint_epochs: int = 1, rank: int = 0, reducer=None):
    """Epoch loop of the reference (train.py:19-88): seed, build or adopt model and optimizer, run `config.epochs`
    passes over `train_loader`, and on rank 0 write `checkpoint_<step>.pth` + `config_<step>.json` into `model_dir`
    every `checkpoint_epochs` epochs.  Returns the final global step."""
    from .checkpoint import Checkpoint, save_checkpoint
    from .models import setup_model

    torch.manual_seed(config.seed)
    model, optimizer = setup_model(config, model=model, optimizer=optimizer)
    assert model is not None and optimizer is not None
    model_dir = Path(model_dir)
    for epoch in range(1, config.epochs + 1):
        started = time.perf_counter()
        global_step = train_step(global_step=global_step, epoch=epoch, model=model, optimizer=optimizer, config=config,
                                 train_loader=train_loader, fp16_run=config.fp16_run, reducer=reducer,
                                 on_loss=lambda e, loss, step: _LOGGER.info(
                                     "Avg. Loss for epoch %s: %s (global step=%s)", e, loss, step))
        if epoch % checkpoint_epochs == 0 and rank == 0:
            path = model_dir / f"checkpoint_{global_step}.pth"
            save_checkpoint(Checkpoint(model=model, optimizer=optimizer, learning_rate=optimizer.cur_lr,
                                       global_step=global_step, version=config.version), path)
            with open(model_dir / f"config_{global_step}.json", "w") as config_file:
                config.save(config_file)
            _LOGGER.info("Saved checkpoint to %s", path)
        _LOGGER.debug("Epoch %s complete in %s second(s) (global step=%s)", epoch, time.perf_counter() - started,
                      global_step)
    return global_step


def train_batch(model, optimizer, batch, grad_clip: float, reducer=None, scaler=None) -> torch.Tensor:
    """One optimisation step on one already-resident batch; returns the (device) loss tensor, un-synchronised.
    `scaler` (a torch GradScaler, reduced-precision runs only): the reference's sequence train.py:133-141 — scale the loss,
    un-scale the gradients before clipping, let the scaler skip the update on overflow."""
    x, x_lengths, y, y_lengths, speaker_ids = batch
    optimizer.zero_grad()
    flat = getattr(optimizer, "_optim", optimizer)
    with zero_scope(y.device):          # the step's atomically-accumulated temporaries share one zero fill
        (z, z_m, z_logs, logdet, z_mask), _, (_attn, logw, logw_) = model(x, x_lengths, y, y_lengths, g=speaker_ids)
        loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, x_lengths)
        (loss if scaler is None else scaler.scale(loss)).backward()
        join_side_streams()             # the encoder branch ran (forward and backward) on a second stream
        flush_groups()                  # weight gradients still packed in a ConvGroup (none, unless a backward was skipped)
        if reducer is not None:
            reducer.finish()
        loss = loss.detach()
    if scaler is not None:
        scaler.unscale_(flat)
    if not (hasattr(flat, "clip_grad_value_") and flat.clip_grad_value_(grad_clip) is not None):
        clip_grad_value_(model.parameters(), grad_clip)
    if scaler is None:
        optimizer.step()
    else:
        # train.py:140: scaler.step(optimizer._optim).  The reference's wrapper never sees that call, so its Noam counter
        # stands still in fp16 runs; here the schedule lives on the device inside FlatAdam.step and advances
        # with every update that is actually applied — the host mirror follows it: GradScaler.step calls flat.step only
        # when the un-scaled gradients are finite, so the mirror advances exactly when that call happened (a skipped update
        # must not move step_num / cur_lr, which checkpoints save and load_state_dict re-imposes).
        applied = []
        inner_step = flat.step
        flat.step = lambda *a, **k: (applied.append(1), inner_step(*a, **k))[1]
        try:
            scaler.step(flat)
        finally:
            del flat.step                   # drop the instance attribute: the class's method is visible again
        scaler.update()
        if applied and hasattr(optimizer, "_update_learning_rate"):
            optimizer._update_learning_rate()
    return loss


class GraphedTrainStep:
    """The whole training step (zero_grad .. Adam/Noam) captured once into a hipGraph and replayed per batch.

    Every kernel on the path is asynchronous, allocation-free and reads its step-dependent scalars (Adam step, Noam
    learning rate) from device memory, so one captured graph is valid for every later step; replay removes the host
    launch cost of the ~1 400 kernels of a step.  Batches must keep the shapes of `example_batch` (the reference's
    collate pads to the longest utterance of each batch, so a production loop keeps one graph per padded shape).
    """

    def __init__(self, model, optimizer, grad_clip: float, example_batch, warmup: int = 2):
        self.model, self.optimizer, self.grad_clip = model, optimizer, grad_clip
        self.static = tuple(None if t is None else t.clone() for t in example_batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up on a side stream, as graph capture requires
            for _ in range(warmup):
                train_batch(model, optimizer, self.static, grad_clip)
        torch.cuda.current_stream().wait_stream(side)
        host_state = (optimizer.step_num, optimizer.cur_lr)   # capture runs the Python of a step but no kernel
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = train_batch(model, optimizer, self.static, grad_clip)
        optimizer.step_num, optimizer.cur_lr = host_state

    def __call__(self, batch=None) -> torch.Tensor:
        if batch is not None:
            for dst, src in zip(self.static, batch):
                if dst is not None and src is not dst:
                    dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.optimizer._update_learning_rate()               # host mirror of the on-device schedule
        return self.loss


def train_step(global_step: int, epoch: int, model, optimizer, config, train_loader, fp16_run: bool = False,
               scaler=None, reducer=None, on_loss: typing.Optional[typing.Callable] = None) -> int:
    """Same signature and return value as the reference's `train_step` (train.py:91-100).

    `fp16_run` (reference train.py:116-121, 133-141: `autocast()` + GradScaler) selects the reduced-precision form of THIS
    build: the flow decoder keeps its activation tensors in HBM as bf16 (`decoder.io_bf16 = "all"`, models.FlowSpecDecoder)
    with fp32 parameters, log-determinants and accumulation, and the text encoder's attention contractions run on the bf16
    matrix pipe (`MultiHeadAttention.bf16_mma`).  bf16 has fp32's exponent range, so no loss scaling is needed:
    `scaler` may be None; a GradScaler that is passed in is driven exactly as the reference drives it."""
    from .dataset import DeviceBatches

    model.train()
    bare = model.module if hasattr(model, "module") else model
    decoder = getattr(bare, "decoder", None)
    before = getattr(decoder, "io_bf16", False)
    if fp16_run and decoder is not None and not before:
        decoder.io_bf16 = "all"
    from .attentions import MultiHeadAttention
    mha = [m for m in bare.modules() if isinstance(m, MultiHeadAttention)] if fp16_run else []
    mha_before = [m.bf16_mma for m in mha]
    for m in mha:
        m.bf16_mma = True
    losses = []
    device = next(model.parameters()).device
    try:
        for batch in DeviceBatches(train_loader, device):      # batch k+1 is copied to HBM while step k runs
            losses.append(train_batch(model, optimizer, batch, config.grad_clip, reducer, scaler if fp16_run else None))
            global_step += 1
    finally:
        if decoder is not None:
            decoder.io_bf16 = before
        for m, b in zip(mha, mha_before):
            m.bf16_mma = b
    if losses and on_loss is not None:
        on_loss(epoch, float(torch.stack(losses).mean()), global_step)   # ONE sync per epoch
    return global_step
