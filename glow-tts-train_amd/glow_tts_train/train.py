"""The inner training step (reference: glow_tts_train/train.py:91-162) without its host synchronisations.

The reference pays >= 521 device->host syncs per step (two `loss.item()` plus one `.item()` per parameter tensor in
`clip_grad_value_`); here the loss stays on device, clipping and Adam/Noam are one kernel each over flat buffers,
the alignment search never leaves the GPU, and (with `reducer`) gradient all-reduce overlaps the backward.
Checkpoint cadence, logging and dataset plumbing stay with the caller.
"""
from __future__ import annotations

import typing

import torch

from .utils import clip_grad_value_, duration_loss, mle_loss, to_gpu


def train_batch(model, optimizer, batch, grad_clip: float, reducer=None) -> torch.Tensor:
    """One optimisation step on one already-resident batch; returns the (device) loss tensor, un-synchronised."""
    x, x_lengths, y, y_lengths, speaker_ids = batch
    optimizer.zero_grad()
    (z, z_m, z_logs, logdet, z_mask), _, (_attn, logw, logw_) = model(x, x_lengths, y, y_lengths, g=speaker_ids)
    loss = mle_loss(z, z_m, z_logs, logdet, z_mask) + duration_loss(logw, logw_, x_lengths)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    clip_grad_value_(model.parameters(), grad_clip)
    optimizer.step()
    return loss.detach()


def train_step(global_step: int, epoch: int, model, optimizer, config, train_loader, fp16_run: bool = False,
               scaler=None, reducer=None, on_loss: typing.Optional[typing.Callable] = None) -> int:
    """Same signature and return value as the reference's `train_step` (train.py:91-100); fp32 only (the reference's
    fp16 branch bypasses the Noam schedule)."""
    if fp16_run:
        raise NotImplementedError("glow_tts_train (MI355X build): fp16_run is not implemented; the path is fp32")
    model.train()
    losses = []
    for batch in train_loader:
        x, x_lengths, y, y_lengths, speaker_ids = batch
        batch = (to_gpu(x), to_gpu(x_lengths), to_gpu(y), to_gpu(y_lengths),
                 None if speaker_ids is None else to_gpu(speaker_ids))
        losses.append(train_batch(model, optimizer, batch, config.grad_clip, reducer))
        global_step += 1
    if losses and on_loss is not None:
        on_loss(epoch, float(torch.stack(losses).mean()), global_step)   # ONE sync per epoch
    return global_step
