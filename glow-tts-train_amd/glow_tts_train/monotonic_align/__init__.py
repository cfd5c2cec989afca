"""Monotonic alignment search on the GPU (reference: glow_tts_train/monotonic_align/__init__.py + core.pyx).

`maximum_path(value, mask)` keeps the reference signature; there is no D2H copy, no host sync and no CPU kernel:
the lattice stays in HBM and one workgroup per utterance runs the dynamic programme (csrc/mas.hip).
"""
import torch

from .. import ops


def maximum_path(value, mask):
    """value, mask: [b, t_x, t_y] -> 0/1 path of value's dtype on value's device, bit-exact with the reference's
    Cython kernel for the same fp32 `value`."""
    mask = mask.to(value.dtype)
    # the reference reads the lengths off the first column / row of the mask (__init__.py:18-19)
    t_x = mask[:, :, 0].sum(1).to(torch.int32)
    t_y = mask[:, 0, :].sum(1).to(torch.int32)
    # only in-band cells (mask == 1) are read by the kernel, so `value * mask` (__init__.py:11) is not needed
    return maximum_path_lengths(value, t_x, t_y)


def maximum_path_lengths(value, t_x, t_y):
    """Same search with the valid lengths given directly (what FlowGenerator already has)."""
    path = ops.mas_path(value.float(), t_x, t_y)
    return path.to(value.dtype)
