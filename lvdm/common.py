"""Small helpers with the reference's names (lvdm/common.py:25-43,81-93)."""
from inspect import isfunction

import torch


def exists(val):
    return val is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def extract_into_tensor(a, t, x_shape):
    """a[t] reshaped to broadcast over x_shape (reference common.py:25-28)."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def noise_like(shape, device, repeat=False):
    """Gaussian noise from the device generator; repeat=True shares one draw across the batch (common.py:31-34)."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


def checkpoint(func, inputs, params, flag):
    """Activation checkpointing only matters for backward; the HIP path is inference-only, so this is a call-through
    (the reference's driver forces use_checkpoint=False as well: virtual_pose_render.py:156)."""
    return func(*inputs)
