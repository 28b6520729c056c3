"""Parameter containers with the reference's factory names (lvdm/basics.py:19-87).

The layers below only OWN parameters (so state_dicts, optimizers and the training-time surgery in
main/utils_train.py keep working); arithmetic happens in mudg_amd's kernels, which read `.weight/.bias` directly.
Calling one of them eagerly is a bug on this path, so `forward` refuses instead of silently running ATen.
"""
import torch.nn as nn

from utils.utils import instantiate_from_config  # noqa: F401  (re-exported like the reference)


def _refuse(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter container: the MI355X path runs it inside a fused HIP "
                       "kernel (mudg_amd.engine); there is no eager fallback")


class Conv1d(nn.Conv1d):
    forward = _refuse


class Conv2d(nn.Conv2d):
    forward = _refuse


class Conv3d(nn.Conv3d):
    forward = _refuse


class Linear(nn.Linear):
    forward = _refuse


class GroupNorm(nn.GroupNorm):
    forward = _refuse


class LayerNorm(nn.LayerNorm):
    forward = _refuse


class GroupNormSpecific(GroupNorm):
    """GroupNorm whose statistics are always fp32 (reference basics.py:76-78) — the kernels always do that."""


def disabled_train(self, mode=True):
    return self


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def scale_module(module, scale):
    for p in module.parameters():
        p.detach().mul_(scale)
    return module


def conv_nd(dims, *args, **kwargs):
    try:
        return {1: Conv1d, 2: Conv2d, 3: Conv3d}[dims](*args, **kwargs)
    except KeyError:
        raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return Linear(*args, **kwargs)


def avg_pool_nd(dims, *args, **kwargs):
    raise NotImplementedError("average-pool resampling (conv_resample=False) is not on the MuDG path")


def normalization(channels, num_groups=32):
    return GroupNormSpecific(num_groups, channels)
