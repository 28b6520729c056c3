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
    """common.py:81-93: evaluate func(*inputs) without keeping its intermediate activations when `flag` is set — the forward is
    replayed during backward.  Without gradients (sampling; the reference's driver forces use_checkpoint=False as well,
    virtual_pose_render.py:156) or with the flag off this is a plain call.  With both, torch.utils.checkpoint does the replay (it
    restores the RNG state, so dropout masks repeat); `params` needs no handling: the autograd Functions of mudg_amd.train hold
    the parameters through their graph edges."""
    if flag and torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in (*inputs, *params)):
        from torch.utils.checkpoint import checkpoint as _ckpt
        return _ckpt(func, *inputs, use_reentrant=False)
    return func(*inputs)


# ---- the remaining public helpers of the reference module (lvdm/common.py:8-78), so that MuDG modules which import
# them from here (lvdm/modules/encoders/condition.py:7 `from lvdm.common import autocast`) keep working when this
# package overlays the reference's.  They carry no hot-path arithmetic.
def gather_data(data, return_np=True):
    """all_gather of a tensor over the default process group (common.py:8-13)."""
    import torch.distributed as dist
    parts = [torch.zeros_like(data) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, data)
    return [p.cpu().numpy() for p in parts] if return_np else parts


def autocast(f):
    """Decorator: run f under torch.cuda.amp.autocast with the ambient autocast dtype / cache settings (common.py:16-22).
    The HIP kernels fix their own precision, so this only matters for third-party modules that use it (the CLIP towers)."""
    import functools

    @functools.wraps(f)
    def wrapped(*args, **kwargs):
        with torch.cuda.amp.autocast(enabled=True, dtype=torch.get_autocast_gpu_dtype(),
                                     cache_enabled=torch.is_autocast_cache_enabled()):
            return f(*args, **kwargs)
    return wrapped


def identity(*args, **kwargs):
    return torch.nn.Identity()


def uniq(arr):
    return {el: True for el in arr}.keys()


def mean_flat(tensor):
    """Mean over all non-batch dimensions (common.py:51-55); host-side loss bookkeeping in the reference."""
    return tensor.mean(dim=list(range(1, tensor.dim())))


def ismap(x):
    return isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[1] > 3


def isimage(x):
    return isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[1] in (1, 3)


def max_neg_value(t):
    return -torch.finfo(t.dtype).max


def shape_to_str(x):
    return "x".join(str(s) for s in x.shape)


def init_(tensor):
    import math
    std = 1.0 / math.sqrt(tensor.shape[-1])
    tensor.uniform_(-std, std)
    return tensor
