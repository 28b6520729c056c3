"""Deterministic synthetic weights shared by the golden generator (which runs against the reference) and the tests
(which run against oracle/ and the HIP path).  Weights are never stored: both sides rebuild them from
(sorted parameter names, shapes, seed) and compare a checksum.

Zero-initialised reference parameters (out conv, proj_out, conv4, fps_embedding[-1], ...) are re-randomised too,
otherwise the UNet output is identically zero.
"""
import hashlib

import torch


def _key_seed(seed, name):
    return int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:7], "little")


def seeded_tensor(name, shape, seed):
    g = torch.Generator().manual_seed(_key_seed(seed, name))
    shape = tuple(shape)
    if len(shape) <= 1:
        t = torch.randn(shape, generator=g)
        if name.endswith(".weight"):        # norm scales
            return 1.0 + 0.1 * t
        return 0.05 * t                     # biases
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) / fan_in ** 0.5


def seeded_state_dict(shapes, seed):
    """shapes: {name: shape}.  Returns {name: fp32 tensor} in sorted-name order."""
    return {k: seeded_tensor(k, shapes[k], seed) for k in sorted(shapes)}


def checksum(sd):
    tot = 0.0
    for k in sorted(sd):
        if sd[k].dtype.is_floating_point:
            tot += float(sd[k].double().abs().sum())
    return tot


def seeded_input(name, shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(_key_seed(seed, "input:" + name))
    return torch.randn(tuple(shape), generator=g) * scale
