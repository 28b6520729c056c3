"""Synthetic weights and inputs for benchmarks and smoke runs (no checkpoints or datasets are reachable offline).

Every tensor is drawn from a seeded generator ON the target device: matrices/filters N(0, 0.02^2) — including the
ones the reference zero-initialises, otherwise the UNet output is identically zero — norm scales 1 + 0.02 N,
biases / norm shifts 0.02 N.
"""
import torch


def materialize(module, device, seed=123):
    """Instantiate a module built under torch.device('meta') on `device` with synthetic parameters."""
    module.to_empty(device=device)
    fill_synthetic(module, seed)
    return module


@torch.no_grad()
def fill_synthetic(module, seed=123):
    gens = {}
    for i, (name, p) in enumerate(sorted(module.named_parameters(), key=lambda kv: kv[0])):
        g = gens.setdefault(p.device, torch.Generator(device=p.device))
        g.manual_seed(seed * 1000003 + i)
        noise = torch.empty(p.shape, dtype=torch.float32, device=p.device).normal_(0.0, 0.02, generator=g)
        if p.dim() <= 1 and name.endswith("weight"):
            noise += 1.0
        p.copy_(noise.to(p.dtype))
    return module


def rebuild_buffers(model, fresh):
    """Buffers (schedules) are deterministic functions of the config: copy them from a CPU-built twin."""
    src = dict(fresh.named_buffers())
    for name, buf in model.named_buffers():
        buf.copy_(src[name].to(buf.device))
