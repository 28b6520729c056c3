"""Diagonal Gaussian posterior of the VAE encoder (reference: lvdm/distributions.py:24-65).

`sample` keeps the reference's RNG contract — the noise is drawn with torch.randn ON THE CPU from the global
generator (distributions.py:37), then moved to the device — so seeded runs consume the host generator identically;
the arithmetic (clamp, exp, scale-and-add) is one HIP launch."""
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        if deterministic:
            raise NotImplementedError("deterministic posteriors are not on the MuDG path")
        self.parameters = parameters
        self.deterministic = False

    @property
    def mean(self):
        return torch.chunk(self.parameters, 2, dim=1)[0]

    @property
    def logvar(self):
        return torch.chunk(self.parameters, 2, dim=1)[1]

    def mode(self):
        from mudg_amd import ops
        return ops.gaussian_sample(self.parameters, None, 1.0)

    def sample(self, noise=None, scale=1.0):
        from mudg_amd import ops
        if noise is None:
            n, c2, h, w = self.parameters.shape
            noise = torch.randn((n, c2 // 2, h, w))
        return ops.gaussian_sample(self.parameters, noise, scale)
