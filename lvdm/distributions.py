"""Diagonal Gaussian posterior of the VAE encoder (reference: lvdm/distributions.py:24-65) — boundary type only;
the encoder is a "next" row of the scope table, so sampling from it is not yet on the HIP path."""
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.deterministic = deterministic

    def mode(self):
        return self.mean

    def sample(self, noise=None):
        raise NotImplementedError("posterior sampling belongs to the VAE-encode row (SURVEY §8(f) rank 1)")
