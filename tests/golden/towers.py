"""Deterministic stand-ins for the CLIP towers (outside the path), shared by make_golden.py (which hangs
them on the REFERENCE model) and by the tests (which hang them on this repo's model): the same seeded tensors come out
on both sides, so everything downstream of the towers — Resampler, VAE encodes, dict assembly, sampler, decode — is
compared for real.  Pure harness code; no reference source."""
import torch


class FakeImageTower(torch.nn.Module):
    """model.embedder: (b, 3, h, w) image -> (b, tokens, dim); an all-zero image (the unconditional branch,
    virtual_pose_render.py:96) gets a different draw than a real one."""

    def __init__(self, tokens, dim, seed):
        super().__init__()
        self.tokens, self.dim, self.seed = tokens, dim, seed

    def forward(self, img):
        salt = int(float(img.detach().abs().sum()) > 0)
        g = torch.Generator().manual_seed(self.seed + salt)
        return torch.randn(img.shape[0], self.tokens, self.dim, generator=g).to(img.device)


class FakeTextTower(torch.nn.Module):
    """model.cond_stage_model: list of prompts -> (b, 77, dim); the empty prompt gets its own draw."""

    def __init__(self, dim, seed, device="cpu"):
        super().__init__()
        self.dim, self.seed, self.dev = dim, seed, device

    def encode(self, prompts):
        salt = int(sum(len(p) for p in prompts) > 0)
        g = torch.Generator().manual_seed(self.seed + salt)
        return torch.randn(len(prompts), 77, self.dim, generator=g).to(self.dev)

    def forward(self, prompts):
        return self.encode(prompts)
