"""Perceiver Resampler at the drop-in boundary (reference: lvdm/modules/encoders/resampler.py, Resampler 96-145,
PerceiverAttention 48-93, FeedForward 27-34): turns the 257 x 1280 CLIP image tokens of the conditioning frame into
16 context tokens per video frame (video_length x num_queries x output_dim).  Runs once per clip; same state_dict keys
(`latents`, `proj_in`, `layers.{i}.0.*`, `layers.{i}.1.{0,1,3}`, `proj_out`, `norm_out`).  Arithmetic: mudg_amd.engine.
resampler on the GEMM / LayerNorm / flash-attention kernels."""
import torch
import torch.nn as nn

from lvdm.basics import LayerNorm, Linear


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(LayerNorm(dim), Linear(dim, inner, bias=False), nn.GELU(), Linear(inner, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("the MI355X attention kernel is specialised for head dim 64")
        self.scale, self.dim_head, self.heads = dim_head ** -0.5, dim_head, heads
        inner = dim_head * heads
        self.norm1, self.norm2 = LayerNorm(dim), LayerNorm(dim)
        self.to_q = Linear(dim, inner, bias=False)
        self.to_kv = Linear(dim, inner * 2, bias=False)
        self.to_out = Linear(inner, dim, bias=False)


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, video_length=None):
        super().__init__()
        self.num_queries, self.video_length = num_queries, video_length
        total = num_queries * video_length if video_length is not None else num_queries
        self.latents = nn.Parameter(torch.randn(1, total, dim) / dim ** 0.5)
        self.proj_in = Linear(embedding_dim, dim)
        self.proj_out = Linear(dim, output_dim)
        self.norm_out = LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])

    def forward(self, x):
        """x (B, n_tokens, embedding_dim) image tokens -> (B, total_queries, output_dim), fp32.  With gradients enabled and a
        parameter (or the input) that wants one, the forward runs on the autograd Functions of mudg_amd.train (the MuDG training
        configs train this module: image_proj_model_trainable); otherwise on the fused inference executor."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())) and self.training:
            from mudg_amd.train import resampler as train_resampler
            return train_resampler.forward(self, x)
        from mudg_amd.engine import resampler
        return resampler.forward(self, x)
