"""Resampler.forward for TRAINING (reference: lvdm/modules/encoders/resampler.py:48-145; the MuDG training configs set
image_proj_model_trainable, ddpm3d.py:1281-1284, so the 48.8 M parameters of the image-token Perceiver receive gradients).

The same walk as mudg_amd/engine/resampler.py with every layer expressed through the autograd Functions of functions.py —
Linear (GEMM + weight-gradient kernels), LayerNorm, the flash attention pair, the exact-erf GELU kernel — so that a loss on the
context tokens fills `.grad` of the reference-named parameters (`latents`, `proj_in`, `layers.{i}.0.{norm1,norm2,to_q,to_kv,
to_out}`, `layers.{i}.1.{0,1,3}`, `proj_out`, `norm_out`).  Rows are (batch, token) with channels contiguous, fp32 between layers."""
import torch

from . import functions as F_
from . import kernels as K


class Gelu(torch.autograd.Function):
    """Exact GELU (nn.GELU() of the Perceiver feed-forward, resampler.py:27-34)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.gelu(x, dy)


def _lin(mod, x, residual=None):
    return F_.Linear.apply(x, mod.weight, mod.bias, residual)


def _ln(mod, x):
    return F_.LayerNorm.apply(x, mod.weight, mod.bias, mod.eps)


def forward(mod, x):
    """x (B, n_tokens, embedding_dim) -> (B, total_queries, output_dim), fp32, with an autograd graph."""
    if not x.is_cuda:
        raise RuntimeError("Resampler: inputs must be on the GPU; the MI355X path has no CPU fallback")
    b, n1, e = x.shape
    n2, dim = mod.latents.shape[1], mod.latents.shape[2]
    xs = _lin(mod.proj_in, x.reshape(b * n1, e).float().contiguous())                    # (b n1, dim)
    lat = mod.latents.float().repeat(b, 1, 1).reshape(b * n2, dim)                       # the learned queries, once per batch entry
    nk = n1 + n2
    for attn, ff in mod.layers:
        xn = _ln(attn.norm1, xs)
        ln = _ln(attn.norm2, lat)
        # keys / values see [image tokens ; latents] of their batch entry (resampler.py:72-73)
        kv_in = torch.cat((xn.reshape(b, n1, dim), ln.reshape(b, n2, dim)), dim=1).reshape(b * nk, dim)
        q = _lin(attn.to_q, ln)
        kv = _lin(attn.to_kv, kv_in)
        inner = kv.shape[1] // 2
        # q and k are each scaled by dim_head^-1/4 in the reference: 1 / sqrt(dim_head) on the logits
        att = F_.Attention.apply(q, kv[:, :inner].contiguous(), kv[:, inner:].contiguous(), None, None,
                                 (b, attn.heads, n2, nk, 1, 0, 1, attn.scale))
        lat = _lin(attn.to_out, att, residual=lat)
        hid = Gelu.apply(_lin(ff[1], _ln(ff[0], lat)))
        lat = _lin(ff[3], hid, residual=lat)
    out = _ln(mod.norm_out, _lin(mod.proj_out, lat))
    return out.reshape(b, n2, -1)
