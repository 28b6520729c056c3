"""oracle/resampler.py — TEST INFRASTRUCTURE.  CPU fp32 restatement of the Perceiver Resampler
(lvdm/modules/encoders/resampler.py: Resampler.forward 133-145, PerceiverAttention.forward 64-93, FeedForward 27-34)."""
import math

import torch
import torch.nn.functional as F


def _ln(sd, p, x):
    return F.layer_norm(x, x.shape[-1:], sd[p + ".weight"], sd[p + ".bias"])


def _heads(x, heads):
    b, n, w = x.shape
    return x.view(b, n, heads, -1).transpose(1, 2)


@torch.no_grad()
def forward(sd, x, heads, depth):
    latents = sd["latents"].repeat(x.size(0), 1, 1)
    x = F.linear(x, sd["proj_in.weight"], sd["proj_in.bias"])
    for i in range(depth):
        p = f"layers.{i}.0"
        xn, ln = _ln(sd, p + ".norm1", x), _ln(sd, p + ".norm2", latents)
        b, l, _ = ln.shape
        q = F.linear(ln, sd[p + ".to_q.weight"])
        k, v = F.linear(torch.cat((xn, ln), dim=-2), sd[p + ".to_kv.weight"]).chunk(2, dim=-1)
        q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
        scale = 1 / math.sqrt(math.sqrt(q.shape[-1]))
        w = torch.softmax(((q * scale) @ (k * scale).transpose(-2, -1)).float(), dim=-1)
        out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
        latents = F.linear(out, sd[p + ".to_out.weight"]) + latents
        f = f"layers.{i}.1"
        h = F.linear(F.gelu(F.linear(_ln(sd, f + ".0", latents), sd[f + ".1.weight"])), sd[f + ".3.weight"])
        latents = h + latents
    latents = F.linear(latents, sd["proj_out.weight"], sd["proj_out.bias"])
    return _ln(sd, "norm_out", latents)
