"""oracle/unet.py — TEST INFRASTRUCTURE.  CPU fp32 restatement of the reference 3D-UNet forward.

Functional PyTorch over a plain state_dict that uses the reference's parameter names; the topology is derived
from the constructor kwargs the way lvdm/modules/networks/openaimodel3d.py:399-565 lays the blocks out.  Each
function cites the reference lines it restates.  Nothing here is used by the product path.
"""
import math

import torch
import torch.nn.functional as F


def sinusoid(t, dim, max_period=10000):
    # utils_diffusion.py:8-28
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _mlp(sd, p, x):
    # nn.Sequential(linear, SiLU, linear): openaimodel3d.py:377-395
    return _lin(sd, p + ".2", F.silu(_lin(sd, p + ".0", x)))


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, **kw):
    w = sd[p + ".weight"]
    fn = {3: F.conv1d, 4: F.conv2d, 5: F.conv3d}[w.dim()]
    return fn(x, w, sd.get(p + ".bias"), **kw)


def attention_core(q, k, v, heads, head_chunk=None):
    """softmax(q k^T * d^-0.5) v per head — attention.py:101-125 (the einsum path, no mask / relative position)."""
    b, n, c = q.shape
    d = c // heads
    q = q.reshape(b, n, heads, d).transpose(1, 2)
    k = k.reshape(b, k.shape[1], heads, d).transpose(1, 2)
    v = v.reshape(b, v.shape[1], heads, d).transpose(1, 2)
    scale = d ** -0.5
    if head_chunk is None:
        out = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1) @ v
    else:   # same arithmetic, bounded memory (CPU baseline at large N)
        out = torch.empty_like(q)
        for b0 in range(0, b, head_chunk):
            s = torch.softmax(q[b0:b0 + head_chunk] @ k[b0:b0 + head_chunk].transpose(-1, -2) * scale, dim=-1)
            out[b0:b0 + head_chunk] = s @ v[b0:b0 + head_chunk]
    return out.transpose(1, 2).reshape(b, n, c)


def cross_attention(sd, p, x, context, heads, image_cross, head_chunk=None):
    """CrossAttention.forward, attention.py:81-144.  context None = self-attention."""
    q = _lin(sd, p + ".to_q", x)
    if context is None:
        k, v = _lin(sd, p + ".to_k", x), _lin(sd, p + ".to_v", x)
        out = attention_core(q, k, v, heads, head_chunk)
    else:
        text, img = context[:, :77], context[:, 77:]
        out = attention_core(q, _lin(sd, p + ".to_k", text), _lin(sd, p + ".to_v", text), heads, head_chunk)
        if image_cross:   # attention.py:128-142, image_cross_attention_scale = 1.0
            out = out + 1.0 * attention_core(q, _lin(sd, p + ".to_k_ip", img), _lin(sd, p + ".to_v_ip", img), heads,
                                             head_chunk)
    return _lin(sd, p + ".to_out.0", out)


def transformer_block(sd, p, x, context, heads, image_cross, head_chunk=None):
    """BasicTransformerBlock._forward, attention.py:392-400; FeedForward/GEGLU 579-606."""
    x = cross_attention(sd, p + ".attn1", F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]),
                        None, heads, False, head_chunk) + x
    x = cross_attention(sd, p + ".attn2", F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]),
                        context, heads, image_cross, head_chunk) + x
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    val, gate = _lin(sd, p + ".ff.net.0.proj", h).chunk(2, dim=-1)
    return _lin(sd, p + ".ff.net.2", val * F.gelu(gate)) + x


def spatial_transformer(sd, p, x, context, heads, depth, head_chunk=None):
    """SpatialTransformer.forward, attention.py:451-467 (use_linear True or False)."""
    f, c, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    linear = sd[p + ".proj_in.weight"].dim() == 2
    if not linear:
        x = _conv(sd, p + ".proj_in", x)
    x = x.permute(0, 2, 3, 1).reshape(f, h * w, -1)
    if linear:
        x = _lin(sd, p + ".proj_in", x)
    for d in range(depth):
        x = transformer_block(sd, f"{p}.transformer_blocks.{d}", x, context, heads, True, head_chunk)
    if linear:
        x = _lin(sd, p + ".proj_out", x)
    x = x.reshape(f, h, w, -1).permute(0, 3, 1, 2)
    if not linear:
        x = _conv(sd, p + ".proj_out", x)
    return x + x_in


def temporal_transformer(sd, p, x, heads, depth):
    """TemporalTransformer.forward with only_self_att, no mask, attention.py:529-576.  x is (b, c, t, h, w)."""
    b, c, t, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
    linear = sd[p + ".proj_in.weight"].dim() == 2
    if not linear:
        x = _conv(sd, p + ".proj_in", x)
    x = x.transpose(1, 2)
    if linear:
        x = _lin(sd, p + ".proj_in", x)
    for d in range(depth):
        x = transformer_block(sd, f"{p}.transformer_blocks.{d}", x, None, heads, False)
    if linear:
        x = _lin(sd, p + ".proj_out", x)
        x = x.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2)
    else:
        x = _conv(sd, p + ".proj_out", x.transpose(1, 2))
        x = x.reshape(b, h, w, c, t).permute(0, 3, 4, 1, 2)
    return x + x_in


def temporal_conv_block(sd, p, x):
    """TemporalConvBlock.forward, openaimodel3d.py:272-279: GroupNorm statistics span (C/32, T, H, W)."""
    y = x
    for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        y = _conv(sd, f"{p}.{name}.{ci}", F.silu(_gn(sd, f"{p}.{name}.0", y, 1e-5)), padding=(1, 0, 0))
    return x + y


def res_block(sd, p, x, emb, batch):
    """ResBlock._forward (no up/down, no scale-shift), openaimodel3d.py:210-236."""
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), padding=1)
    h = h + _lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), padding=1)
    if p + ".skip_connection.weight" in sd:
        x = _conv(sd, p + ".skip_connection", x)
    h = x + h
    if p + ".temopral_conv.conv1.0.weight" in sd:   # sic: the reference's attribute name
        f, c, hh, ww = h.shape
        h5 = h.reshape(batch, f // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(f, c, hh, ww)
    return h


def topology(cfg):
    """Block list per UNetModel.__init__ (openaimodel3d.py:399-565): for every input / middle / output block a list
    of (kind, channels, heads) in module order.  kind in {conv, res, st, tt, down, up}."""
    mc, mult, nres = cfg["model_channels"], list(cfg["channel_mult"]), cfg["num_res_blocks"]
    ares, hc = set(cfg["attention_resolutions"]), cfg["num_head_channels"]
    temporal = cfg.get("temporal_attention", True)
    inputs = [[("conv", mc, 0)]]
    chans, ch, ds = [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            blk = [("res", m * mc, 0)]
            ch = m * mc
            if ds in ares:
                blk.append(("st", ch, ch // hc))
                if temporal:
                    blk.append(("tt", ch, ch // hc))
            inputs.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("down", ch, 0)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, 0), ("st", ch, ch // hc)] + ([("tt", ch, ch // hc)] if temporal else []) + [("res", ch, 0)]
    outputs = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            chans.pop()
            blk = [("res", m * mc, 0)]
            ch = m * mc
            if ds in ares:
                blk.append(("st", ch, ch // hc))
                if temporal:
                    blk.append(("tt", ch, ch // hc))
            if level and i == nres:
                blk.append(("up", ch, 0))
                ds //= 2
            outputs.append(blk)
    return inputs, middle, outputs


def _run_block(sd, p, blk, h, emb, context, batch, depth, head_chunk):
    for i, (kind, ch, heads) in enumerate(blk):
        q = f"{p}.{i}"
        if kind == "conv":
            h = _conv(sd, q, h, padding=1)
        elif kind == "res":
            h = res_block(sd, q, h, emb, batch)
        elif kind == "st":
            h = spatial_transformer(sd, q, h, context, heads, depth, head_chunk)
        elif kind == "tt":
            f, c, hh, ww = h.shape
            h5 = h.reshape(batch, f // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, q, h5, heads, depth)
            h = h5.permute(0, 2, 1, 3, 4).reshape(f, c, hh, ww)
        elif kind == "down":
            h = _conv(sd, q + ".op", h, stride=2, padding=1)          # Downsample, openaimodel3d.py:66-77
        elif kind == "up":
            h = _conv(sd, q + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"), padding=1)   # :98-106
    return h


@torch.no_grad()
def unet_forward(sd, cfg, x, timesteps, c_label, context, fs=None, head_chunk=None):
    """UNetModel.forward, openaimodel3d.py:567-628.  x (B, C, T, H, W); returns (B, out_channels, T, H, W)."""
    b, _, t, _, _ = x.shape
    mc = cfg["model_channels"]
    depth = cfg.get("transformer_depth", 1)
    emb = _mlp(sd, "time_embed", sinusoid(timesteps, mc))
    if cfg.get("class_label_condition", False):
        emb = emb + _mlp(sd, "class_embed", sinusoid(c_label, mc))
    if context.shape[1] == 77 + t * 16:
        text = context[:, :77].repeat_interleave(t, dim=0)
        img = context[:, 77:].reshape(b * t, 16, -1)
        context = torch.cat([text, img], dim=1)
    else:
        context = context.repeat_interleave(t, dim=0)
    emb = emb.repeat_interleave(t, dim=0)
    if cfg.get("fs_condition", False):
        if fs is None:
            fs = torch.tensor([cfg.get("default_fs", 4)] * b, dtype=torch.long)
        emb = emb + _mlp(sd, "fps_embedding", sinusoid(fs, mc)).repeat_interleave(t, dim=0)

    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4]).float()
    inputs, middle, outputs = topology(cfg)
    hs = []
    for i, blk in enumerate(inputs):
        h = _run_block(sd, f"input_blocks.{i}", blk, h, emb, context, b, depth, head_chunk)
        if i == 0 and cfg.get("addition_attention", False):
            f, c, hh, ww = h.shape
            h5 = h.reshape(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, "init_attn.0", h5, 8, depth)        # n_heads=8, openaimodel3d.py:404-414
            h = h5.permute(0, 2, 1, 3, 4).reshape(f, c, hh, ww)
        hs.append(h)
    h = _run_block(sd, "middle_block", middle, h, emb, context, b, depth, head_chunk)
    for i, blk in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"output_blocks.{i}", blk, h, emb, context, b, depth, head_chunk)
    y = _conv(sd, "out.2", F.silu(_gn(sd, "out.0", h, 1e-5)), padding=1)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
