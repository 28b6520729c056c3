"""oracle/vae.py — TEST INFRASTRUCTURE.  CPU fp32 restatement of AutoencoderKL.decode.

Follows lvdm/models/autoencoder.py:104-107 and lvdm/modules/networks/ae_modules.py: Decoder.forward 539-578,
ResnetBlock 190-210 (temb=None), AttnBlock 52-78, Upsample 123-127, Normalize 15-16 (eps 1e-6), nonlinearity 10-12;
the per-frame decode loop and 1/scale_factor are ddpm3d.py:646-667.
"""
import torch
import torch.nn.functional as F


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(sd, p, x, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(sd, p, x):
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h)), 1)
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    h = _gn(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h), _conv(sd, p + ".k", h), _conv(sd, p + ".v", h)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h)


@torch.no_grad()
def encode(sd, ddconfig, x):
    """AutoencoderKL.encode up to the posterior parameters (autoencoder.py:97-102; Encoder.forward ae_modules.py:
    433-463; Downsample pads (0,1,0,1) then 3x3 stride 2, ae_modules.py:102-107).  Returns moments (N, 2*z, h, w)."""
    nlev, nres = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    h = _conv(sd, "encoder.conv_in", x, 1)
    for lvl in range(nlev):
        for i in range(nres):
            h = resnet_block(sd, f"encoder.down.{lvl}.block.{i}", h)
        if lvl != nlev - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"encoder.down.{lvl}.downsample.conv.weight"],
                         sd[f"encoder.down.{lvl}.downsample.conv.bias"], stride=2)
    h = resnet_block(sd, "encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", _swish(_gn(sd, "encoder.norm_out", h)), 1)
    return _conv(sd, "quant_conv", h)


def posterior_sample(moments, noise, scale_factor=1.0):
    """DiagonalGaussianDistribution.sample + get_first_stage_encoding (distributions.py:24-40, ddpm3d.py:611-618)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return scale_factor * (mean + std * noise)


@torch.no_grad()
def decode(sd, ddconfig, z, prefix=""):
    """AutoencoderKL.decode: post_quant_conv then Decoder.forward.  z (N, z_channels, h, w) already divided by
    scale_factor.  sd keys: post_quant_conv.*, decoder.* (optionally under `prefix`)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    nlev, nres = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    h = _conv(sd, "post_quant_conv", z)
    h = _conv(sd, "decoder.conv_in", h, 1)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for lvl in reversed(range(nlev)):
        for i in range(nres + 1):
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = _conv(sd, f"decoder.up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"), 1)
    return _conv(sd, "decoder.conv_out", _swish(_gn(sd, "decoder.norm_out", h)), 1)


@torch.no_grad()
def decode_first_stage(sd, ddconfig, z, scale_factor=0.18215, prefix=""):
    """LatentDiffusion.decode_core with perframe_ae: z (B, C, T, h, w) -> (B, 3, T, 8h, 8w)."""
    b, c, t, hh, ww = z.shape
    frames = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
    out = torch.cat([decode(sd, ddconfig, 1. / scale_factor * frames[i:i + 1], prefix) for i in range(b * t)], 0)
    return out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4)
