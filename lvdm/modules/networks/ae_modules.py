"""AutoencoderKL building blocks at the drop-in boundary (reference: lvdm/modules/networks/ae_modules.py —
ResnetBlock 151-210, AttnBlock 26-78, Upsample 111-127, Downsample 90-109, Encoder 364-463, Decoder 466-578).
Parameter containers with the reference's attribute names; Decoder.forward runs on mudg_amd.engine.vae.
"""
import numpy as np
import torch.nn as nn

from lvdm.basics import Conv2d, GroupNorm, Linear


def Normalize(in_channels, num_groups=32):
    return GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1)


def make_attn(in_channels, attn_type="vanilla"):
    if attn_type == "vanilla":
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError(f"attention type '{attn_type}' is not on the MuDG path")


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("pooling downsample is not on the MuDG path")
        self.with_conv, self.in_channels = True, in_channels
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("bare nearest upsample is not on the MuDG path")
        self.with_conv, self.in_channels = True, in_channels
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        if conv_shortcut:
            raise NotImplementedError("3x3 conv shortcuts are not on the MuDG path")
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, False
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


def _level(block_in, block_out, n_blocks, res, attn_resolutions, attn_type, dropout):
    blocks, attns = nn.ModuleList(), nn.ModuleList()
    for _ in range(n_blocks):
        blocks.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
        block_in = block_out
        if res in attn_resolutions:
            attns.append(make_attn(block_in, attn_type=attn_type))
    holder = nn.Module()
    holder.block, holder.attn = blocks, attns
    return holder, block_in


def _middle(ch, attn_type, dropout):
    mid = nn.Module()
    mid.block_1 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0, dropout=dropout)
    mid.attn_1 = make_attn(ch, attn_type=attn_type)
    mid.block_2 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0, dropout=dropout)
    return mid


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.conv_in = Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        widths = (1,) + tuple(ch_mult)
        self.in_ch_mult = widths
        res, self.down = resolution, nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            holder, block_in = _level(ch * widths[lvl], ch * ch_mult[lvl], num_res_blocks, res, attn_resolutions,
                                      attn_type, dropout)
            if lvl != self.num_resolutions - 1:
                holder.downsample = Downsample(block_in, resamp_with_conv)
                res //= 2
            self.down.append(holder)
        self.mid = _middle(block_in, attn_type, dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        raise RuntimeError("Encoder runs inside AutoencoderKL.encode (mudg_amd.engine.vae.encode_moments)")


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not on the MuDG path")
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.give_pre_end, self.tanh_out = False, False
        block_in = ch * ch_mult[-1]
        res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, res, res)
        print("AE working on z of shape {} = {} dimensions.".format(self.z_shape, np.prod(self.z_shape)))
        self.conv_in = Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = _middle(block_in, attn_type, dropout)
        levels = []
        for lvl in reversed(range(self.num_resolutions)):
            holder, block_in = _level(block_in, ch * ch_mult[lvl], num_res_blocks + 1, res, attn_resolutions,
                                      attn_type, dropout)
            if lvl != 0:
                holder.upsample = Upsample(block_in, resamp_with_conv)
                res *= 2
            levels.insert(0, holder)
        self.up = nn.ModuleList(levels)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        from mudg_amd.engine import vae
        return vae.decoder_forward(self, z)
