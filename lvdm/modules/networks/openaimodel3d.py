"""The 3D (2D-spatial + 1D-temporal) UNet denoiser at the drop-in boundary.

Reference: lvdm/modules/networks/openaimodel3d.py — UNetModel 281-628, ResBlock 109-236, TemporalConvBlock 239-279,
Downsample 51-77, Upsample 80-106, TimestepEmbedSequential 30-48.  Constructor kwargs, attribute names (including
the reference's `temopral_conv` spelling) and the module tree are kept so that the published checkpoints load with
strict=True and the training-time surgery of main/utils_train.py (assigning new nn.Conv2d / nn.Linear into the tree)
still works.  `forward` hands the whole network to mudg_amd.engine.unet, which executes it on channels-last bf16
rows with hand-written gfx950 kernels; nothing here runs eagerly.
"""
import torch
import torch.nn as nn

from lvdm.basics import avg_pool_nd, conv_nd, linear, normalization, zero_module, Conv3d, GroupNorm  # noqa: F401
from lvdm.modules.attention import SpatialTransformer, TemporalTransformer


class TimestepBlock(nn.Module):
    """Marker base: blocks that consume the timestep embedding."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Ordered container of one UNet stage (ResBlock / SpatialTransformer / TemporalTransformer / resampler)."""

    def forward(self, x, emb, context=None, batch_size=None):
        raise RuntimeError("UNet stages are scheduled by mudg_amd.engine.unet, not run one by one")


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError("only the strided 3x3 conv downsampler (conv_resample=True, dims=2) is supported")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, True, dims
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        if not use_conv or dims != 2:
            raise NotImplementedError("only nearest-2x + 3x3 conv upsampling (conv_resample=True, dims=2) is supported")
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, True, dims
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)


class TemporalConvBlock(nn.Module):
    """Four [GroupNorm(32) + SiLU + (3,1,1) conv] stages with an identity shortcut; the last conv starts at zero."""

    def __init__(self, in_channels, out_channels=None, dropout=0.0, spatial_aware=False):
        super().__init__()
        if spatial_aware:
            raise NotImplementedError("tempspatial_aware temporal convolutions are not on the MuDG path")
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels

        def stage(cin, cout, with_dropout):
            mods = [GroupNorm(32, cin), nn.SiLU()] + ([nn.Dropout(dropout)] if with_dropout else [])
            return nn.Sequential(*mods, Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0)))

        self.conv1 = stage(in_channels, out_channels, False)
        self.conv2 = stage(out_channels, in_channels, True)
        self.conv3 = stage(out_channels, in_channels, True)
        self.conv4 = stage(out_channels, in_channels, True)
        zero_module(self.conv4[-1])


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, use_conv=False, up=False, down=False, use_temporal_conv=False,
                 tempspatial_aware=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv or dims != 2:
            raise NotImplementedError("scale-shift norm / resblock resampling / 3x3 skip convs are not on the MuDG path")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_conv, self.use_checkpoint, self.use_scale_shift_norm = False, use_checkpoint, False
        self.use_temporal_conv, self.updown = use_temporal_conv, False
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels, self.out_channels, dropout=0.1,
                                                   spatial_aware=tempspatial_aware)

    def forward(self, x, emb, batch_size=None):
        from mudg_amd.engine import standalone
        return standalone.res_block(self, x, emb, batch_size)


def _stage_plan(model_channels, channel_mult, num_res_blocks, attention_resolutions):
    """Yield ('res', cin, cout, attn) / ('down', c) / ('up', c) in construction order for encoder and decoder."""
    enc, skips, ch, ds = [], [model_channels], model_channels, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            enc.append(("res", ch, mult * model_channels, ds in attention_resolutions))
            ch = mult * model_channels
            skips.append(ch)
        if level != len(channel_mult) - 1:
            enc.append(("down", ch))
            skips.append(ch)
            ds *= 2
    mid_ch = ch
    dec = []
    for level, mult in reversed(list(enumerate(channel_mult))):
        for i in range(num_res_blocks + 1):
            dec.append(("res", ch + skips.pop(), mult * model_channels, ds in attention_resolutions,
                        bool(level) and i == num_res_blocks))
            ch = mult * model_channels
            if level and i == num_res_blocks:
                ds //= 2
    return enc, mid_ch, dec, ch


class UNetModel(nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0.0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None, use_scale_shift_norm=False,
                 resblock_updown=False, num_heads=-1, num_head_channels=-1, transformer_depth=1, use_linear=False,
                 use_checkpoint=False, temporal_conv=False, tempspatial_aware=False, temporal_attention=True,
                 use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False,
                 class_label_condition=False, domain_cross_attention=False, num_tasks=1, temporal_frozen=False):
        super().__init__()
        if num_heads == -1 and num_head_channels == -1:
            raise AssertionError("Either num_heads or num_head_channels has to be set")
        if resblock_updown or not conv_resample or dims != 2:
            raise NotImplementedError("resblock_updown / pooling resamplers / dims != 2 are not on the MuDG path")
        attention_resolutions = list(attention_resolutions)
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.dropout, self.channel_mult, self.conv_resample = dropout, channel_mult, conv_resample
        self.temporal_attention, self.use_checkpoint = temporal_attention, use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.addition_attention, self.temporal_length = addition_attention, temporal_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale_learnable = image_cross_attention_scale_learnable
        self.default_fs, self.fs_condition = default_fs, fs_condition
        self.class_label_condition, self.domain_cross_attention, self.num_tasks = class_label_condition, domain_cross_attention, num_tasks
        self.transformer_depth, self.context_dim = transformer_depth, context_dim
        emb_dim = model_channels * 4

        def embed_mlp():
            return nn.Sequential(linear(model_channels, emb_dim), nn.SiLU(), linear(emb_dim, emb_dim))

        self.time_embed = embed_mlp()
        if class_label_condition:
            self.class_embed = embed_mlp()
        if fs_condition:
            self.fps_embedding = embed_mlp()
            zero_module(self.fps_embedding[-1])

        def heads_of(ch):
            return (ch // num_head_channels, num_head_channels) if num_head_channels != -1 else (num_heads, ch // num_heads)

        def res(cin, cout):
            return ResBlock(cin, emb_dim, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, tempspatial_aware=tempspatial_aware,
                            use_temporal_conv=temporal_conv)

        def attn_pair(ch, with_tasks):
            nh, dh = heads_of(ch)
            extra = dict(domain_cross_attention=domain_cross_attention, num_tasks=num_tasks) if with_tasks else {}
            mods = [SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                                       use_linear=use_linear, use_checkpoint=use_checkpoint, disable_self_attn=False,
                                       video_length=temporal_length, image_cross_attention=image_cross_attention,
                                       image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                       **extra)]
            if temporal_attention:
                mods.append(TemporalTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                                                use_linear=use_linear, use_checkpoint=use_checkpoint,
                                                only_self_att=True, causal_attention=use_causal_attention,
                                                relative_position=use_relative_position,
                                                temporal_length=temporal_length, temporal_frozen=temporal_frozen))
            return mods

        enc, mid_ch, dec, last_ch = _stage_plan(model_channels, channel_mult, num_res_blocks, attention_resolutions)
        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(TemporalTransformer(
                model_channels, n_heads=8, d_head=num_head_channels, depth=transformer_depth, context_dim=context_dim,
                use_checkpoint=use_checkpoint, only_self_att=temporal_selfatt_only, causal_attention=False,
                relative_position=use_relative_position, temporal_length=temporal_length))
        for item in enc:
            if item[0] == "res":
                _, cin, cout, attn = item
                mods = [res(cin, cout)] + (attn_pair(cout, True) if attn else [])
                self.input_blocks.append(TimestepEmbedSequential(*mods))
            else:
                self.input_blocks.append(TimestepEmbedSequential(
                    Downsample(item[1], conv_resample, dims=dims, out_channels=item[1])))
        self.middle_block = TimestepEmbedSequential(res(mid_ch, mid_ch), *attn_pair(mid_ch, False), res(mid_ch, mid_ch))
        self.output_blocks = nn.ModuleList()
        for _, cin, cout, attn, up in dec:
            mods = [res(cin, cout)] + (attn_pair(cout, False) if attn else [])
            if up:
                mods.append(Upsample(cout, conv_resample, dims=dims, out_channels=cout))
            self.output_blocks.append(TimestepEmbedSequential(*mods))
        self.out = nn.Sequential(normalization(last_ch), nn.SiLU(),
                                 zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

    def forward(self, x, timesteps, c_label=None, context=None, features_adapter=None, fs=None, **kwargs):
        """x (B, in_channels, T, H, W) — or a list of tensors to be concatenated along channels — timesteps (B,) long,
        c_label (B,) long, context (B, 77 + 16 T, context_dim), fs (B,) long.  Extra kwargs the reference's callers
        pass (sparse_x, class_label, cfg_img, ...) are accepted and ignored, as in the reference.  Returns
        (B, out_channels, T, H, W) in x's dtype."""
        if self.training and torch.is_grad_enabled() and self._wants_grad(x, context):
            # training step: the same network on autograd Functions whose forward and backward are HIP kernels
            if features_adapter is not None:
                raise NotImplementedError("features_adapter is not used on the MuDG path")
            from mudg_amd.train import unet as train_engine
            return train_engine.forward(self, x, timesteps, c_label=c_label, context=context, fs=fs)
        from mudg_amd.engine import unet as engine
        return engine.forward_entry(self, x, timesteps, c_label=c_label, context=context,
                                    features_adapter=features_adapter, fs=fs)

    def _wants_grad(self, x, context):
        """A module is in training mode by default: only a caller that can actually receive a gradient (a trainable parameter, or an
        input / context that requires one) is sent down the training path; everybody else keeps the fused inference engine."""
        xs = x if isinstance(x, (list, tuple)) else [x]
        if any(isinstance(t, torch.Tensor) and t.requires_grad for t in (*xs, context)):
            return True
        return any(p.requires_grad for p in self.parameters())

    def prepare_context(self, context, temporal_length=None):
        """Not in the reference: make a (B, L, D) context ready once for many forward calls — operand rows plus every
        cross-attention layer's K / V^T projections of the (step-invariant) tokens.  Pass the result as `context=`."""
        from mudg_amd.engine import unet as engine
        with torch.no_grad():
            return engine.PreparedContext(self, context, temporal_length or self.temporal_length)
