"""Transformer blocks of the 3D-UNet at the drop-in boundary (reference: lvdm/modules/attention.py).

Same class names, constructor kwargs, sub-module names and therefore state_dict keys as the reference
(CrossAttention 42-79, BasicTransformerBlock 348-379, SpatialTransformer 403-448, TemporalTransformer 470-523,
GEGLU/FeedForward 579-603).  The classes hold parameters; their arithmetic is mudg_amd/engine/unet.py, which runs
whole transformer blocks on channels-last rows with the MFMA GEMM, flash attention and norm kernels.

Options the MuDG configs never enable are rejected at construction instead of being half-supported:
relative position tables, causal temporal masks, the learnable image-attention gate and domain (joint) attention.
"""
import torch.nn as nn

from lvdm.basics import GroupNorm, LayerNorm, Linear, Conv1d, Conv2d, zero_module
from lvdm.common import default

XFORMERS_IS_AVAILBLE = False   # (sic) kept for callers that read it; the HIP flash kernel replaces xformers


def _unsupported(flag, what):
    if flag:
        raise NotImplementedError(f"{what} is not used by the MuDG configurations and is not implemented on the "
                                  "MI355X path")


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0., relative_position=False,
                 temporal_length=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77):
        super().__init__()
        _unsupported(relative_position, "relative position attention")
        _unsupported(image_cross_attention_scale_learnable, "a learnable image cross-attention scale")
        # the attention kernels are written for 64-wide heads (what every MuDG config uses: num_head_channels 64, and
        # 8 x 64 in init_attn); any other width would build and then compute the wrong thing
        _unsupported(dim_head != 64, f"attention head width {dim_head} (only 64)")
        inner = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(context_dim, inner, bias=False)
        self.to_v = Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(Linear(inner, query_dim), nn.Dropout(dropout))
        self.relative_position = False
        self.temporal_length = temporal_length
        self.video_length = video_length
        self.image_cross_attention = image_cross_attention
        self.image_cross_attention_scale = image_cross_attention_scale
        self.image_cross_attention_scale_learnable = False
        self.text_context_len = text_context_len
        if image_cross_attention:
            self.to_k_ip = Linear(context_dim, inner, bias=False)
            self.to_v_ip = Linear(context_dim, inner, bias=False)

    def forward(self, x, context=None, mask=None):
        raise RuntimeError("CrossAttention runs fused inside its transformer block (mudg_amd.engine.unet)")


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward is on the MuDG path")
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), Linear(inner, default(dim_out, dim)))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attention_cls=None, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale=1.0, image_cross_attention_scale_learnable=False, text_context_len=77,
                 domain_cross_attention=False, num_tasks=1):
        super().__init__()
        _unsupported(domain_cross_attention, "domain (joint) cross-attention")
        _unsupported(disable_self_attn, "disable_self_attn")
        make = CrossAttention if attention_cls is None else attention_cls
        self.disable_self_attn = False
        self.attn1 = make(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = make(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                          video_length=video_length, image_cross_attention=image_cross_attention,
                          image_cross_attention_scale=image_cross_attention_scale,
                          image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                          text_context_len=text_context_len)
        self.image_cross_attention = image_cross_attention
        self.norm1, self.norm2, self.norm3 = LayerNorm(dim), LayerNorm(dim), LayerNorm(dim)
        self.checkpoint = checkpoint
        self.domain_cross_attention = False


class _TransformerStack(nn.Module):
    """Shared skeleton: GroupNorm(32, eps 1e-6) -> proj_in -> blocks -> zero-initialised proj_out (+ input)."""

    def _build(self, in_channels, inner, use_linear, conv_cls, blocks):
        self.in_channels = in_channels
        self.norm = GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        if use_linear:
            self.proj_in = Linear(in_channels, inner)
        else:
            self.proj_in = conv_cls(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(blocks)
        if use_linear:
            self.proj_out = zero_module(Linear(inner, in_channels))
        else:
            self.proj_out = zero_module(conv_cls(inner, in_channels, kernel_size=1, stride=1, padding=0))
        self.use_linear = use_linear


class SpatialTransformer(_TransformerStack):
    """Per-frame transformer over the H*W tokens: self-attention, text+image cross-attention, GEGLU MLP."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, use_checkpoint=True,
                 disable_self_attn=False, use_linear=False, video_length=None, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, domain_cross_attention=False, num_tasks=1):
        super().__init__()
        inner = n_heads * d_head
        blocks = [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                        disable_self_attn=disable_self_attn, checkpoint=use_checkpoint,
                                        video_length=video_length, image_cross_attention=image_cross_attention,
                                        image_cross_attention_scale_learnable=image_cross_attention_scale_learnable,
                                        domain_cross_attention=domain_cross_attention, num_tasks=num_tasks)
                  for _ in range(depth)]
        self._build(in_channels, inner, use_linear, Conv2d, blocks)

    def forward(self, x, context=None, **kwargs):
        from mudg_amd.engine import standalone
        return standalone.spatial_transformer(self, x, context)


class TemporalTransformer(_TransformerStack):
    """Per-pixel transformer over the T frames (two self-attentions + GEGLU MLP when only_self_att)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, use_checkpoint=True,
                 use_linear=False, only_self_att=True, causal_attention=False, causal_block_size=1,
                 relative_position=False, temporal_length=None, temporal_frozen=False):
        super().__init__()
        _unsupported(relative_position, "relative position attention")
        _unsupported(causal_attention, "causal temporal attention")
        _unsupported(not only_self_att, "temporal cross-attention")
        self.only_self_att, self.relative_position = True, False
        self.causal_attention, self.causal_block_size = False, causal_block_size
        inner = n_heads * d_head

        def attention_cls(**kw):
            return CrossAttention(temporal_length=temporal_length, **kw)

        blocks = [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=None,
                                        attention_cls=attention_cls, checkpoint=use_checkpoint)
                  for _ in range(depth)]
        self._build(in_channels, inner, use_linear, Conv1d, blocks)
        if temporal_frozen:
            self._frozen_model()

    def _frozen_model(self):
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x, context=None):
        from mudg_amd.engine import standalone
        return standalone.temporal_transformer(self, x)
