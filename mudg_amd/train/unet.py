"""UNetModel.forward for TRAINING: the same walk over a boundary UNetModel as mudg_amd/engine/unet.py, every layer expressed
with the autograd Functions of functions.py, so that loss.backward() fills `.grad` of the reference-named nn.Parameters.

The inference executor fuses aggressively (fp16 stream, GEGLU and norm statistics in GEMM epilogues, K / V cached per run,
hipGraph replay); a training forward keeps what the backward pass needs instead: fp32 activations between layers, GEGLU as
its own kernel over the saved pre-activation, conditioning projected inside the graph.  Reference: openaimodel3d.py:567-628
(forward), :210-236 / :272-279 (ResBlock / TemporalConvBlock), attention.py:392-400, 451-467, 529-576, 81-144."""
import torch
import torch.nn as nn

from .. import hip, ops
from . import functions as F_


def _lin(mod, x, residual=None):
    return F_.Linear.apply(x, mod.weight, mod.bias, residual)


def _lin_packed(mods, x):
    """The bias-free projections `mods` of one input as ONE GEMM: the weights are concatenated (autograd hands each its block
    of the gradient), the input gradient comes back from one GEMM over the packed output gradient instead of a GEMM and an
    accumulation per projection."""
    if any(m.bias is not None for m in mods):
        raise NotImplementedError("packed projections are bias-free (to_q / to_k / to_v)")
    return F_.Linear.apply(x, torch.cat([m.weight for m in mods], dim=0), None, None)


def _gn(mod, x, samples, rows, silu):
    return F_.GroupNorm.apply(x, mod.weight, mod.bias, samples, rows, mod.eps, silu, mod.num_groups)


def _gn_skip(mod, x, samples, rows, silu):
    """(GroupNorm(x), x) with the skip routed through the norm's node (functions.GroupNormSkip)."""
    return F_.GroupNormSkip.apply(x, mod.weight, mod.bias, samples, rows, mod.eps, silu, mod.num_groups)


def _ln(mod, x):
    return F_.LayerNorm.apply(x, mod.weight, mod.bias, mod.eps)


def _ln_skip(mod, x):
    """(LayerNorm(x), x) with the skip routed through the norm's node (functions.LayerNormSkip)."""
    return F_.LayerNormSkip.apply(x, mod.weight, mod.bias, mod.eps)


def _conv(mod, x, frames, h, w, stride=1, gbias=None, rows_per_group=None, residual=None):
    return F_.Conv3x3.apply(x, mod.weight, mod.bias, gbias, residual, (frames, h, w, stride), rows_per_group)


class _Ctx:
    __slots__ = ("B", "T", "emb", "text", "img", "n_img", "img_div")


def _ckpt(flag, fn, *args):
    """lvdm/common.py:81-93 `checkpoint(func, inputs, params, flag)`: with the flag set (use_checkpoint of the UNet config) the
    block's activations are not kept — its forward is replayed during backward.  torch.utils.checkpoint restores the RNG state
    for the replay, so dropout masks (seeded from the CPU generator) come out the same."""
    if not flag:
        return fn(*args)
    from torch.utils.checkpoint import checkpoint
    return checkpoint(fn, *args, use_reentrant=False)


def temporal_conv_block(mod, x, ctx, hw):
    y = x
    stages = (mod.conv1, mod.conv2, mod.conv3, mod.conv4)
    for i, seq in enumerate(stages):
        norm, conv = seq[0], seq[-1]
        if i == 0:
            y, x = _gn_skip(norm, y, ctx.B, ctx.T * hw, True)          # x: the identity the block adds back at its end
        else:
            y = _gn(norm, y, ctx.B, ctx.T * hw, True)
        if len(seq) == 4:                                   # GroupNorm, SiLU, Dropout, conv (openaimodel3d.py:256-266)
            y = F_.dropout(seq[2], y)
        y = F_.TConv3.apply(y, conv.weight, conv.bias, x if i == len(stages) - 1 else None, (ctx.B, ctx.T, hw))
    return y


def res_block(mod, x, h, w, ctx):
    with F_.frame_rows(h * w):
        return _res_block(mod, x, h, w, ctx)


def _res_block(mod, x, h, w, ctx):
    frames, hw = ctx.B * ctx.T, h * w
    identity = isinstance(mod.skip_connection, nn.Identity)
    a, x = _gn_skip(mod.in_layers[0], x, frames, hw, True)             # x: what the skip branch (identity or 1x1 projection) reads
    emb_out = _lin(mod.emb_layers[1], F_.Silu.apply(ctx.emb))                          # (B, Cout): one row per clip
    a = _conv(mod.in_layers[2], a, frames, h, w, gbias=emb_out, rows_per_group=ctx.T * hw)
    a = F_.dropout(mod.out_layers[2], _gn(mod.out_layers[0], a, frames, hw, True))
    skip = x if identity else _lin(mod.skip_connection, x)
    out = _conv(mod.out_layers[3], a, frames, h, w, residual=skip)
    if mod.use_temporal_conv:
        out = temporal_conv_block(mod.temopral_conv, out, ctx, hw)
    return out


def _out_proj(attn, att, residual):
    """to_out = Sequential(Linear, Dropout) and then "+ x" (attention.py:144, 393-396): the residual rides in the GEMM epilogue
    unless the dropout between them is active."""
    drop = attn.to_out[1] if len(attn.to_out) > 1 else None
    if drop is not None and drop.training and drop.p > 0.0:
        return F_.Add.apply(F_.dropout(drop, _lin(attn.to_out[0], att)), residual)
    return _lin(attn.to_out[0], att, residual=residual)


def _feed_forward(ff, x_norm, residual):
    hidden = F_.geglu_dropout(ff.net[1], _lin(ff.net[0].proj, x_norm))
    return _lin(ff.net[2], hidden, residual=residual)


def spatial_block(blk, cur, frames, hw, ctx):
    with F_.frame_rows(hw):
        return _spatial_block(blk, cur, frames, hw, ctx)


def _spatial_block(blk, cur, frames, hw, ctx):
    a1, a2 = blk.attn1, blk.attn2
    n1, cur = _ln_skip(blk.norm1, cur)
    if hip.planes() == 1:
        att = F_.SelfAttention.apply(_lin_packed((a1.to_q, a1.to_k, a1.to_v), n1), (frames, a1.heads, hw, a1.scale))
    else:
        att = F_.Attention.apply(_lin(a1.to_q, n1), _lin(a1.to_k, n1), _lin(a1.to_v, n1), None, None,
                                 (frames, a1.heads, hw, hw, 1, 0, 1, a1.scale))
    cur = _out_proj(a1, att, cur)
    n2, cur = _ln_skip(blk.norm2, cur)
    q2 = _lin(a2.to_q, n2)
    k_t, v_t = _lin(a2.to_k, ctx.text), _lin(a2.to_v, ctx.text)
    if ctx.img is not None and a2.image_cross_attention:
        if a2.image_cross_attention_scale != 1.0:
            raise NotImplementedError("image_cross_attention_scale != 1.0")
        k_i, v_i = _lin(a2.to_k_ip, ctx.img), _lin(a2.to_v_ip, ctx.img)
        att2 = F_.Attention.apply(q2, k_t, v_t, k_i, v_i, (frames, a2.heads, hw, 77, ctx.T, ctx.n_img, ctx.img_div, a2.scale))
    else:
        att2 = F_.Attention.apply(q2, k_t, v_t, None, None, (frames, a2.heads, hw, 77, ctx.T, 0, 1, a2.scale))
    cur = _out_proj(a2, att2, cur)
    n3, cur = _ln_skip(blk.norm3, cur)
    return _feed_forward(blk.ff, n3, cur)


def spatial_transformer(mod, x, h, w, ctx):
    hw, frames = h * w, ctx.B * ctx.T
    n, x = _gn_skip(mod.norm, x, frames, hw, False)
    with F_.frame_rows(hw):
        cur = _lin(mod.proj_in, n)
    for blk in mod.transformer_blocks:
        cur = _ckpt(blk.checkpoint, spatial_block, blk, cur, frames, hw, ctx)
    with F_.frame_rows(hw):
        return _lin(mod.proj_out, cur, residual=x)


def temporal_block(blk, cur, hw, ctx):
    with F_.frame_rows(hw):
        return _temporal_block(blk, cur, hw, ctx)


def _temporal_block(blk, cur, hw, ctx):
    for attn, norm in ((blk.attn1, blk.norm1), (blk.attn2, blk.norm2)):
        n, cur = _ln_skip(norm, cur)
        qkv = _lin_packed((attn.to_q, attn.to_k, attn.to_v), n)                                    # [q | k | v] columns
        att = F_.TemporalAttention.apply(qkv, (ctx.B, ctx.T, hw, attn.heads, attn.scale))
        cur = _out_proj(attn, att, cur)
    n3, cur = _ln_skip(blk.norm3, cur)
    return _feed_forward(blk.ff, n3, cur)


def temporal_transformer(mod, x, h, w, ctx):
    hw = h * w
    n, x = _gn_skip(mod.norm, x, ctx.B, ctx.T * hw, False)
    with F_.frame_rows(hw):
        cur = _lin(mod.proj_in, n)
    for blk in mod.transformer_blocks:
        cur = _ckpt(blk.checkpoint, temporal_block, blk, cur, hw, ctx)
    with F_.frame_rows(hw):
        return _lin(mod.proj_out, cur, residual=x)


def run_stage(seq, x, h, w, ctx):
    for m in seq:
        name = type(m).__name__
        frames = ctx.B * ctx.T
        if name == "ResBlock":
            x = _ckpt(m.use_checkpoint, res_block, m, x, h, w, ctx)
        elif name == "SpatialTransformer":
            x = spatial_transformer(m, x, h, w, ctx)
        elif name == "TemporalTransformer":
            x = temporal_transformer(m, x, h, w, ctx)
        elif name == "Downsample":
            x = _conv(m.op, x, frames, h, w, stride=2)
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        elif name == "Upsample":
            x = _conv(m.conv, F_.Upsample2x.apply(x, (frames, h, w)), frames, 2 * h, 2 * w)
            h, w = 2 * h, 2 * w
        elif isinstance(m, nn.Conv2d):
            x = _conv(m, x, frames, h, w)
        else:
            raise NotImplementedError(f"no training implementation for UNet stage member {name}")
    return x, h, w


def _embed_mlp(seq, sin, residual=None):
    hid = F_.Silu.apply(_lin(seq[0], sin))
    return _lin(seq[2], hid, residual=residual)


def forward(model, x, timesteps, c_label=None, context=None, fs=None):
    """UNetModel.forward with an autograd graph behind it.  x (B, C, T, H, W) fp32 (or the channel pieces of it)."""
    parts = list(x) if isinstance(x, (list, tuple)) else [x]
    first = parts[0]
    if not first.is_cuda:
        raise RuntimeError("UNetModel.forward (training step): inputs must be on the GPU; the MI355X path has no CPU fallback")
    xin = first if len(parts) == 1 else torch.cat(parts, dim=1)
    B, cin, T, H, W = xin.shape
    if cin != model.in_channels:
        raise ValueError(f"expected {model.in_channels} input channels, got {cin}")
    device = xin.device
    ctx = _Ctx()
    ctx.B, ctx.T = B, T
    mc = model.model_channels
    as_long = lambda v, what: torch.as_tensor(v, device=device).reshape(B).to(torch.int64)
    emb = _embed_mlp(model.time_embed, ops.timestep_embedding(as_long(timesteps, "timesteps"), mc))
    if model.class_label_condition:
        emb = _embed_mlp(model.class_embed, ops.timestep_embedding(as_long(c_label, "class_label"), mc), residual=emb)
    if model.fs_condition:
        fsv = torch.full((B,), model.default_fs, dtype=torch.int64, device=device) if fs is None else as_long(fs, "fs")
        emb = _embed_mlp(model.fps_embedding, ops.timestep_embedding(fsv, mc), residual=emb)
    ctx.emb = emb
    # conditioning tokens (openaimodel3d.py:581-587): 77 text tokens per clip, then per-frame (or whole-clip) image tokens
    if context is None or context.dim() != 3 or context.shape[0] != B or context.shape[1] <= 77:
        raise ValueError("context must be (B, 77 + image tokens, D)")
    L, D = context.shape[1], context.shape[2]
    cf = context.float()
    ctx.text = cf[:, :77].reshape(B * 77, D).contiguous()
    ctx.img = cf[:, 77:].reshape(B * (L - 77), D).contiguous()
    ctx.n_img, ctx.img_div = (16, 1) if L == 77 + 16 * T else (L - 77, T)

    rows = F_.ToRows.apply(xin.float())
    h, w = H, W
    skips = []
    cur = rows
    for i, stage in enumerate(model.input_blocks):
        cur, h, w = run_stage(stage, cur, h, w, ctx)
        if i == 0 and model.addition_attention:
            cur, h, w = run_stage(model.init_attn, cur, h, w, ctx)
        skips.append((cur, h, w))
    cur, h, w = run_stage(model.middle_block, cur, h, w, ctx)
    for stage in model.output_blocks:
        skip, sh, sw = skips.pop()
        if (sh, sw) != (h, w):
            raise RuntimeError(f"skip resolution {sh}x{sw} does not match {h}x{w}")
        cur, h, w = run_stage(stage, torch.cat([cur, skip], dim=1), h, w, ctx)            # openaimodel3d.py:621 (layout only)
    cur = _gn(model.out[0], cur, B * T, h * w, True)
    y = _conv(model.out[2], cur, B * T, h, w)
    return F_.FromRows.apply(y, (B, model.out_channels, T, h, w))
