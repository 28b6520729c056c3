"""Executor of the 3D-UNet on MI355X: walks a boundary UNetModel (lvdm.modules.networks.openaimodel3d) and runs it
as a sequence of HIP kernel launches on channels-last bf16 "rows" (pixels x channels) — no permutes, no eager ops.

Reference semantics reproduced (file:line in the reference tree):
  UNetModel.forward                     openaimodel3d.py:567-628
  ResBlock._forward / TemporalConvBlock openaimodel3d.py:210-236 / 272-279
  SpatialTransformer / TemporalTransformer / BasicTransformerBlock / CrossAttention / GEGLU
                                        attention.py:451-467 / 529-576 / 392-400 / 81-144 / 579-606
Layout: the reference flips between (b t) c h w, b c t h w, (b hw) t c ...; here every activation is one matrix
whose rows are ordered ((b t) h w) with channels contiguous, so all of those rearranges are no-ops.  Fusions:
bias / timestep-embedding add / residual add / GEGLU ride in GEMM epilogues, nearest-upsample, stride-2 and the
skip-concat ride in the implicit-GEMM loader, V is produced already transposed for the flash kernel.

Precision: MFMA operands (normalised activations, weights) are bf16 with fp32 accumulation; the RESIDUAL STREAM —
every tensor that a later block adds onto: block outputs, the transformers' inner token stream, the encoder skips —
is kept in ops.STREAM(): IEEE fp16 with bf16 operands (11 significand bits: three more than a bf16 operand, so
the ~150 residual adds stay below the operand rounding — a bf16 stream was measured to dominate the end-to-end error,
2.1e-2 against 1.6e-2 per forward — at half the HBM bytes of fp32; it is also what the reference's stream is under
torch.autocast), fp32 with fp16 operands and in the split-operand precision builds.  GroupNorm / LayerNorm read it and
compute in fp32.
"""
import torch
import torch.nn as nn

from .. import ops
from . import packing as pk



import os as _os
_LEAN_ATTENTION = _os.environ.get("MUDG_ATTN_LEAN", "1") != "0"       # A/B switch, read once
_TCONV_SLAB = _os.environ.get("MUDG_TCONV_SLAB", "1") != "0"         # A/B switch: slab-major temporal convs (tiles of 8 pixels x 16 frames)
_FP8_ATTENTION = _os.environ.get("MUDG_ATTN_FP8", "0") == "1"         # opt-in: MX-fp8 scores in the long self-attention


class _Ctx:
    """Per-forward state shared by all blocks."""
    __slots__ = ("B", "T", "emb", "text", "img", "n_text", "n_img", "img_div", "kv_cache", "replicas", "skips")

    def __init__(self):
        self.replicas, self.skips = 1, None

    def fan_out(self, rows):
        """End of the context-free prefix of a forward over guidance replicas (see forward): the batch becomes
        replicas * B — `rows`, the per-clip embeddings and the skip tensors made so far are repeated."""
        r, self.replicas = self.replicas, 1
        self.B *= r
        self.emb = self.emb.repeat(r, 1)
        if self.skips is not None:
            self.skips[:] = [(ops.repeat_rows(t, r), h, w) for t, h, w in self.skips]
        return ops.repeat_rows(rows, r)


# ------------------------------------------------------------------------------------------------ building blocks
def _gn(mod, x, x2, samples, rows, silu):
    return ops.groupnorm(x, pk.f32(mod, "weight"), pk.f32(mod, "bias"), samples=samples, rows=rows, eps=mod.eps,
                         silu=silu, groups=mod.num_groups, x2=x2)


def _ln(mod, x):
    return ops.layernorm(x, pk.f32(mod, "weight"), pk.f32(mod, "bias"), eps=mod.eps)


def _linear(mod, x, residual=None, x2=None, stream=False, stats=False, hw=0):
    """stream=True: the result is a residual-stream tensor and is written in ops.STREAM() (see module docstring).
    stats=True: the result feeds a GroupNorm next — the epilogue also writes that norm's partial sums.
    hw: the rows of one frame (a hint for the library's tile choice, MudgGemmDesc.HW in mode 0)."""
    return ops.gemm(x, pk.linear(mod), bias=pk.f32(mod, "bias"), residual=residual, x2=x2, out_stream=stream, stats=stats, frame_rows=hw)


def _conv3x3(mod, x, frames, h, w, *, stride=1, upsample=False, gbias=None, rows_per_group=0, residual=None, x2=None,
             stream=False, stats=True, fp32=False):
    """Every 3x3 conv of the UNet is followed by a GroupNorm (the next block's, or out_layers'): stats defaults on."""
    wmat, cpad, korder = pk.conv3x3(mod)
    return ops.conv3x3(x, wmat, frames=frames, hin=h, win=w, cin=cpad, stride=stride, upsample=upsample,
                       bias=pk.f32(mod, "bias"), gbias=gbias, rows_per_group=rows_per_group, residual=residual, x2=x2,
                       out_stream=stream, out_fp32=fp32, korder=korder, stats=stats)


_SUBPIXEL = _os.environ.get("MUDG_SUBPIXEL", "1") != "0"          # A/B switch: sub-pixel form of the upsample convs


def upsample_conv(mod, x, frames, h, w, stream=True):
    """Nearest-2x upsample + 3x3 conv (openaimodel3d.py:92-106; ae_modules.py:77-92 for the VAE): as four 2x2 convs on the
    low-resolution rows when the layer qualifies (4/9 of the multiply-adds, MudgGemmDesc.subpixel), else through the
    upsampling loader of the 16-wave kernel.  Neither writes GroupNorm partials."""
    wsub = pk.conv3x3_subpixel(mod) if _SUBPIXEL else None
    if wsub is not None:
        out = ops.conv3x3_up2(x, wsub, frames=frames, hin=h, win=w, cin=mod.weight.shape[1], bias=pk.f32(mod, "bias"),
                              out_stream=stream)
        if out is not None:
            return out
    wmat, cpad, korder = pk.conv3x3(mod)
    return ops.conv3x3(x, wmat, frames=frames, hin=h, win=w, cin=cpad, upsample=True, bias=pk.f32(mod, "bias"),
                       out_stream=stream, korder=korder, stats=False)


def _vt_projection(mod, src_rows, batches, n_per_batch):
    """V^T per batch entry: out[b] = W_v (C x K) @ src[b]^T (K x n) -> [batches * C, ld] with ld = n rounded to 8."""
    w = pk.linear(mod)
    c, k = w.shape
    ld = (n_per_batch + 7) // 8 * 8
    out = ops.empty_rows(batches * c, ld, ops.H16(), src_rows.device)
    ops.gemm(w, src_rows, out=out, batch=batches, sx=0, sw=n_per_batch * src_rows.stride(0), sy=c * out.stride(0),
             M=c, N=n_per_batch, K=k)
    return out, out.stride(0)


def _feed_forward(ff, x_norm, residual, stream=True, hw=0):
    wg, bg = pk.geglu(ff.net[0].proj)
    hidden = ops.gemm(x_norm, wg, bias=bg, geglu=True, frame_rows=hw)
    return _linear(ff.net[2], hidden, residual=residual, stream=stream, hw=hw)


def temporal_conv_block(mod, x, ctx, hw):
    """identity + conv4(conv3(conv2(conv1(x)))); GroupNorm statistics span (C/32, T, H, W) of each clip."""
    y = x
    stages = (mod.conv1, mod.conv2, mod.conv3, mod.conv4)
    for i, seq in enumerate(stages):
        norm, conv = seq[0], seq[-1]
        y = _gn(norm, y, None, ctx.B, ctx.T * hw, True)
        last = i == len(stages) - 1
        # conv1-3 feed the block's own clip-level GroupNorms: tiles of 8 pixels x 16 frames, one staged slab for the three taps
        # (korder 1).  conv4's partials go to whatever follows (a frame-level norm reads 128 consecutive rows): the plain order.
        # (where the library runs the conv on its 288 x 320 tile — whole-tile frames — the plain order is asked for: its tiles are rows)
        slab = (not last and _TCONV_SLAB and ops.tconv3_slab_ok(ctx.T, hw, conv.weight.shape[1])
                and not ops.tconv3_wide(ctx.T, hw, conv.weight.shape[1], conv.weight.shape[0]))
        y = ops.tconv3(y, pk.tconv(conv, slab), clips=ctx.B, t=ctx.T, hw=hw, cin=conv.weight.shape[1],
                       bias=pk.f32(conv, "bias"), residual=x if last else None, out_stream=last, stats=True, korder=int(slab))
    return y


def res_block(mod, x, x2, h, w, ctx):
    """x (+ x2: the skip tensor of a decoder stage, read in place of torch.cat) -> rows [F*h*w, out_channels]."""
    frames, hw = ctx.B * ctx.T, h * w
    a = _gn(mod.in_layers[0], x, x2, frames, hw, True)
    lin = mod.emb_layers[1]
    emb_out = ops.small_linear(ctx.emb, pk.f32(lin, "weight"), pk.f32(lin, "bias"), act_in=True)     # (B, Cout) fp32
    a = _conv3x3(mod.in_layers[2], a, frames, h, w, gbias=emb_out, rows_per_group=ctx.T * hw)
    a = _gn(mod.out_layers[0], a, None, frames, hw, True)
    if isinstance(mod.skip_connection, nn.Identity):
        if x2 is not None:
            raise RuntimeError("identity skip with a concatenated input")
        skip = x
    else:   # 1x1 conv on the raw stream: the MFMA operand copy is bf16, the result goes back to the fp32 stream
        skip = _linear(mod.skip_connection, ops.cast_bf16(x), x2=None if x2 is None else ops.cast_bf16(x2), stream=True, hw=hw)
    out = _conv3x3(mod.out_layers[3], a, frames, h, w, residual=skip, stream=True)
    if mod.use_temporal_conv:
        out = temporal_conv_block(mod.temopral_conv, out, ctx, hw)
    return out


def _cross_kv(attn, ctx):
    """K and V^T of the text tokens (shared by a clip's frames) and of the image tokens for one attn2 layer."""
    k_text = ops.gemm(ctx.text, pk.linear(attn.to_k))
    vt_text, ld_text = _vt_projection(attn.to_v, ctx.text, ctx.B, ctx.n_text)
    if ctx.img is None or not attn.image_cross_attention:
        return k_text, vt_text, ld_text, None, None, 0
    k_img = ops.gemm(ctx.img, pk.linear(attn.to_k_ip))
    vt_img, ld_img = _vt_projection(attn.to_v_ip, ctx.img, ctx.img.shape[0] // ctx.n_img, ctx.n_img)
    return k_text, vt_text, ld_text, k_img, vt_img, ld_img


def spatial_block(blk, hcur, frames, hw, ctx, last):
    a1, a2 = blk.attn1, blk.attn2
    c, heads = hcur.shape[1], a1.heads
    # self-attention over the hw tokens of each frame
    n1 = _ln(blk.norm1, hcur)
    # 16-bit builds and bf16x3: scale * log2(e) rides in the packed q weights (folded in fp32 before the one cast / split)
    lean = _LEAN_ATTENTION and ops.hip.planes() <= 2
    wqk = pk.qk_prescaled(a1, a1.to_q, a1.to_k, a1.scale) if lean else pk.linear_cat(a1, "qk", (a1.to_q, a1.to_k))
    use_fp8 = lean and ops.hip.planes() == 1 and _FP8_ATTENTION and hw >= 512 and hw % 64 == 0      # BASELINE config 5: Q K^T on the MX-fp8 MFMA
    fp8 = None
    if use_fp8:         # the projection's own epilogue writes the MX-fp8 copy of q | k (no separate quantisation passes)
        qk, q8, s8 = ops.gemm(n1, wqk, fp8=True)
        fp8 = (q8[:, :c], s8[:, :c // 32], q8[:, c:], s8[:, c // 32:])
    else:
        qk = ops.gemm(n1, wqk, frame_rows=hw)
    vt, ldv = _vt_projection(a1.to_v, n1, frames, hw)
    att = ops.empty_rows(frames * hw, c, ops.H16(), hcur.device)
    ops.attention(qk[:, :c], qk[:, c:], vt, att, frames=frames, heads=heads, nq=hw, nk=hw, ldvt=ldv, svt=c * ldv,
                  scale=a1.scale, q_prescaled=lean, fp8=fp8)
    hcur = _linear(a1.to_out[0], att, residual=hcur, stream=True, hw=hw)
    if ctx.replicas > 1:        # first use of the context in this forward: from here on the guidance replicas differ
        hcur = ctx.fan_out(hcur)
        frames = ctx.B * ctx.T
    # text (+ image) cross-attention: two softmaxes, outputs summed (image_cross_attention_scale == 1)
    n2 = _ln(blk.norm2, hcur)
    q2 = ops.gemm(n2, pk.linear(a2.to_q), frame_rows=hw)
    key = id(a2)
    if key not in ctx.kv_cache:
        ctx.kv_cache[key] = _cross_kv(a2, ctx)
    k_text, vt_text, ld_text, k_img, vt_img, ld_img = ctx.kv_cache[key]
    att2 = ops.empty_rows(frames * hw, c, ops.H16(), hcur.device)
    if k_img is not None and a2.image_cross_attention_scale != 1.0:
        raise NotImplementedError("image_cross_attention_scale != 1.0")
    # both softmaxes (text keys, image keys) in ONE launch: q2 is read once, the summed result written once
    ops.attention(q2, k_text, vt_text, att2, frames=frames, heads=heads, nq=hw, nk=ctx.n_text, ldvt=ld_text,
                  svt=c * ld_text, kv_div=ctx.T, scale=a2.scale, k2=k_img, vt2=vt_img, nk2=ctx.n_img,
                  ldvt2=ld_img if k_img is not None else None, svt2=c * ld_img if k_img is not None else None,
                  kv_div2=ctx.img_div)
    hcur = _linear(a2.to_out[0], att2, residual=hcur, stream=True, hw=hw)
    # the last block's output only feeds proj_out (an MFMA operand), so it is written as bf16 directly
    return _feed_forward(blk.ff, _ln(blk.norm3, hcur), hcur, stream=not last, hw=hw)


def spatial_transformer(mod, x, h, w, ctx):
    hw = h * w
    cur = _linear(mod.proj_in, _gn(mod.norm, x, None, ctx.B * ctx.T, hw, False), stream=True, hw=hw)
    n = len(mod.transformer_blocks)
    for i, blk in enumerate(mod.transformer_blocks):
        shared = ctx.replicas
        cur = spatial_block(blk, cur, ctx.B * ctx.T, hw, ctx, i == n - 1)
        if ctx.replicas != shared:          # the block fanned the batch out: so must the residual input
            x = ops.repeat_rows(x, shared)
    return _linear(mod.proj_out, cur, residual=x, stream=True, stats=True, hw=hw)


def temporal_block(blk, hcur, hw, ctx, last):
    for attn, norm in ((blk.attn1, blk.norm1), (blk.attn2, blk.norm2)):     # both are self-attention over T
        c = hcur.shape[1]
        qkv = ops.gemm(_ln(norm, hcur), pk.linear_cat(attn, "qkv", (attn.to_q, attn.to_k, attn.to_v)), frame_rows=hw)
        att = ops.empty_rows(hcur.shape[0], c, ops.H16(), hcur.device)
        ops.temporal_attention(qkv, att, clips=ctx.B, t=ctx.T, hw=hw, heads=attn.heads, scale=attn.scale)
        hcur = _linear(attn.to_out[0], att, residual=hcur, stream=True, hw=hw)
    return _feed_forward(blk.ff, _ln(blk.norm3, hcur), hcur, stream=not last, hw=hw)


def temporal_transformer(mod, x, h, w, ctx):
    hw = h * w
    cur = _linear(mod.proj_in, _gn(mod.norm, x, None, ctx.B, ctx.T * hw, False), stream=True, hw=hw)
    n = len(mod.transformer_blocks)
    for i, blk in enumerate(mod.transformer_blocks):
        cur = temporal_block(blk, cur, hw, ctx, i == n - 1)
    return _linear(mod.proj_out, cur, residual=x, stream=True, stats=True, hw=hw)


def run_stage(seq, x, x2, h, w, ctx):
    """One TimestepEmbedSequential.  Returns (rows, h, w)."""
    for m in seq:
        name = type(m).__name__
        frames = ctx.B * ctx.T
        if name == "ResBlock":
            x, x2 = res_block(m, x, x2, h, w, ctx), None
        elif name == "SpatialTransformer":
            x = spatial_transformer(m, x, h, w, ctx)
        elif name == "TemporalTransformer":
            x = temporal_transformer(m, x, h, w, ctx)
        elif name == "Downsample":
            x = _conv3x3(m.op, ops.cast_bf16(x), frames, h, w, stride=2, stream=True)
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        elif name == "Upsample":
            # nearest-2x convs run on the 16-wave 256x256 kernel, which does not emit GroupNorm partials
            x = upsample_conv(m.conv, ops.cast_bf16(x), frames, h, w)
            h, w = 2 * h, 2 * w
        elif isinstance(m, nn.Conv2d):                      # the stem (possibly swapped in by training-time surgery)
            x = _conv3x3(m, x, frames, h, w, stream=True)
        else:
            raise NotImplementedError(f"no MI355X implementation for UNet stage member {name}")
    return x, h, w


# ------------------------------------------------------------------------------------------------ conditioning
def _embed_mlp(seq, sin):
    l0, l2 = seq[0], seq[2]
    hid = ops.small_linear(sin, pk.f32(l0, "weight"), pk.f32(l0, "bias"), act_out=True)
    return ops.small_linear(hid, pk.f32(l2, "weight"), pk.f32(l2, "bias"))


def _to_long(v, n, device, what):
    if v is None:
        raise AssertionError(f"{what} is required")
    v = torch.as_tensor(v, device=device)
    if v.numel() != n:
        raise ValueError(f"{what}: expected {n} entries, got {tuple(v.shape)}")
    return v.reshape(n).to(torch.int64)


class PreparedContext:
    """The cross-attention conditioning of one sampling run, made ready ONCE: the (B, L, D) context tokens as MFMA operand
    rows (text / image tokens in their own matrices) and every attn2 layer's K and V^T projections of them.  The tokens
    are the same at all 50 DDIM steps and in every pass of the guidance batch, so a sampler hands the same
    PreparedContext to every UNet call (UNetModel.prepare_context; DDIMSampler does) instead of having each forward
    re-project them (the reference recomputes to_k / to_v per layer per call, attention.py:87-94).
    The projections are redone when any model parameter changed since they were made (`signature`)."""

    def __init__(self, model, context, t_len, device=None, project=True):
        if context is None:
            raise AssertionError("context is required (text + per-frame image tokens)")
        if context.dim() != 3:
            raise ValueError(f"context must be (B, L, D), got {tuple(context.shape)}")
        if not context.is_cuda:
            raise RuntimeError("context must be a GPU tensor")
        device = device or context.device
        self.B, L, D = context.shape
        if L <= 77:
            raise ValueError("context needs more than the 77 text tokens (image tokens follow them)")
        flat = context.contiguous()
        if flat.dtype != torch.float32:
            flat = flat.float()
        self.shape, self.T, self.n_text = tuple(context.shape), t_len, 77
        # openaimodel3d.py:581-587: per-frame image tokens when L == 77 + 16 T, otherwise the whole context per frame
        if L == 77 + 16 * t_len:
            self.n_img, self.img_div = 16, 1
        else:
            self.n_img, self.img_div = L - 77, t_len
        text = ops.empty_rows(self.B * 77, D, ops.H16(), device)
        img = ops.empty_rows(self.B * (L - 77), D, ops.H16(), device)
        for b in range(self.B):       # fp32 tokens -> MFMA operand rows, text and image tokens into their own matrices
            ops.cast_rows(flat[b, :77], text[b * 77:(b + 1) * 77])
            ops.cast_rows(flat[b, 77:], img[b * (L - 77):(b + 1) * (L - 77)])
        self.text, self.img = text, img
        self.kv, self.signature = {}, None
        if project:
            self.project(model)

    def project(self, model):
        """K / V^T of the text and image tokens for every spatial transformer's attn2, in module order."""
        from .graph import _params_signature
        self.kv = {}
        for m in model.modules():
            if type(m).__name__ == "SpatialTransformer":
                for blk in m.transformer_blocks:
                    self.kv[id(blk.attn2)] = _cross_kv(blk.attn2, self)
        self.signature = _params_signature(model)

    def tensors(self):
        """Every device buffer, in a fixed order (graph replays copy a new context's buffers over the captured one's)."""
        out = [self.text, self.img]
        for key in self.kv:
            out += [t for t in self.kv[key] if torch.is_tensor(t)]
        return out

    def clone(self):
        twin = object.__new__(PreparedContext)
        twin.__dict__.update(self.__dict__)
        twin.text, twin.img = _clone_rows(self.text), _clone_rows(self.img)
        twin.kv = {k: tuple(_clone_rows(t) if torch.is_tensor(t) else t for t in v) for k, v in self.kv.items()}
        return twin

    def copy_from(self, other):
        mine, theirs = self.tensors(), other.tensors()
        if len(mine) != len(theirs) or self.shape != other.shape:
            raise RuntimeError("PreparedContext.copy_from: different layouts")
        for dst, src in zip(mine, theirs):
            _base(dst).copy_(_base(src))
        self.signature = other.signature

    def bind(self, ctx, model):
        """Attach to a forward's state; stale projections (parameters changed since they were made) are redone."""
        from .graph import _params_signature
        if ctx.B * ctx.replicas != self.B:
            raise ValueError(f"context prepared for batch {self.B}, forward has batch {ctx.B * ctx.replicas}")
        if self.signature != _params_signature(model):
            self.project(model)
        ctx.text, ctx.img, ctx.n_text, ctx.n_img, ctx.img_div, ctx.kv_cache = self.text, self.img, self.n_text, self.n_img, \
            self.img_div, self.kv


def _base(t):
    """The whole allocation behind an operand view (split-operand builds: the [rows, planes * cols] buffer)."""
    return t if t._base is None else t._base


def _clone_rows(t):
    full = _base(t).clone()
    return full if t._base is None else full[:, :t.shape[1]]


def make_context(model, ctx, context, t_len, device):
    """Per-forward conditioning state from a raw (B, L, D) context tensor (a one-shot PreparedContext)."""
    if context is not None and context.dim() == 3 and context.shape[0] != ctx.B * ctx.replicas:
        raise ValueError(f"context must be (B, L, D), got {tuple(context.shape)}")
    prepared = PreparedContext(model, context, t_len, device, project=False)
    prepared.T = ctx.T
    ctx.text, ctx.img, ctx.n_text, ctx.n_img, ctx.img_div, ctx.kv_cache = prepared.text, prepared.img, prepared.n_text, \
        prepared.n_img, prepared.img_div, {}


@torch.no_grad()
def forward_entry(model, x, timesteps, c_label=None, context=None, features_adapter=None, fs=None):
    """UNetModel.forward: eager launch sequence, or hipGraph replay when `model.use_hip_graph` is set."""
    if context is None:
        raise AssertionError("context is required (text + per-frame image tokens)")
    if not isinstance(context, PreparedContext) and (not torch.is_tensor(context) or context.dim() != 3):
        raise ValueError(f"context must be (B, L, D), got {tuple(context.shape) if torch.is_tensor(context) else type(context)}")
    if getattr(model, "use_hip_graph", False) and features_adapter is None \
            and not torch.cuda.is_current_stream_capturing():
        from .graph import UNetGraphs
        graphs = model.__dict__.get("_mudg_graphs")
        if graphs is None:
            graphs = model.__dict__["_mudg_graphs"] = UNetGraphs(model)
        parts = list(x) if isinstance(x, (list, tuple)) else [x]
        if all(p.is_cuda for p in parts) and (isinstance(context, PreparedContext) or context.is_cuda):
            from .. import hip
            if not hip.prof_enabled():           # per-kernel hipEvents cannot be recorded inside a graph
                return graphs(parts, timesteps, c_label, context, fs)
    return forward(model, x, timesteps, c_label=c_label, context=context, features_adapter=features_adapter, fs=fs)


@torch.no_grad()
def forward(model, x, timesteps, c_label=None, context=None, features_adapter=None, fs=None):
    if features_adapter is not None:
        raise NotImplementedError("features_adapter is not used on the MuDG path")
    parts = list(x) if isinstance(x, (list, tuple)) else [x]
    first = parts[0]
    if first.dim() != 5:
        raise ValueError(f"x must be (B, C, T, H, W), got {tuple(first.shape)}")
    if not first.is_cuda:
        raise RuntimeError("UNetModel.forward: inputs must be on the GPU; the MI355X path has no CPU fallback")
    B, _, T, H, W = first.shape
    device = first.device
    cin = sum(p.shape[1] for p in parts)
    if cin != model.in_channels:
        raise ValueError(f"expected {model.in_channels} input channels, got {cin}")
    stem = model.input_blocks[0][0]
    cpad = (stem.weight.shape[1] + 7) // 8 * 8

    ctx = _Ctx()
    ctx.B, ctx.T, ctx.kv_cache = B, T, {}
    # Guidance replicas: a context of batch R * B against latents of batch B means "the same latents under R
    # conditionings" (the cond / uncond / image-only passes of classifier-free guidance).  Everything up to the first
    # cross-attention — stem, init_attn, the first ResBlock, the first spatial self-attention: all at full resolution —
    # does not see the context, so it runs ONCE on B; the rows, the embeddings and the skips made so far are then
    # repeated R times (_Ctx.fan_out) and the rest runs on R * B.  Bit-identical to running the R * B batch throughout
    # (every kernel is per clip / per frame / per pixel).
    cb = context.B if isinstance(context, PreparedContext) else (context.shape[0] if context is not None and context.dim() == 3 else B)
    if cb != B and (cb % B != 0 or cb < B):
        raise ValueError(f"context batch {cb} is not a multiple of the latent batch {B}")
    ctx.replicas = cb // B
    # ---- embeddings: time (+ class) then + fps, all per clip (openaimodel3d.py:569-602)
    mc = model.model_channels
    ts = _to_long(timesteps, B, device, "timesteps")
    emb = _embed_mlp(model.time_embed, ops.timestep_embedding(ts, mc))
    if model.class_label_condition:
        lab = _to_long(c_label, B, device, "class_label")
        ops.add_(emb, _embed_mlp(model.class_embed, ops.timestep_embedding(lab, mc)))
    if model.fs_condition:
        fsv = torch.full((B,), model.default_fs, dtype=torch.int64, device=device) if fs is None else _to_long(fs, B, device, "fs")
        ops.add_(emb, _embed_mlp(model.fps_embedding, ops.timestep_embedding(fsv, mc)))
    ctx.emb = emb
    if isinstance(context, PreparedContext):
        context.bind(ctx, model)
    else:
        make_context(model, ctx, context, T, device)

    # ---- input: (b c t h w) pieces -> rows with channels side by side (replaces torch.cat + rearrange, 591 / ddpm3d 1317)
    rows = ops.empty_rows(B * T * H * W, cpad, ops.H16(), device)
    off = 0
    for p in parts:
        if p.shape[0] != B or p.shape[2:] != first.shape[2:]:
            raise ValueError("all input pieces must share (B, T, H, W)")
        src = p if p.dtype in (torch.float32, ops.H16()) else p.float()
        ops.ncthw_to_rows(src, rows, off)
        off += p.shape[1]
    if cpad > off:
        ops.zero_channels(rows, off, cpad)

    h, w = H, W
    skips = ctx.skips = []
    cur = rows
    for i, stage in enumerate(model.input_blocks):
        cur, h, w = run_stage(stage, cur, None, h, w, ctx)
        if i == 0 and model.addition_attention:
            cur, h, w = run_stage(model.init_attn, cur, None, h, w, ctx)
        skips.append((cur, h, w))
    cur, h, w = run_stage(model.middle_block, cur, None, h, w, ctx)
    for stage in model.output_blocks:
        skip, sh, sw = skips.pop()
        if (sh, sw) != (h, w):
            raise RuntimeError(f"skip resolution {sh}x{sw} does not match {h}x{w}: H and W must be divisible by "
                               f"{2 ** (len(model.channel_mult) - 1)}")
        cur, h, w = run_stage(stage, cur, skip, h, w, ctx)
    if ctx.replicas > 1:                     # a UNet without any cross-attention: the replicas are plain copies
        cur = ctx.fan_out(cur)
    norm, conv = model.out[0], model.out[2]
    cur = _gn(norm, cur, None, ctx.B * T, h * w, True)
    y = _conv3x3(conv, cur, ctx.B * T, h, w, fp32=True, stats=False)      # the prediction itself leaves in fp32
    out_dtype = first.dtype if first.dtype in (torch.float32, ops.H16()) else torch.float32
    out = ops.rows_to_ncthw(y, (ctx.B, model.out_channels, T, h, w), dtype=out_dtype)
    return out if out.dtype == first.dtype else out.to(first.dtype)
